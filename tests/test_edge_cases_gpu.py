"""GPU: edge cases of the call surface -- empty and very short inputs, empty batches, capacity limits.  The reference
reaches these through `process_file` (e.g. `_extract_segment` returns zeros(1, 100) for an invalid slice,
back/api.py:840-860; `_extract_embedding` refuses crops under 0.5 s, 862-876), so they must fail loudly or degrade the
same way, never crash the process or hang the GPU."""
import numpy as np
import pytest
import torch

from tests.conftest import within

from clearconverse_amd import _lib
from clearconverse_amd.audio import synthetic_clip
from clearconverse_amd.weights import (SepDims, WhisperDims, synthetic_pyannet_state_dict, synthetic_sepformer_state_dict,
                                       synthetic_whisper_state_dict, synthetic_xvector_state_dict)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def whisper(ccx_ctx):
    from clearconverse_amd.whisper import WhisperModel
    dims = WhisperDims.mini(n_layer=2, n_state=128)
    m = WhisperModel(dims, synthetic_whisper_state_dict(dims, seed=3), max_batch=4, ctx=ccx_ctx)
    yield m
    m.close()


def test_transcribe_empty_and_tiny_audio(whisper):
    out = whisper.transcribe(np.zeros(0, dtype=np.float32))
    assert out["text"] == "" and out["segments"] == []
    out = whisper.transcribe(np.zeros(100, dtype=np.float32))           # the reference's zeros(1, 100) placeholder crop
    assert isinstance(out["text"], str)
    outs = whisper.transcribe_batch([np.zeros(0, dtype=np.float32), synthetic_clip(1, 30.0)[:16000]])
    assert outs[0]["text"] == "" and isinstance(outs[1]["text"], str)
    assert whisper.transcribe_batch([]) == []


def test_decode_argument_errors(whisper):
    rules = whisper.rules
    with pytest.raises(_lib.CcxError):
        whisper.decode_greedy([[rules.sot]] * 5)                           # more sequences than max_batch
    with pytest.raises(_lib.CcxError):
        whisper.decode_greedy([[rules.sot, 10 ** 6]])                      # token id out of range
    with pytest.raises(_lib.CcxError):
        whisper.decode_greedy([[rules.sot]], sample_len=10 ** 4)           # beyond n_text_ctx
    with pytest.raises(_lib.CcxError):
        whisper.decode([[rules.sot]], temperature=-1.0)


def test_separator_short_and_ragged(ccx_ctx):
    from clearconverse_amd.separator import SepformerSeparator
    from oracle import sepformer_ref as S
    dims = SepDims(n_layers=1, n_blocks=1)
    sd = synthetic_sepformer_state_dict(dims, seed=1)
    sep = SepformerSeparator(dims, sd, max_tokens=20000, max_utts=4, ctx=ccx_ctx)
    rng = np.random.default_rng(0)
    mix = torch.from_numpy(rng.standard_normal((3, 4000)).astype(np.float32) * 0.1).cuda()
    lens = [4000, 40, 1234]                                                # one utterance barely longer than the encoder kernel
    out = sep.separate_batch(mix, lens)
    assert out.shape == (3, 4000, 2) and torch.isfinite(out).all()
    for b, n in enumerate(lens):
        assert float(out[b, n:].abs().max()) == 0.0 if n < 4000 else True   # rows past the utterance are zero
    orc = S.SepformerRef(S.SepDims(**dims.__dict__), sd)
    ref = orc.separate(mix[1:2, :40].cpu())
    got = out[1:2, :40].cpu()
    within("sepformer 2-layer: 40-sample utterance rel-L2", float((got - ref).norm() / (ref.norm() + 1e-9)), 1e-2)
    with pytest.raises(_lib.CcxError):
        sep.separate_batch(mix, [4000, 0, 10])                             # an empty utterance is an error, not a crash


def test_speaker_nets_short_crops_and_empty_batches(ccx_ctx):
    from clearconverse_amd.speaker import SegmentationNet, XVectorEmbedder
    emb = XVectorEmbedder(synthetic_xvector_state_dict(seed=2), max_crops=8, max_samples=16000 * 30, ctx=ccx_ctx)
    assert emb.embed_batch([]).shape == (0, 512)
    with pytest.raises(_lib.CcxError):
        emb.embed_batch([torch.zeros(100)])                                # the zeros(1, 100) placeholder is too short
    ok = emb.embed_batch([torch.from_numpy(synthetic_clip(2, 30.0)[:8000])])   # 0.5 s: the reference's minimum (back/api.py:864)
    assert ok.shape == (1, 512) and torch.isfinite(ok).all()
    seg = SegmentationNet(synthetic_pyannet_state_dict(7, seed=3), n_classes=7, powerset=True, max_crops=8, max_samples=16000 * 30,
                          ctx=ccx_ctx)
    assert seg.segment_numpy([]) == []
    with pytest.raises(_lib.CcxError):
        seg.segment_numpy([torch.zeros(100)])
    emb.close(); seg.close()


def test_spectral_gate_short_and_silent(ccx_ctx):
    from clearconverse_amd.denoise import SpectralGate
    g = SpectralGate(max_samples=48000, max_clips=2, ctx=ccx_ctx)
    y = torch.zeros(2, 48000, device="cuda")
    out = g.reduce_batch(y, [48000, 3000], prop_decrease=0.5)
    assert torch.isfinite(out).all() and float(out.abs().max()) == 0.0   # silence stays silence (no NaN from log / division)
    # more clips than max_clips: processed in groups, each clip as if it were alone
    rng = np.random.default_rng(1)
    noisy = torch.from_numpy(rng.standard_normal((3, 48000)).astype(np.float32) * 0.05).cuda()
    n = [48000, 20000, 3000]
    many = g.reduce_batch(noisy, n, prop_decrease=0.5)
    for b in range(3):
        alone = g.reduce_batch(noisy[b:b + 1].contiguous(), n[b:b + 1], prop_decrease=0.5)
        assert torch.equal(many[b], alone[0])
    with pytest.raises(_lib.CcxError):
        g.reduce_batch(torch.zeros(1, 50000, device="cuda"), [50000], prop_decrease=0.5)         # longer than max_samples
    with pytest.raises(_lib.CcxError):
        g(np.zeros(16000, np.float32), sr=8000)                                                  # wrong sample rate
