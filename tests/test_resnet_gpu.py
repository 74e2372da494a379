"""GPU parity of the WeSpeaker ResNet-34 embedder (K21, csrc/resnet.hip) against oracle/wespeaker_ref.py,
through the C ABI (ccx_resnet_*).  Tolerance: rel-L2 <= 8e-3 (measured 3.6e-3) on the embedding (bf16 activations through
33 convolutions against the fp32 restatement; SURVEY.md 8c states 1e-2 per bf16 tensor)."""
import numpy as np
import pytest
import torch

from tests.conftest import within

from clearconverse_amd.audio import synthetic_clip
from clearconverse_amd.weights import synthetic_resnet34_state_dict
from oracle import wespeaker_ref as R

pytestmark = pytest.mark.gpu
TOL = 8e-3      # rel-L2 of the 256-d embedding; worst measured on MI355X 3.6e-3 (profiles/r03_measured_deviations.json)


def _clip(seed, n, off=0):
    return np.asarray(synthetic_clip(seed), dtype=np.float32)[off:off + n]


def _rel(a, b):
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-12))


@pytest.fixture(scope="module")
def net(ccx_ctx):
    from clearconverse_amd.speaker import ResNetEmbedder
    sd = synthetic_resnet34_state_dict(0)
    return sd, ResNetEmbedder(sd, max_chunks=8, max_samples=160000, max_masks=32, ctx=ccx_ctx)


def test_unweighted_embeddings_match_oracle(net):
    sd, emb = net
    waves = np.stack([_clip(11, 32000), _clip(12, 32000, 16000), _clip(13, 32000, 40000)])
    got = emb.embed_chunks(torch.from_numpy(waves)).cpu().numpy()
    want = R.resnet_embed(sd, waves)
    assert got.shape == want.shape == (3, 256)
    for i in range(3):
        within("resnet34: embedding rel-L2 (unweighted)", _rel(got[i], want[i]), TOL, i)


def test_masked_embeddings_share_the_trunk(net):
    sd, emb = net
    waves = np.stack([_clip(21, 48000), _clip(22, 48000, 8000)])
    rng = np.random.default_rng(0)
    w = (rng.random((5, 117)) > 0.4).astype(np.float32)
    w[2] = rng.random(117).astype(np.float32)              # soft weights
    mc = np.array([0, 0, 0, 1, 1])
    got = emb.embed_chunks(torch.from_numpy(waves), torch.from_numpy(w), mc).cpu().numpy()
    want = R.resnet_embed(sd, waves, w, mc)
    for j in range(5):
        within("resnet34: embedding rel-L2 (masked pooling)", _rel(got[j], want[j]), TOL, j)


def test_ten_second_chunk_and_geometry_change(net):
    sd, emb = net
    wave = _clip(31, 160000)[None]                          # the diarization pipeline's chunk length: 998 frames
    got = emb.embed_chunks(torch.from_numpy(wave)).cpu().numpy()
    want = R.resnet_embed(sd, wave)
    within("resnet34: embedding rel-L2 (10 s chunk)", _rel(got[0], want[0]), TOL)
    # a shorter call afterwards moves the halo cells: results must not depend on the previous geometry
    short = np.stack([_clip(32, 24000)])
    got2 = emb.embed_chunks(torch.from_numpy(short)).cpu().numpy()
    assert _rel(got2[0], R.resnet_embed(sd, short)[0]) < TOL


def test_results_do_not_depend_on_the_geometry_history(net, ccx_ctx):
    """Only the halo cells of the chunks in use are re-zeroed when the frame count changes or more chunks come into use
    (halo_zero_kernel): long -> short -> long with MORE chunks -> shorter with fewer must each be bit-identical to a fresh instance."""
    from clearconverse_amd.speaker import ResNetEmbedder
    sd, emb = net
    calls = [(160000, 2), (24000, 5), (160000, 6), (8000, 1), (24000, 8), (159000, 8), (160000, 1)]
    for k, (n, chunks) in enumerate(calls):
        waves = torch.from_numpy(np.stack([_clip(60 + k * 8 + i, n, 100 * i) for i in range(chunks)]))
        got = emb.embed_chunks(waves).cpu()
        fresh = ResNetEmbedder(sd, max_chunks=8, max_samples=160000, max_masks=32, ctx=ccx_ctx)
        want = fresh.embed_chunks(waves).cpu()
        assert torch.isfinite(got).all()
        assert torch.equal(got, want), (k, n, chunks)
        del fresh


def test_chunk_results_do_not_depend_on_batch_mates(net):
    """Size-independent property at the pipeline's chunk size (10 s): an embedding is bit-identical whatever else is in
    the launch (GEMM tiles straddle chunk boundaries, but every output element sees only its own chunk)."""
    sd, emb = net
    waves = np.stack([_clip(40 + i, 160000) for i in range(6)])
    t = torch.from_numpy(waves)
    a = emb.embed_chunks(t).cpu()
    b = emb.embed_chunks(t.flip(0)).cpu().flip(0)
    assert torch.equal(a, b)
    c = torch.cat([emb.embed_chunks(t[i:i + 1]).cpu() for i in range(6)])
    assert torch.equal(a, c)
    # masks: all-ones weights equal the unweighted statistics up to the 1e-8 regulariser of the weighted formula
    w = torch.ones(6, 589)
    d = emb.embed_chunks(t, w, list(range(6))).cpu()
    assert float((a - d).abs().max()) < 1e-3 * float(a.abs().max())


def test_capacity_and_argument_errors(net):
    from clearconverse_amd import _lib
    _, emb = net
    with pytest.raises(_lib.CcxError):
        emb.embed_chunks(torch.zeros(1, 161000))            # longer than max_samples
    with pytest.raises(_lib.CcxError):
        emb.embed_chunks(torch.zeros(1, 1000))              # too short for two pooled frames
    with pytest.raises(_lib.CcxError):
        emb.embed_chunks(torch.zeros(2, 32000), torch.ones(2, 10), [1, 0])   # unsorted mask_chunk
