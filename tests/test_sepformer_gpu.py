"""-m gpu: RE-SepFormer through the C ABI vs oracle/sepformer_ref.py (CPU fp32).
Tolerance: separated waveforms rel-L2 <= 8e-3 (9e-3 at full depth; measured 3.9e-3 / 4.4e-3 on MI355X, profiles/r03_measured_deviations.json; was 3e-2 / 5e-2) (bf16 GEMM inputs / bf16 attention probabilities through
2 x 8 + 8 transformer layers; fp32 residual stream and norms) -- SURVEY.md section 8c states 1e-2 for
single bf16 stages."""
import numpy as np
import pytest
import torch

from tests.conftest import within

from clearconverse_amd.weights import SepDims, synthetic_sepformer_state_dict
from oracle import sepformer_ref as S

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = a.double().flatten(); b = b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _mix(lengths, seed=0):
    g = torch.Generator().manual_seed(seed)
    T = max(lengths)
    x = torch.zeros(len(lengths), T)
    for i, n in enumerate(lengths):
        t = torch.arange(n) / 8000.0
        x[i, :n] = 0.3 * torch.sin(2 * np.pi * (180 + 40 * i) * t) + 0.2 * torch.sin(2 * np.pi * (310 + 25 * i) * t + 1.0) \
            + 0.02 * torch.randn(n, generator=g)
    return x


@pytest.fixture(scope="module")
def small(ccx_ctx):
    from clearconverse_amd.separator import SepformerSeparator
    dims = SepDims(n_layers=2)
    sd = synthetic_sepformer_state_dict(dims, seed=4)
    m = SepformerSeparator(dims, sd, max_tokens=20000, max_utts=8, ctx=ccx_ctx)
    yield dims, sd, m
    m.close()


def test_separate_matches_oracle_ragged(small):
    dims, sd, m = small
    lengths = [8000, 5213, 1216, 16 + 8 * 149]      # incl. L == 150 exactly (a full extra zero chunk) and a short clip
    mix = _mix(lengths)
    got = m.separate_batch(mix, lengths).cpu()
    orc = S.SepformerRef(S.SepDims(**dims.__dict__), sd)
    for i, n in enumerate(lengths):
        ref = orc.separate(mix[i:i + 1, :n])[0]
        assert torch.isfinite(got[i]).all()
        within("sepformer 2-layer: separated waveform rel-L2 (ragged batch)", _rel(got[i, :n], ref), 8e-3, i)
        assert float(got[i, n:].abs().max()) == 0.0 if n < mix.shape[1] else True


def test_batch_rows_are_independent(small):
    dims, sd, m = small
    mix = _mix([4000, 4000])
    both = m.separate_batch(mix).cpu()
    one = m.separate_batch(mix[1:2]).cpu()
    assert torch.equal(both[1], one[0])      # same kernels, same tiles -> bit-identical regardless of batch mates


def test_full_depth_model(ccx_ctx):
    from clearconverse_amd.separator import SepformerSeparator
    dims = SepDims()                          # 8 layers x (2 segment + 1 memory) blocks: the resepformer-wsj02mix geometry
    sd = synthetic_sepformer_state_dict(dims, seed=9)
    m = SepformerSeparator(dims, sd, max_tokens=4000, max_utts=2, ctx=ccx_ctx)
    try:
        mix = _mix([12000])
        got = m.separate_batch(mix).cpu()
        ref = S.SepformerRef(S.SepDims(**dims.__dict__), sd).separate(mix)
        assert got.shape == (1, 12000, 2)
        within("sepformer FULL depth (8 layers x 3 blocks): separated waveform rel-L2", _rel(got, ref), 9e-3)
    finally:
        m.close()


def test_baseline_config3_size_rows_independent(ccx_ctx):
    """BASELINE configs[2]: 16 x 8 s 8 kHz mixtures ([16, 64000] -> [16, 64000, 2]) at full depth.  The oracle takes minutes
    at this size, so the check is the size-independent property: every row equals the same utterance separated alone
    (ragged lengths included), rows past an utterance's length are zero, everything is finite."""
    from clearconverse_amd.separator import SepformerSeparator
    dims = SepDims()
    sd = synthetic_sepformer_state_dict(dims, seed=9)
    m = SepformerSeparator(dims, sd, max_tokens=16 * 8100, max_utts=16, ctx=ccx_ctx)
    try:
        lens = [64000] * 12 + [48000, 32000, 8000, 640]
        mix = _mix([64000] * 16).cuda()
        out = m.separate_batch(mix, lens)
        assert out.shape == (16, 64000, 2) and torch.isfinite(out).all()
        for b in (0, 7, 12, 15):
            alone = m.separate_batch(mix[b:b + 1, :lens[b]].contiguous(), [lens[b]])
            assert torch.equal(out[b, :lens[b]], alone[0])
            if lens[b] < 64000:
                assert float(out[b, lens[b]:].abs().max()) == 0.0
    finally:
        m.close()


def test_rejects_too_short_input(small):
    dims, sd, m = small
    from clearconverse_amd._lib import CcxError
    with pytest.raises(CcxError):
        m.separate_batch(torch.zeros(1, 8))


@pytest.mark.parametrize("d_ffn", [128, 384])
def test_fused_ffn_other_widths(ccx_ctx, d_ffn):
    """The fused LayerNorm + FFN kernel streams W1 / W2 in stages of 128 hidden units over two LDS buffers: one stage and
    an odd number of stages (the default width is 8 stages) against the oracle."""
    from clearconverse_amd.separator import SepformerSeparator
    dims = SepDims(n_layers=1, d_ffn=d_ffn)
    sd = synthetic_sepformer_state_dict(dims, seed=9)
    m = SepformerSeparator(dims, sd, max_tokens=20000, max_utts=4, ctx=ccx_ctx)
    try:
        lengths = [6000, 3111]
        mix = _mix(lengths)
        got = m.separate_batch(mix, lengths).cpu()
        orc = S.SepformerRef(S.SepDims(**dims.__dict__), sd)
        for i, n in enumerate(lengths):
            ref = orc.separate(mix[i:i + 1, :n])[0]
            assert _rel(got[i, :n], ref) < 3e-2, (i, _rel(got[i, :n], ref))
    finally:
        m.close()


def test_fused_attention_block_matches_the_separate_kernels(small, monkeypatch):
    """Sequences of up to 160 tokens run the attention half of a layer as ONE kernel (sep_attn_block_kernel); longer ones
    (memory sequences of utterances above ~24 s) go through LayerNorm / QKV GEMM / sep_attention_kernel / out-proj GEMM.
    Same math, different summation orders and one bf16 rounding less: the separated waveforms agree to 1e-2 of their norm,
    and both stay within the oracle tolerance."""
    dims, sd, m = small
    lengths = [8000, 5213, 1216]
    mix = _mix(lengths, seed=3)
    fused = m.separate_batch(mix, lengths).cpu()
    monkeypatch.setenv("CCX_SEP_FUSED_ATTN", "0")
    split = m.separate_batch(mix, lengths).cpu()
    monkeypatch.delenv("CCX_SEP_FUSED_ATTN")
    assert not torch.equal(fused, split)          # the switch did select another path
    orc = S.SepformerRef(S.SepDims(**dims.__dict__), sd)
    for i, n in enumerate(lengths):
        assert _rel(fused[i, :n], split[i, :n]) < 1e-2, (i, _rel(fused[i, :n], split[i, :n]))
        ref = orc.separate(mix[i:i + 1, :n])[0]
        assert _rel(split[i, :n], ref) < 3e-2 and _rel(fused[i, :n], ref) < 3e-2


def test_long_utterance_takes_the_separate_kernels(ccx_ctx):
    """An utterance of 26 s has 174 chunks: its memory sequences exceed the fused kernel's 160 tokens, its chunks do not."""
    from clearconverse_amd.separator import SepformerSeparator
    dims = SepDims(n_layers=1)
    sd = synthetic_sepformer_state_dict(dims, seed=6)
    m = SepformerSeparator(dims, sd, max_tokens=30000, max_utts=2, ctx=ccx_ctx)
    n = 26 * 8000
    mix = _mix([n], seed=5)
    got = m.separate_batch(mix, [n]).cpu()
    ref = S.SepformerRef(S.SepDims(**dims.__dict__), sd).separate(mix[:, :n])[0]
    m.close()
    assert torch.isfinite(got).all()
    assert _rel(got[0, :n], ref) < 3e-2, _rel(got[0, :n], ref)
