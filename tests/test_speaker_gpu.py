"""-m gpu: SincNet x-vector embedder and PyanNet segmentation through the C ABI vs
oracle/pyannote_ref.py (CPU fp32).  Tolerances: embeddings rel-L2 <= 2e-2 (bf16 GEMM inputs, fp32
sinc conv / norms / LSTM cell state; the LSTM's h and W_hh enter the matrix cores as bf16); segmentation scores abs 5e-2 in log-prob, identical frame argmax
wherever the oracle's top-2 margin exceeds 0.1."""
import numpy as np
import pytest
import torch

from tests.conftest import within

from clearconverse_amd.audio import synthetic_clip
from clearconverse_amd.weights import synthetic_pyannet_state_dict, synthetic_xvector_state_dict
from oracle import pyannote_ref as P

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = a.double().flatten(); b = b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _crops(lengths):
    clip = synthetic_clip(3, 30.0)
    out, pos = [], 1000
    for n in lengths:
        out.append(torch.from_numpy(clip[pos:pos + n].copy()))
        pos += 7919
    return out


def test_sinc_filters_match_oracle():
    from clearconverse_amd.weights import sinc_filters
    sd = synthetic_xvector_state_dict(1)
    a = sinc_filters(sd["sincnet.conv1d.0.filterbank.low_hz_"], sd["sincnet.conv1d.0.filterbank.band_hz_"])
    b = P.sinc_filters(sd["sincnet.conv1d.0.filterbank.low_hz_"], sd["sincnet.conv1d.0.filterbank.band_hz_"])
    assert a.shape == (80, 251) and float((a - b).abs().max()) < 1e-6


def test_xvector_matches_oracle_ragged(ccx_ctx):
    from clearconverse_amd.speaker import XVectorEmbedder
    sd = synthetic_xvector_state_dict(seed=2)
    m = XVectorEmbedder(sd, max_crops=8, max_samples=16000 * 40, ctx=ccx_ctx)
    try:
        crops = _crops([12800, 8000, 144000, 31234])      # 0.8 s window, 0.5 s minimum, a 9 s segment, odd length
        got = m.embed_batch(crops).cpu()
        for i, c in enumerate(crops):
            ref = P.xvector_forward(sd, c[None])
            assert torch.isfinite(got[i]).all()
            within("x-vector: embedding rel-L2", _rel(got[i], ref), 7e-3, i)
            cos = torch.nn.functional.cosine_similarity(got[i], ref, dim=0).item()
            assert cos > 0.9995
        # reference call shape (back/api.py:869-872)
        one = m({"waveform": crops[0][None], "sample_rate": 16000})
        assert isinstance(one, np.ndarray) and one.shape == (512,)
        assert np.allclose(one, got[0].numpy(), atol=0, rtol=0)     # batch mates do not change a crop's result
    finally:
        m.close()


def test_xvector_weighted_pooling(ccx_ctx):
    from clearconverse_amd.speaker import XVectorEmbedder
    sd = synthetic_xvector_state_dict(seed=5)
    m = XVectorEmbedder(sd, max_crops=4, max_samples=16000 * 25, ctx=ccx_ctx)
    try:
        crops = _crops([160000, 160000])
        g = torch.Generator().manual_seed(0)
        w = [(torch.rand(589, generator=g) > 0.5).float(), torch.rand(293, generator=g)]
        got = m.embed_batch(crops, weights=w).cpu()
        for i in range(2):
            ref = P.xvector_forward(sd, crops[i][None], weights=w[i])
            assert _rel(got[i], ref) < 2e-2, (i, _rel(got[i], ref))
    finally:
        m.close()


def test_xvector_rejects_short_crop(ccx_ctx):
    from clearconverse_amd._lib import CcxError
    from clearconverse_amd.speaker import XVectorEmbedder
    m = XVectorEmbedder(synthetic_xvector_state_dict(0), max_crops=2, max_samples=16000 * 4, ctx=ccx_ctx)
    try:
        with pytest.raises(CcxError):
            m.embed_batch([torch.zeros(3000)])
    finally:
        m.close()


@pytest.mark.parametrize("powerset,n_classes", [(True, 7), (False, 3)])
def test_pyannet_matches_oracle(ccx_ctx, powerset, n_classes):
    from clearconverse_amd.speaker import SegmentationNet
    sd = synthetic_pyannet_state_dict(n_classes, seed=3)
    m = SegmentationNet(sd, n_classes=n_classes, powerset=powerset, max_crops=4, max_samples=16000 * 30, ctx=ccx_ctx)
    try:
        crops = _crops([160000, 80000, 20000])            # a 10 s chunk (589 frames), a 5 s chunk, a short tail
        outs = m.segment_batch(crops)
        osd = dict(sd); osd["powerset"] = torch.tensor(1 if powerset else 0)
        for c, got in zip(crops, outs):
            ref = P.pyannet_forward(osd, c[None, None])[0]
            got = got.cpu()
            assert got.shape == ref.shape, (got.shape, ref.shape)
            within(f"pyannet ({n_classes} classes): frame score max abs error (seeded weights)", float((got - ref).abs().max()), 2.5e-3)
            top2 = torch.topk(ref, 2, dim=-1).values
            decided = (top2[:, 0] - top2[:, 1]) > 0.1
            assert torch.equal(got.argmax(-1)[decided], ref.argmax(-1)[decided])
        assert outs[0].shape[0] == 589
    finally:
        m.close()


def test_pyannet_block_mates_and_ragged_groups(ccx_ctx):
    """The LSTM advances 16 windows per block on the matrix cores: a window's scores must not depend on which windows
    share its block, on its row inside the block, or on ragged lengths in the same block (37 windows = 2 full blocks +
    a partial one per direction)."""
    from clearconverse_amd.speaker import SegmentationNet
    sd = synthetic_pyannet_state_dict(7, seed=5)
    m = SegmentationNet(sd, n_classes=7, powerset=True, max_crops=64, max_samples=16000 * 400, ctx=ccx_ctx)
    try:
        lens = [80000, 160000, 20000, 48000, 80000, 33000, 160000] * 5 + [80000, 12000]
        crops = _crops(lens)
        together = m.segment_numpy(crops)
        assert [t.shape[0] for t in together] == [m.segment_numpy([c])[0].shape[0] for c in crops[:3]] + [t.shape[0] for t in together[3:]]
        for i in (0, 1, 2, 5, 15, 16, 17, 31, 35, 36):
            alone = m.segment_numpy([crops[i]])[0]
            assert np.array_equal(alone, together[i]), i
        # a different order puts every window on another accumulator row
        perm = list(reversed(range(len(crops))))
        shuffled = m.segment_numpy([crops[i] for i in perm])
        for k, i in enumerate(perm):
            assert np.array_equal(shuffled[k], together[i]), i
        # oracle check of a window that sits in the partial block
        osd = dict(sd); osd["powerset"] = torch.tensor(1)
        ref = P.pyannet_forward(osd, crops[36][None, None])[0]
        assert float((torch.from_numpy(together[36]) - ref).abs().max()) < 5e-2
    finally:
        m.close()
