"""Sanity properties of the CPU restatements that have NO reference fixture (spectral gate, PyanNet / x-vector,
RE-SepFormer): they stay "parity unpinned" (DESIGN.md section 3), but the GPU parity tests lean on them, so their
structural invariants are checked here: frame counts, identity / monotonic behaviour of the knobs, batch independence,
determinism.  Small sizes: the whole file runs in a few seconds on the CPU."""
import numpy as np
import torch

from clearconverse_amd.audio import synthetic_clip
from clearconverse_amd.weights import SepDims, synthetic_pyannet_state_dict, synthetic_sepformer_state_dict, synthetic_xvector_state_dict
from oracle import pyannote_ref as P
from oracle import sepformer_ref as S
from oracle import spectral_gate_ref as G


def test_spectral_gate_knob_limits():
    x = synthetic_clip(0, 2.0)
    assert G.reduce_noise(x, 16000, prop_decrease=0.5).shape == x.shape
    # prop_decrease = 0: the mask is all ones except where its smoothing runs off the lowest / highest bins -> nearly the input
    same = G.reduce_noise(x, 16000, prop_decrease=0.0)
    assert float(np.dot(same, x) / (np.linalg.norm(same) * np.linalg.norm(x))) > 0.995
    # more suppression never adds energy
    e = [float(np.square(G.reduce_noise(x, 16000, prop_decrease=p)).sum()) for p in (0.0, 0.5, 1.0)]
    assert e[0] >= e[1] >= e[2] > 0.0


def test_spectral_gate_multi_chunk_equals_single_chunk_away_from_the_seams(monkeypatch):
    """The chunked path (ONE threshold, chunks with real-neighbour padding): with a small chunk size and the threshold taken from the
    whole signal (clip_noise_stationary=False) the result must equal the one-chunk result except near chunk seams, where each chunk's
    own dB floor / mask smoothing ends.  With the noise clip cut to the first chunk (the default) the threshold -- and the result --
    differ, which is what the switch is for."""
    x = synthetic_clip(3, 6.0)
    one = G.reduce_noise(x, 16000, prop_decrease=0.8)
    monkeypatch.setattr(G, "CHUNK", 40000)
    monkeypatch.setattr(G, "PADDING", 4000)
    many = G.reduce_noise(x, 16000, prop_decrease=0.8, clip_noise_stationary=False)
    clipped = G.reduce_noise(x, 16000, prop_decrease=0.8, clip_noise_stationary=True)
    assert float(np.linalg.norm(clipped - many) / np.linalg.norm(many)) > 1e-4
    assert many.shape == one.shape
    rel = float(np.linalg.norm(many - one) / np.linalg.norm(one))
    assert rel < 0.05, rel              # same threshold; only the per-chunk dB floor (max - 80) can differ


def test_pyannet_frame_counts_and_determinism():
    sd = dict(synthetic_pyannet_state_dict(7, seed=3)); sd["powerset"] = torch.tensor(1)
    clip = synthetic_clip(1, 10.0)
    for n, frames in ((160000, 589), (80000, 293)):           # the 10 s / 5 s windows of the two pipelines
        out = P.pyannet_forward(sd, torch.from_numpy(clip[:n].copy())[None, None])[0]
        assert out.shape == (frames, 7)
        assert torch.allclose(out.exp().sum(-1), torch.ones(frames), atol=1e-5)      # powerset head: log-softmax
    a = P.pyannet_forward(sd, torch.from_numpy(clip[:20000].copy())[None, None])[0]
    b = P.pyannet_forward(sd, torch.from_numpy(clip[:20000].copy())[None, None])[0]
    assert torch.equal(a, b)


def test_sinc_filters_are_symmetric_band_passes():
    sd = synthetic_xvector_state_dict(1)
    f = P.sinc_filters(sd["sincnet.conv1d.0.filterbank.low_hz_"], sd["sincnet.conv1d.0.filterbank.band_hz_"])
    assert f.shape == (80, 251)
    assert torch.allclose(f.abs(), torch.flip(f, dims=[1]).abs(), atol=1e-6)         # even (cos) and odd (sin) filters: linear phase


def test_xvector_is_scale_invariant():
    sd = synthetic_xvector_state_dict(2)
    x = torch.from_numpy(synthetic_clip(2, 1.0))[None]
    e1, e2 = P.xvector_forward(sd, x), P.xvector_forward(sd, 0.25 * x)
    assert e1.shape[-1] == 512
    assert float((e1 - e2).abs().max()) < 1e-3 * float(e1.abs().max())              # waveform instance norm


def test_sepformer_oracle_shapes_and_batch_independence():
    dims = SepDims(n_layers=1, n_blocks=1)
    sd = synthetic_sepformer_state_dict(dims, seed=4)
    ref = S.SepformerRef(S.SepDims(**dims.__dict__), sd)
    g = torch.Generator().manual_seed(0)
    mix = torch.randn(2, 3000, generator=g) * 0.1
    both = ref.separate(mix)
    assert both.shape == (2, 3000, 2) and torch.isfinite(both).all()
    one = ref.separate(mix[1:2])
    assert torch.allclose(both[1], one[0], atol=1e-5)
