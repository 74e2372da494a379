"""CPU: oracle/resample_ref.py (restatement of torchaudio's sinc_interp_hann resampler, K1) -- parity unpinned: torchaudio is not
installed and the reference holds no fixture.  Checked here: the product's polyphase table equals the oracle's kernel, and the
oracle behaves like a band-limited resampler (tone in -> tone out, DC gain, length rule, agreement with scipy's polyphase
resampler to filter accuracy)."""
import math

import numpy as np
import pytest
import torch

from clearconverse_amd.audio import sinc_resample_kernel
from oracle import resample_ref as RS


@pytest.mark.parametrize("orig,new", [(44100, 16000), (8000, 16000), (48000, 16000), (22050, 16000), (11025, 16000), (32000, 16000)])
def test_product_table_equals_oracle_kernel(orig, new):
    kT, width, o, n = sinc_resample_kernel(orig, new)
    k, w2, o2, n2 = RS.sinc_resample_kernel(orig, new)
    assert (width, o, n) == (w2, o2, n2) and kT.shape == (2 * width + o, n)
    assert np.array_equal(kT, k[:, 0].numpy().T)             # bit-equal: same float64 formula, same cast


@pytest.mark.parametrize("orig", [8000, 44100, 48000])
def test_oracle_resamples_a_tone_and_keeps_dc(orig):
    t = np.arange(orig) / orig                                # 1 s
    x = torch.from_numpy((0.5 * np.sin(2 * np.pi * 440 * t) + 0.25).astype(np.float32))[None]
    y = RS.resample(x, orig, 16000)
    assert y.shape == (1, math.ceil(16000 * orig / orig))
    ref = 0.5 * np.sin(2 * np.pi * 440 * np.arange(16000) / 16000.0) + 0.25
    assert np.abs(y[0, 300:-300].numpy() - ref[300:-300]).max() < 3e-3
    assert RS.resample(x, 16000, 16000) is x


def test_oracle_length_rule_and_scipy_agreement():
    from scipy.signal import resample_poly
    rng = np.random.default_rng(0)
    n = 44100 + 123
    # band-limited test signal (well below both Nyquist rates): different anti-aliasing filters then agree closely
    t = np.arange(n) / 44100.0
    x = sum(np.sin(2 * np.pi * f * t + rng.uniform(0, 6.28)) / (1 + i) for i, f in enumerate((120.0, 555.0, 1830.0, 3100.0))).astype(np.float32)
    y = RS.resample(torch.from_numpy(x), 44100, 16000).numpy()
    assert y.shape[0] == math.ceil(160 * n / 441)
    z = resample_poly(x.astype(np.float64), 160, 441)
    m = min(len(y), len(z))
    err = np.abs(y[500:m - 500] - z[500:m - 500]).max() / np.abs(z).max()
    assert err < 5e-3, err
