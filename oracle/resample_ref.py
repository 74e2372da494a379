"""CPU restatement of `torchaudio.transforms.Resample(orig_freq, new_freq)(waveform)` with its default arguments
(resampling_method "sinc_interp_hann", lowpass_filter_width 6, rolloff 0.99), which the reference applies to every input that
is not 16 kHz (/root/reference/back/api.py:824-830).

TEST INFRASTRUCTURE ONLY.  torchaudio is an un-pinned dependency of the reference (back/requirements.txt: `torchaudio>=2.5.1`)
and is NOT installed in this image; this follows torchaudio/functional/functional.py::_get_sinc_resample_kernel and
_apply_sinc_resample_kernel from recollection [UPSTREAM-RECALL].  **Parity unpinned**: the reference holds no fixture for it.
tests/test_resample_cpu.py checks it against properties any band-limited resampler must have and against
scipy.signal.resample_poly (a different filter design, so agreement is to filter accuracy only)."""
from __future__ import annotations

import math

import torch


def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """_get_sinc_resample_kernel: (kernel [new, 1, taps] float32, width).  Computed in float64, cast to float32 (upstream does the
    same when no dtype is forced)."""
    g = math.gcd(int(orig_freq), int(new_freq))
    o, n = int(orig_freq) // g, int(new_freq) // g
    base_freq = min(o, n) * rolloff
    width = math.ceil(lowpass_filter_width * o / base_freq)
    idx = torch.arange(-width, width + o, dtype=torch.float64)[None, None] / o
    t = torch.arange(0, -n, -1, dtype=torch.float64)[:, None, None] / n + idx
    t *= base_freq
    t = t.clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t *= math.pi
    scale = base_freq / o
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t)
    kernels *= window * scale
    return kernels.to(torch.float32), width, o, n


def resample(waveform: torch.Tensor, orig_freq: int, new_freq: int) -> torch.Tensor:
    """waveform [..., time] float32 -> [..., ceil(new * time / orig)] (torchaudio.functional.resample)."""
    if int(orig_freq) == int(new_freq):
        return waveform
    kernel, width, o, n = sinc_resample_kernel(orig_freq, new_freq)
    shape = waveform.size()
    x = waveform.reshape(-1, shape[-1]).to(torch.float32)
    num_wavs, length = x.shape
    x = torch.nn.functional.pad(x, (width, width + o))
    y = torch.nn.functional.conv1d(x[:, None], kernel, stride=o)          # [num_wavs, n, frames]
    y = y.transpose(1, 2).reshape(num_wavs, -1)
    target_length = int(math.ceil(n * length / o))
    y = y[..., :target_length]
    return y.view(shape[:-1] + y.shape[-1:])
