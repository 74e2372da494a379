"""Host-side audio helpers of the product path: mel filterbank table, WAV I/O, synthetic clips.

`mel_filterbank` rebuilds the table openai-whisper ships as assets/mel_filters.npz
(librosa.filters.mel(sr=16000, n_fft=400, n_mels=80): slaney mel scale, slaney area norm)
[UPSTREAM-RECALL]; it is handed to libccx as the tensor "mel_filters".
"""
from __future__ import annotations

import wave
from typing import Tuple

import numpy as np

SAMPLE_RATE = 16000


def mel_filterbank(n_mels: int = 80, n_fft: int = 400, sr: int = SAMPLE_RATE) -> np.ndarray:
    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        lin = f / (200.0 / 3.0)
        log = 15.0 + np.log(np.maximum(f, 1e-10) / 1000.0) * (27.0 / np.log(6.4))
        return np.where(f >= 1000.0, log, lin)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        lin = m * (200.0 / 3.0)
        log = 1000.0 * np.exp((m - 15.0) * (np.log(6.4) / 27.0))
        return np.where(m >= 15.0, log, lin)

    freqs = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    pts = mel_to_hz(np.linspace(hz_to_mel(0.0), hz_to_mel(sr / 2.0), n_mels + 2))
    fb = np.zeros((n_mels, freqs.size), dtype=np.float64)
    for m in range(n_mels):
        lo, ce, hi = pts[m], pts[m + 1], pts[m + 2]
        up = (freqs - lo) / (ce - lo)
        down = (hi - freqs) / (hi - ce)
        fb[m] = np.maximum(0.0, np.minimum(up, down)) * (2.0 / (hi - lo))
    return fb.astype(np.float32)


def read_wav(path: str) -> Tuple[np.ndarray, int]:
    """PCM WAV -> (float32 [channels, n] in [-1, 1], sample_rate); the subset of
    torchaudio.load the reference relies on (back/api.py:820)."""
    with wave.open(path, "rb") as w:
        nch, sw, sr, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(n)
    if sw == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif sw == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif sw == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError(f"unsupported WAV sample width {sw}")
    return x.reshape(-1, nch).T.copy(), sr


def write_wav(path: str, audio: np.ndarray, sr: int = SAMPLE_RATE) -> None:
    """float [channels, n] or [n] -> 16-bit PCM WAV (torchaudio.save stand-in, back/api.py:1155)."""
    a = np.asarray(audio, dtype=np.float32)
    if a.ndim == 1:
        a = a[None]
    pcm = np.clip(np.round(a.T * 32767.0), -32768, 32767).astype("<i2")
    with wave.open(path, "wb") as w:
        w.setnchannels(a.shape[0])
        w.setsampwidth(2)
        w.setframerate(sr)
        w.writeframes(pcm.tobytes())


def sinc_resample_kernel(orig_sr: int, new_sr: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """Polyphase table of `torchaudio.transforms.Resample(orig_sr, new_sr)` with its default arguments (sinc_interp_hann)
    [UPSTREAM-RECALL], which the reference applies when a file is not 16 kHz (back/api.py:824-830).  Built in float64 and cast
    to float32 as upstream does.  Returns (kernT [taps, n] float32 -- TRANSPOSED for the device kernel --, width, o, n) with
    o / n the two rates over their gcd and taps = 2 * width + o.  A table, like the mel filterbank: all arithmetic on the audio
    runs in libccx (csrc/resample.hip)."""
    import math
    g = math.gcd(int(orig_sr), int(new_sr))
    o, n = int(orig_sr) // g, int(new_sr) // g
    base = min(o, n) * rolloff
    width = math.ceil(lowpass_filter_width * o / base)
    idx = np.arange(-width, width + o, dtype=np.float64)[None, :] / o
    t = (np.arange(0, -n, -1, dtype=np.float64)[:, None] / n + idx) * base
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    tp = t * math.pi
    kern = np.where(tp == 0, 1.0, np.sin(tp) / np.where(tp == 0, 1.0, tp)) * window * (base / o)   # [n, taps]
    return np.ascontiguousarray(kern.T.astype(np.float32)), width, o, n


class SincResampler:
    """Drop-in for the `torchaudio.transforms.Resample(orig_freq=, new_freq=)` object the reference keeps in `self.resampler`
    (back/api.py:824-830; it re-creates it when `.orig_freq` differs).  Calling it with a [channels, time] or [time] tensor
    returns the resampled tensor ON THE GPU; the arithmetic is `ccx_resample_sinc` (hand-written HIP, no CPU fallback)."""

    def __init__(self, orig_freq: int, new_freq: int = SAMPLE_RATE, device: int = 0, ctx=None):
        import torch
        from . import _lib
        if not torch.cuda.is_available():
            raise _lib.CcxError("SincResampler needs a ROCm GPU: the HIP path has no CPU fallback")
        self.orig_freq, self.new_freq = int(orig_freq), int(new_freq)
        self.ctx = ctx or _lib.Context(device)
        self.device = torch.device("cuda", device)
        kT, self.width, self.o, self.n = sinc_resample_kernel(self.orig_freq, self.new_freq)
        self.kernT = torch.from_numpy(kT).to(self.device)

    def __call__(self, signal):
        import math
        import torch
        from . import _lib
        x = torch.as_tensor(signal)
        squeeze = x.dim() == 1
        x = (x[None] if squeeze else x).to(self.device, torch.float32).contiguous()
        if self.orig_freq == self.new_freq:
            return x[0] if squeeze else x
        B, T = x.shape
        n_out = int(math.ceil(self.n * T / self.o))
        y = torch.empty(B, max(n_out, 1), device=self.device, dtype=torch.float32)
        ni = torch.full((B,), T, dtype=torch.int32, device=self.device)
        no = torch.full((B,), n_out, dtype=torch.int32, device=self.device)
        self.ctx.check(self.ctx.lib.ccx_resample_sinc(self.ctx.handle, x.data_ptr(), x.shape[1], ni.data_ptr(), B, self.o, self.n, self.width,
                                                      self.kernT.data_ptr(), y.data_ptr(), y.shape[1], no.data_ptr(), n_out,
                                                      _lib.current_stream_ptr()), "ccx_resample_sinc")
        y = y[:, :n_out]
        return y[0] if squeeze else y


# 30 s activity schedule of the synthetic benchmark clips
SCHEDULE_30S = [("A", 0.0, 9.0), ("B", 7.0, 16.0), ("A", 18.0, 24.0), ("B", 26.0, 30.0)]
SCHEDULE_10S = [("A", 0.0, 6.0), ("B", 4.0, 10.0)]


def synthetic_clip(index: int, seconds: float = 30.0, sr: int = SAMPLE_RATE) -> np.ndarray:
    """Two-"speaker" harmonic-complex clip, seed 1234+index: each speaker is
    12 harmonics of F0 = 110*2^(k/12) with 1/h roll-off, 4 Hz syllabic AM, -30 dB white noise;
    peak-normalised float32 mono."""
    rng = np.random.default_rng(1234 + index)
    n = int(round(seconds * sr))
    t = np.arange(n, dtype=np.float64) / sr
    sched = SCHEDULE_30S if seconds > 10.0 else SCHEDULE_10S
    scale = seconds / (30.0 if seconds > 10.0 else 10.0)
    out = np.zeros(n, dtype=np.float64)
    f0 = {}
    for spk in ("A", "B"):
        f0[spk] = 110.0 * 2.0 ** (int(rng.integers(0, 12)) / 12.0)
    if abs(f0["A"] - f0["B"]) < 1e-6:
        f0["B"] *= 2.0 ** (5 / 12.0)
    for spk, s, e in sched:
        s, e = s * scale, e * scale
        act = ((t >= s) & (t < e)).astype(np.float64)
        sig = np.zeros(n, dtype=np.float64)
        ph = rng.uniform(0, 2 * np.pi, size=12)
        for h in range(1, 13):
            sig += np.sin(2 * np.pi * f0[spk] * h * t + ph[h - 1]) / h
        am = 0.5 * (1.0 + np.sin(2 * np.pi * 4.0 * t + rng.uniform(0, 2 * np.pi)))
        out += act * sig * am
    out += 10 ** (-30 / 20.0) * rng.standard_normal(n)
    out /= np.max(np.abs(out)) + 1e-12
    return out.astype(np.float32)
