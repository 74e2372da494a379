"""CPU restatement of `noisereduce.reduce_noise(y, sr, stationary=True, prop_decrease=p)` as the
reference calls it (/root/reference/back/api.py:349 on profile crops, 832-833 on the whole clip).

TEST INFRASTRUCTURE ONLY.  noisereduce is an un-pinned, un-vendored, un-installed dependency; this
follows its published SpectralGateStationary from recollection [UPSTREAM-RECALL] and uses the same
scipy.signal primitives it uses (scipy IS installed here).  **Parity unpinned** (no fixture exists).

Defaults restated: n_fft = win_length = 1024, hop 256, n_std_thresh_stationary = 1.5,
freq_mask_smooth_hz = 500, time_mask_smooth_ms = 50, chunk_size 600000, padding 30000, noise statistics
from the signal itself (y_noise=None), clip_noise_stationary=True.  Inputs longer than one chunk take the chunked path:
threshold from the whole signal, every chunk padded with 30000 samples of its real neighbours.  (Upstream always extends a chunk to
chunk_size + 2 * padding with zeros; the zeros after the final 30000 change nothing inside the kept region, so a short signal is
padded by 30000 only.)
"""
from __future__ import annotations

import numpy as np
from scipy.signal import fftconvolve, istft, stft

N_FFT = 1024
HOP = 256
PADDING = 30000
CHUNK = 600000


def _amp_to_db(x, top_db=80.0):
    x_db = 20 * np.log10(np.abs(x) + np.finfo(np.float64).eps)
    return np.maximum(x_db, np.max(x_db, axis=-1, keepdims=True) - top_db)


def smoothing_filter(sr: int):
    n_grad_freq = int(500 / (sr / (N_FFT / 2)))
    n_grad_time = int(50 / ((HOP / sr) * 1000))
    f = np.concatenate([np.linspace(0, 1, n_grad_freq + 1, endpoint=False), np.linspace(1, 0, n_grad_freq + 2)])[1:-1]
    t = np.concatenate([np.linspace(0, 1, n_grad_time + 1, endpoint=False), np.linspace(1, 0, n_grad_time + 2)])[1:-1]
    filt = np.outer(f, t)
    return filt / np.sum(filt), f / f.sum(), t / t.sum()


def _thresholds(y: np.ndarray, n_std: float):
    """Noise statistics from the signal itself (y_noise=None): per-bin mean + n_std * std of the dB spectrogram of ALL of y."""
    _, _, noise_stft = stft(y, nfft=N_FFT, noverlap=N_FFT - HOP, nperseg=N_FFT, padded=False)
    noise_db = _amp_to_db(noise_stft)
    return np.mean(noise_db, axis=1) + np.std(noise_db, axis=1) * n_std


def _filter_chunk(y: np.ndarray, start: int, end: int, thresh: np.ndarray, sr: int, prop_decrease: float) -> np.ndarray:
    """SpectralGate.filter_chunk: the chunk [start, end) read with PADDING samples of real context on either side (zeros beyond
    the ends of y), gated, the padding cut off again."""
    n = y.shape[0]
    i1, i2 = start - PADDING, end + PADDING
    padded = np.zeros(i2 - i1, dtype=np.float32)
    a, b = max(i1, 0), min(i2, n)
    padded[a - i1:b - i1] = y[a:b]
    _, _, sig_stft = stft(padded, nfft=N_FFT, noverlap=N_FFT - HOP, nperseg=N_FFT, padded=False)
    sig_db = _amp_to_db(sig_stft)
    mask = (sig_db > thresh[:, None]).astype(np.float64)
    mask = mask * prop_decrease + np.ones_like(mask) * (1.0 - prop_decrease)
    filt, _, _ = smoothing_filter(sr)
    mask = fftconvolve(mask, filt, mode="same")
    _, den = istft(sig_stft * mask, nfft=N_FFT, noverlap=N_FFT - HOP, nperseg=N_FFT)
    out = np.zeros(i2 - i1, dtype=np.float64)
    m = min(out.shape[0], den.shape[0])
    out[:m] = den[:m]
    return out[PADDING:PADDING + (end - start)]


def reduce_noise(y: np.ndarray, sr: int = 16000, prop_decrease: float = 1.0, n_std: float = 1.5,
                 clip_noise_stationary: bool = True) -> np.ndarray:
    """Any length: signals beyond CHUNK samples are filtered chunk by chunk (SpectralGate.get_traces / _iterate_chunk) against ONE
    threshold; a shorter signal is the single chunk [0, n).  The threshold comes from the noise clip = the signal itself
    (y_noise=None), which SpectralGateStationary.__init__ cuts to its first chunk_size samples when clip_noise_stationary is True
    (the package default) [UPSTREAM-RECALL; parity unpinned -- False takes the whole signal]."""
    y = np.asarray(y, dtype=np.float32).reshape(-1)
    n = y.shape[0]
    thresh = _thresholds(y[:CHUNK] if clip_noise_stationary else y, n_std)
    out = np.zeros(n, dtype=np.float32)
    for start in range(0, n, CHUNK):
        end = min(start + CHUNK, n)
        out[start:end] = _filter_chunk(y, start, end, thresh, sr, prop_decrease).astype(np.float32)
    return out
