"""CPU restatement (numpy / plain PyTorch fp32) of the WeSpeaker ResNet-34 speaker embedder (K21).

TEST INFRASTRUCTURE ONLY (tests/, smoke, bench cpu_baseline).

The reference loads `pyannote/speaker-diarization-3.1` (back/api.py:788-792) and calls it at
back/api.py:1056-1060 and 1124-1128; inside that pipeline the local speakers of every 10 s chunk are
embedded with `pyannote/wespeaker-voxceleb-resnet34-LM`.  Neither pyannote.audio nor wespeaker nor
torchaudio is vendored or installed (back/requirements.txt:12-19, un-pinned), so this file restates,
from recollection of the published sources [UPSTREAM-RECALL]:
  * torchaudio/compliance/kaldi.py::fbank (the arguments pyannote passes: num_mel_bins=80,
    frame_length=25, frame_shift=10, dither=0, window_type="hamming", use_energy=False, waveform
    scaled by 2**15), then per-chunk mean normalisation over frames
    (pyannote/audio/models/embedding/wespeaker/__init__.py::compute_fbank)
  * pyannote/audio/models/embedding/wespeaker/resnet.py::ResNet (BasicBlock [3,4,6,3], m_channels=32,
    feat_dim=80, embed_dim=256, TSTP pooling, two_emb_layer=False) with weighted statistics pooling
  * the pipeline embeds each (chunk, local speaker) pair with the speaker's frame activity as pooling
    weights; the convolutional trunk only sees the chunk waveform, so it is the same for the speakers
    of one chunk.

PARITY STATUS: **parity unpinned** (no fixture in the reference).  The Kaldi fbank is cross-checked against the
independently written kaldi mode of `transformers.audio_utils.spectrogram` (tests/test_oracle_wespeaker.py); the
ResNet-34 itself has no second implementation in this image.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

SAMPLE_RATE = 16000
FRAME_LEN = 400      # 25 ms
FRAME_SHIFT = 160    # 10 ms
N_FFT = 512          # round_to_power_of_two
N_MELS = 80
PREEMPH = 0.97
LOW_FREQ = 20.0
EPS = float(np.finfo(np.float32).eps)
LAYERS = (3, 4, 6, 3)
M_CHANNELS = 32
EMBED_DIM = 256


def num_frames(n_samples: int) -> int:
    """snip_edges=True."""
    return 0 if n_samples < FRAME_LEN else 1 + (n_samples - FRAME_LEN) // FRAME_SHIFT


def _mel(f):
    return 1127.0 * np.log(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def kaldi_mel_banks(num_bins: int = N_MELS, n_fft: int = N_FFT, sample_rate: int = SAMPLE_RATE, low_freq: float = LOW_FREQ,
                    high_freq: float = 0.0) -> np.ndarray:
    """kaldi.get_mel_banks: triangular filters, linear in the mel domain, over the n_fft/2 lowest bins;
    returned [num_bins, n_fft/2 + 1] with a zero column for the Nyquist bin."""
    nyquist = 0.5 * sample_rate
    if high_freq <= 0.0:
        high_freq += nyquist
    n_bins_fft = n_fft // 2
    fft_bin_width = sample_rate / n_fft
    mel_low, mel_high = _mel(low_freq), _mel(high_freq)
    delta = (mel_high - mel_low) / (num_bins + 1)
    b = np.arange(num_bins, dtype=np.float64)[:, None]
    left = mel_low + b * delta
    center = left + delta
    right = center + delta
    mel = _mel(fft_bin_width * np.arange(n_bins_fft, dtype=np.float64))[None, :]
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    banks = np.maximum(0.0, np.minimum(up, down))
    return np.pad(banks, ((0, 0), (0, 1))).astype(np.float32)


def hamming_window(n: int = FRAME_LEN) -> np.ndarray:
    """torch.hamming_window(n, periodic=False, alpha=0.54, beta=0.46)."""
    return (0.54 - 0.46 * np.cos(2.0 * math.pi * np.arange(n, dtype=np.float64) / (n - 1))).astype(np.float32)


def kaldi_fbank(wave: np.ndarray) -> np.ndarray:
    """wave [N] float32 in [-1, 1] -> log mel energies [T, 80] float32 (before mean normalisation)."""
    x = np.asarray(wave, dtype=np.float32) * np.float32(1 << 15)
    T = num_frames(x.shape[0])
    if T == 0:
        return np.zeros((0, N_MELS), np.float32)
    idx = np.arange(T)[:, None] * FRAME_SHIFT + np.arange(FRAME_LEN)[None, :]
    fr = x[idx].astype(np.float32)                                   # [T, 400]
    fr = fr - fr.mean(axis=1, keepdims=True, dtype=np.float32)       # remove_dc_offset
    prev = np.concatenate([fr[:, :1], fr[:, :-1]], axis=1)           # replicate padding on the left
    fr = fr - np.float32(PREEMPH) * prev
    fr = fr * hamming_window()[None, :]
    spec = np.fft.rfft(np.pad(fr, ((0, 0), (0, N_FFT - FRAME_LEN))).astype(np.float64), axis=1)
    power = (spec.real ** 2 + spec.imag ** 2).astype(np.float32)     # [T, 257]
    mel = power @ kaldi_mel_banks().T
    return np.log(np.maximum(mel, np.float32(EPS))).astype(np.float32)


def compute_fbank(wave: np.ndarray) -> np.ndarray:
    """pyannote compute_fbank: kaldi fbank, then subtract the mean over the chunk's frames."""
    f = kaldi_fbank(wave)
    return f - f.mean(axis=0, keepdims=True, dtype=np.float32) if f.shape[0] else f


def _bn(x, sd, name):
    return F.batch_norm(x, sd[name + ".running_mean"].float(), sd[name + ".running_var"].float(), sd[name + ".weight"].float(),
                        sd[name + ".bias"].float(), training=False, eps=1e-5)


def resnet_trunk(sd: Dict[str, torch.Tensor], feats: torch.Tensor, prefix: str = "resnet.") -> torch.Tensor:
    """feats [B, T, 80] -> frame-level features [B, 256 * 10, T'] (channel-major, then frequency)."""
    x = feats.float().permute(0, 2, 1).unsqueeze(1)                  # [B, 1, F, T]
    x = F.relu(_bn(F.conv2d(x, sd[prefix + "conv1.weight"].float(), padding=1), sd, prefix + "bn1"))
    for li, nblocks in enumerate(LAYERS, start=1):
        for bi in range(nblocks):
            p = f"{prefix}layer{li}.{bi}."
            stride = 2 if (li > 1 and bi == 0) else 1
            out = F.relu(_bn(F.conv2d(x, sd[p + "conv1.weight"].float(), stride=stride, padding=1), sd, p + "bn1"))
            out = _bn(F.conv2d(out, sd[p + "conv2.weight"].float(), padding=1), sd, p + "bn2")
            if (p + "shortcut.0.weight") in sd:
                sc = _bn(F.conv2d(x, sd[p + "shortcut.0.weight"].float(), stride=stride), sd, p + "shortcut.1")
            else:
                sc = x
            x = F.relu(out + sc)
    B, C, H, W = x.shape
    return x.reshape(B, C * H, W)


def stats_pool(x: torch.Tensor, weights: Optional[torch.Tensor]) -> torch.Tensor:
    """TSTP / pyannote StatsPool: x [B, D, T'], weights [B, n_w] or None -> [B, 2D] (mean || unbiased std)."""
    if weights is None:
        return torch.cat([x.mean(dim=-1), x.std(dim=-1, unbiased=True)], dim=-1)
    w = F.interpolate(weights.float().unsqueeze(1), size=x.shape[-1], mode="nearest")      # [B, 1, T']
    v1 = w.sum(dim=2) + 1e-8
    mean = torch.sum(x * w, dim=2) / v1
    v2 = torch.square(w).sum(dim=2)
    var = torch.sum(torch.square(x - mean.unsqueeze(2)) * w, dim=2) / (v1 - v2 / v1 + 1e-8)
    return torch.cat([mean, torch.sqrt(var)], dim=-1)


def resnet_embed(sd: Dict[str, torch.Tensor], waves: np.ndarray, weights: Optional[np.ndarray] = None,
                 mask_chunk: Optional[np.ndarray] = None, prefix: str = "resnet.") -> np.ndarray:
    """waves [n_chunks, N] (equal lengths) -> embeddings.
    Without weights: one embedding per chunk, [n_chunks, 256].  With weights [n_masks, n_w] and
    mask_chunk [n_masks] (the chunk each mask pools over): [n_masks, 256]."""
    feats = torch.from_numpy(np.stack([compute_fbank(w) for w in waves]))
    with torch.no_grad():
        x = resnet_trunk(sd, feats, prefix)
        if weights is None:
            pooled = stats_pool(x, None)
        else:
            sel = torch.as_tensor(np.asarray(mask_chunk), dtype=torch.long)
            pooled = stats_pool(x[sel], torch.from_numpy(np.asarray(weights, dtype=np.float32)))
        emb = F.linear(pooled, sd[prefix + "seg_1.weight"].float(), sd[prefix + "seg_1.bias"].float())
    return emb.numpy()
