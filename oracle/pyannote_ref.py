"""CPU restatement (plain PyTorch fp32) of the pyannote.audio models the reference drives.

TEST INFRASTRUCTURE ONLY (tests/, smoke, bench cpu_baseline).

Restates, from recollection of the published pyannote.audio 3.x sources [UPSTREAM-RECALL] (the
package is an un-pinned dependency, /root/reference/back/requirements.txt:12-19, not vendored, not
installed here):
  * models/blocks/sincnet.py::SincNet (+ asteroid_filterbanks ParamSincFB)        -- shared front end
  * models/embedding/xvector.py::XVectorSincNet  = `pyannote/embedding`             -- reference
    back/api.py:776-780 (Inference(window="whole")), called at back/api.py:869
  * models/segmentation/PyanNet.py::PyanNet                                          -- inside the
    VAD / diarization pipelines, reference back/api.py:782-792, called at 1311, 1056, 1124
  * models/blocks/pooling.py::StatsPool (mean || unbiased std)

PARITY STATUS: **parity unpinned** (no fixture in the reference, no independent implementation in
this image).  Items marked (?) in SURVEY.md Appendix A.3 are decided here and listed in DESIGN.md.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

SINC_KERNEL = 251
SINC_STRIDE = 10
N_SINC = 80


def sinc_filters(low_hz_: torch.Tensor, band_hz_: torch.Tensor, sample_rate: int = 16000, min_low_hz: float = 50.0,
                 min_band_hz: float = 50.0, kernel: int = SINC_KERNEL) -> torch.Tensor:
    """asteroid_filterbanks ParamSincFB.filters(): 40 cos + 40 sin band-pass filters [80, kernel]."""
    low_hz_ = low_hz_.double().view(-1, 1)
    band_hz_ = band_hz_.double().view(-1, 1)
    half = kernel // 2
    n_lin = torch.linspace(0, kernel / 2 - 1, steps=half, dtype=torch.float64)
    window = 0.54 - 0.46 * torch.cos(2 * math.pi * n_lin / kernel)
    n_ = 2 * math.pi * torch.arange(-half, 0, dtype=torch.float64).view(1, -1) / sample_rate
    low = min_low_hz + torch.abs(low_hz_)
    high = torch.clamp(low + min_band_hz + torch.abs(band_hz_), min_low_hz, sample_rate / 2)
    band = (high - low)[:, 0]
    ft_low, ft_high = low @ n_, high @ n_
    out = []
    for kind in ("cos", "sin"):
        if kind == "cos":
            left = ((torch.sin(ft_high) - torch.sin(ft_low)) / (n_ / 2)) * window
            center = 2 * band.view(-1, 1)
            right = torch.flip(left, dims=[1])
        else:
            left = ((torch.cos(ft_low) - torch.cos(ft_high)) / (n_ / 2)) * window
            center = torch.zeros_like(band.view(-1, 1))
            right = -torch.flip(left, dims=[1])
        bp = torch.cat([left, center, right], dim=1) / (2 * band[:, None])
        out.append(bp)
    return torch.cat(out, dim=0).float()


def mel_init_sinc_params(n_filters: int = N_SINC, sample_rate: int = 16000, min_low_hz: float = 50.0, min_band_hz: float = 50.0):
    """ParamSincFB._initialize_filters(): cut-offs equally spaced on the mel scale."""
    to_mel = lambda hz: 2595 * np.log10(1 + hz / 700)
    to_hz = lambda mel: 700 * (10 ** (mel / 2595) - 1)
    mel = np.linspace(to_mel(30.0), to_mel(sample_rate / 2 - (min_low_hz + min_band_hz)), n_filters // 2 + 1)
    hz = to_hz(mel)
    return torch.tensor(hz[:-1], dtype=torch.float32).view(-1, 1), torch.tensor(np.diff(hz), dtype=torch.float32).view(-1, 1)


def _instance_norm(x, w, b, eps=1e-5):
    """nn.InstanceNorm1d(affine=True) on [B, C, T]: per (item, channel) biased statistics over time."""
    mean = x.mean(dim=-1, keepdim=True)
    var = x.var(dim=-1, unbiased=False, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * w.view(1, -1, 1) + b.view(1, -1, 1)


def sincnet_forward(sd: Dict[str, torch.Tensor], wav: torch.Tensor, prefix: str = "sincnet.") -> torch.Tensor:
    """SincNet.forward: wav [B, 1, T] -> [B, 60, frames]."""
    g = lambda k: sd[prefix + k].float()
    x = _instance_norm(wav.float(), g("wav_norm1d.weight"), g("wav_norm1d.bias"))
    filt = sinc_filters(g("conv1d.0.filterbank.low_hz_"), g("conv1d.0.filterbank.band_hz_"))
    x = F.conv1d(x, filt[:, None, :], stride=SINC_STRIDE)
    x = torch.abs(x)
    x = F.leaky_relu(_instance_norm(F.max_pool1d(x, 3, stride=3), g("norm1d.0.weight"), g("norm1d.0.bias")))
    x = F.conv1d(x, g("conv1d.1.weight"), g("conv1d.1.bias"))
    x = F.leaky_relu(_instance_norm(F.max_pool1d(x, 3, stride=3), g("norm1d.1.weight"), g("norm1d.1.bias")))
    x = F.conv1d(x, g("conv1d.2.weight"), g("conv1d.2.bias"))
    x = F.leaky_relu(_instance_norm(F.max_pool1d(x, 3, stride=3), g("norm1d.2.weight"), g("norm1d.2.bias")))
    return x


TDNN_OUT = [512, 512, 512, 512, 1500]
TDNN_K = [5, 3, 3, 1, 1]
TDNN_DIL = [1, 2, 3, 1, 1]


def xvector_forward(sd: Dict[str, torch.Tensor], wav: torch.Tensor, weights: Optional[torch.Tensor] = None) -> torch.Tensor:
    """XVectorSincNet.forward for ONE crop (Inference(window="whole")): wav [1, T] -> [512].
    weights [n_w] (optional): StatsPool frame weights, nearest-interpolated to the pooled frames."""
    x = sincnet_forward(sd, wav[None])
    for i, (k, dil) in enumerate(zip(TDNN_K, TDNN_DIL)):
        x = F.conv1d(x, sd[f"tdnns.{i}.0.weight"].float(), sd[f"tdnns.{i}.0.bias"].float(), dilation=dil)
        x = F.leaky_relu(x)
        x = F.batch_norm(x, sd[f"tdnns.{i}.2.running_mean"].float(), sd[f"tdnns.{i}.2.running_var"].float(),
                         sd[f"tdnns.{i}.2.weight"].float(), sd[f"tdnns.{i}.2.bias"].float(), training=False, eps=1e-5)
    if weights is None:
        mean = x.mean(dim=-1)
        std = x.std(dim=-1, unbiased=True)
    else:
        w = F.interpolate(weights.float().view(1, 1, -1), size=x.shape[-1], mode="nearest")   # [1,1,frames]
        v1 = w.sum(dim=2) + 1e-8
        mean = torch.sum(x * w, dim=2) / v1
        v2 = torch.square(w).sum(dim=2)
        var = torch.sum(torch.square(x - mean.unsqueeze(2)) * w, dim=2) / (v1 - v2 / v1 + 1e-8)
        std = torch.sqrt(var)
    pooled = torch.cat([mean, std], dim=-1)
    return F.linear(pooled, sd["embedding.weight"].float(), sd["embedding.bias"].float())[0]


def pyannet_forward(sd: Dict[str, torch.Tensor], wav: torch.Tensor, n_lstm: int = 4, hidden: int = 128,
                    return_features: bool = False) -> torch.Tensor:
    """PyanNet.forward: wav [B, 1, T] -> per-frame class scores [B, frames, C] (log-softmax for the
    powerset model, sigmoid for the multi-label one -- chosen by `sd['activation']`).  `return_features`: the input of the
    classifier [B, frames, 128] instead (used by tests/scripted_nets.py to fit a scripted classifier)."""
    x = sincnet_forward(sd, wav)                # [B, 60, F]
    x = x.permute(0, 2, 1)                      # [B, F, 60]
    B = x.shape[0]
    for layer in range(n_lstm):
        outs = []
        for direction, sfx in enumerate(("", "_reverse")):
            w_ih = sd[f"lstm.weight_ih_l{layer}{sfx}"].float()
            w_hh = sd[f"lstm.weight_hh_l{layer}{sfx}"].float()
            b = sd[f"lstm.bias_ih_l{layer}{sfx}"].float() + sd[f"lstm.bias_hh_l{layer}{sfx}"].float()
            seq = x if direction == 0 else torch.flip(x, dims=[1])
            h = torch.zeros(B, hidden)
            c = torch.zeros(B, hidden)
            hs = []
            gx = seq @ w_ih.T + b
            for t in range(seq.shape[1]):
                gates = gx[:, t] + h @ w_hh.T
                i, f, g_, o = gates.chunk(4, dim=-1)       # torch gate order: input, forget, cell, output
                c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g_)
                h = torch.sigmoid(o) * torch.tanh(c)
                hs.append(h)
            hseq = torch.stack(hs, dim=1)
            outs.append(hseq if direction == 0 else torch.flip(hseq, dims=[1]))
        x = torch.cat(outs, dim=-1)
    for i in range(2):
        x = F.leaky_relu(F.linear(x, sd[f"linear.{i}.weight"].float(), sd[f"linear.{i}.bias"].float()))
    if return_features:
        return x
    x = F.linear(x, sd["classifier.weight"].float(), sd["classifier.bias"].float())
    if int(sd.get("powerset", torch.tensor(1))) == 1:
        return F.log_softmax(x, dim=-1)
    return torch.sigmoid(x)
