"""Weight sources for the models of the hot path.

Real checkpoints are used when they exist under MODEL_CACHE_DIR in the layouts the reference's
loaders expect (back/api.py:657-703: `whisper/<size>.pt` = {"dims", "model_state_dict"},
`whisper-ft/model.safetensors` overlay).  There is no network in the build/bench environment, so
otherwise seeded synthetic weights of the same architecture are generated.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, asdict
from typing import Dict, Optional

import torch


@dataclass
class WhisperDims:
    n_mels: int = 80
    n_audio_ctx: int = 1500
    n_audio_state: int = 768
    n_audio_head: int = 12
    n_audio_layer: int = 12
    n_vocab: int = 51864
    n_text_ctx: int = 448
    n_text_state: int = 768
    n_text_head: int = 12
    n_text_layer: int = 12

    @staticmethod
    def small_en() -> "WhisperDims":
        return WhisperDims()

    @staticmethod
    def mini(n_layer: int = 2, n_state: int = 128) -> "WhisperDims":
        """Reduced-depth/width config for fast parity tests (same kernels, same code path)."""
        return WhisperDims(n_audio_state=n_state, n_audio_head=n_state // 64, n_audio_layer=n_layer,
                           n_text_state=n_state, n_text_head=n_state // 64, n_text_layer=n_layer)


def _sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> torch.Tensor:
    inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2, dtype=torch.float32))
    t = torch.arange(length, dtype=torch.float32)[:, None] * inv[None, :]
    return torch.cat([torch.sin(t), torch.cos(t)], dim=1)


def synthetic_whisper_state_dict(dims: WhisperDims, seed: int = 0, gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Seeded random weights in the openai-whisper key layout.  Linear/conv weights are
    N(0, gain/sqrt(fan_in)) so every sub-layer contributes O(1) to the residual stream (a
    N(0, 0.02) init would hide attention/MLP errors below any tolerance); LayerNorm affine
    parameters are perturbed around (1, 0)."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def lin(name, out_f, in_f, bias=True):
        sd[name + ".weight"] = torch.randn(out_f, in_f, generator=g) * (gain / math.sqrt(in_f))
        if bias:
            sd[name + ".bias"] = torch.randn(out_f, generator=g) * 0.1

    def ln(name, n):
        sd[name + ".weight"] = 1.0 + 0.1 * torch.randn(n, generator=g)
        sd[name + ".bias"] = 0.1 * torch.randn(n, generator=g)

    D = dims.n_audio_state
    sd["encoder.conv1.weight"] = torch.randn(D, dims.n_mels, 3, generator=g) * (gain / math.sqrt(3 * dims.n_mels))
    sd["encoder.conv1.bias"] = torch.randn(D, generator=g) * 0.1
    sd["encoder.conv2.weight"] = torch.randn(D, D, 3, generator=g) * (gain / math.sqrt(3 * D))
    sd["encoder.conv2.bias"] = torch.randn(D, generator=g) * 0.1
    sd["encoder.positional_embedding"] = _sinusoids(dims.n_audio_ctx, D)

    def block(p, n_state, cross):
        lin(p + ".attn.query", n_state, n_state)
        lin(p + ".attn.key", n_state, n_state, bias=False)
        lin(p + ".attn.value", n_state, n_state)
        lin(p + ".attn.out", n_state, n_state)
        ln(p + ".attn_ln", n_state)
        if cross:
            lin(p + ".cross_attn.query", n_state, n_state)
            lin(p + ".cross_attn.key", n_state, n_state, bias=False)
            lin(p + ".cross_attn.value", n_state, n_state)
            lin(p + ".cross_attn.out", n_state, n_state)
            ln(p + ".cross_attn_ln", n_state)
        lin(p + ".mlp.0", 4 * n_state, n_state)
        lin(p + ".mlp.2", n_state, 4 * n_state)
        ln(p + ".mlp_ln", n_state)

    for l in range(dims.n_audio_layer):
        block(f"encoder.blocks.{l}", D, False)
    ln("encoder.ln_post", D)
    T = dims.n_text_state
    sd["decoder.token_embedding.weight"] = torch.randn(dims.n_vocab, T, generator=g) * (gain / math.sqrt(T))
    sd["decoder.positional_embedding"] = torch.randn(dims.n_text_ctx, T, generator=g) * 0.02
    for l in range(dims.n_text_layer):
        block(f"decoder.blocks.{l}", T, True)
    ln("decoder.ln", T)
    return sd


def find_whisper_checkpoint(model_size: str = "small.en", cache_dir: Optional[str] = None):
    """Return (dims, state_dict) from `<cache_dir>/whisper/<size>.pt` (+ the `whisper-ft` overlay the
    reference applies, back/api.py:671-692) or None when no checkpoint is present."""
    cache_dir = cache_dir or os.environ.get("MODEL_CACHE_DIR", "models")
    path = os.path.join(cache_dir, "whisper", f"{model_size}.pt")
    if not os.path.exists(path):
        return None
    ck = torch.load(path, map_location="cpu", weights_only=True)
    dims = WhisperDims(**{k: int(v) for k, v in ck["dims"].items()})
    sd = {k: v.float() for k, v in ck["model_state_dict"].items()}
    ft = os.path.join(cache_dir, "whisper-ft")
    st_path = os.path.join(ft, "model.safetensors")
    pt_path = os.path.join(ft, "model.pt")
    if os.path.exists(st_path):
        from safetensors.torch import load_file
        over = load_file(st_path, device="cpu")
        sd.update({k: v.float() for k, v in over.items() if k in sd})  # strict=False semantics
    elif os.path.exists(pt_path):
        over = torch.load(pt_path, map_location="cpu", weights_only=True)
        sd.update({k: v.float() for k, v in over.items() if k in sd})
    return dims, sd


# ----------------------------------------------------------------------------------------------
# RE-SepFormer (speechbrain/resepformer-wsj02mix hyper-parameters [UPSTREAM-RECALL])
# ----------------------------------------------------------------------------------------------
@dataclass
class SepDims:
    n_filters: int = 128
    kernel: int = 16
    stride: int = 8
    d_model: int = 128
    n_head: int = 8
    d_ffn: int = 1024
    n_layers: int = 8
    n_blocks: int = 2
    segment: int = 150
    n_spk: int = 2


def synthetic_sepformer_state_dict(dims: SepDims, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded weights in the SpeechBrain checkpoint key layout (`encoder.ckpt`, `masknet.ckpt`,
    `decoder.ckpt` of the reference's overlay, back/api.py:729-746, each prefixed with its module
    name).  Fan-in scaled so every sub-layer contributes."""
    g = torch.Generator().manual_seed(seed)
    D, Fd = dims.d_model, dims.d_ffn
    sd: Dict[str, torch.Tensor] = {}
    sd["encoder.conv1d.weight"] = torch.randn(dims.n_filters, 1, dims.kernel, generator=g) / math.sqrt(dims.kernel) * 2.0
    sd["decoder.weight"] = torch.randn(dims.n_filters, 1, dims.kernel, generator=g) / math.sqrt(dims.n_filters)

    def block(prefix):
        for l in range(dims.n_layers):
            p = f"{prefix}.mdl.layers.{l}"
            sd[p + ".self_att.att.in_proj_weight"] = torch.randn(3 * D, D, generator=g) / math.sqrt(D)
            sd[p + ".self_att.att.in_proj_bias"] = 0.1 * torch.randn(3 * D, generator=g)
            sd[p + ".self_att.att.out_proj.weight"] = torch.randn(D, D, generator=g) / math.sqrt(D)
            sd[p + ".self_att.att.out_proj.bias"] = 0.1 * torch.randn(D, generator=g)
            sd[p + ".pos_ffn.ffn.0.weight"] = torch.randn(Fd, D, generator=g) / math.sqrt(D)
            sd[p + ".pos_ffn.ffn.0.bias"] = 0.1 * torch.randn(Fd, generator=g)
            sd[p + ".pos_ffn.ffn.3.weight"] = torch.randn(D, Fd, generator=g) / math.sqrt(Fd)
            sd[p + ".pos_ffn.ffn.3.bias"] = 0.1 * torch.randn(D, generator=g)
            for nm in ("norm1", "norm2"):
                sd[f"{p}.{nm}.norm.weight"] = 1 + 0.1 * torch.randn(D, generator=g)
                sd[f"{p}.{nm}.norm.bias"] = 0.1 * torch.randn(D, generator=g)
        sd[prefix + ".mdl.norm.norm.weight"] = 1 + 0.1 * torch.randn(D, generator=g)
        sd[prefix + ".mdl.norm.norm.bias"] = 0.1 * torch.randn(D, generator=g)
        sd[prefix + ".norm.weight"] = (1 + 0.1 * torch.randn(D, generator=g)).view(D, 1)
        sd[prefix + ".norm.bias"] = (0.1 * torch.randn(D, generator=g)).view(D, 1)

    for i in range(dims.n_blocks):
        block(f"masknet.model.seg_model.{i}")
        if i < dims.n_blocks - 1:
            block(f"masknet.model.mem_model.{i}")
    sd["masknet.model.output_fc.0.weight"] = torch.tensor([0.25])
    sd["masknet.model.output_fc.1.weight"] = torch.randn(dims.n_filters * dims.n_spk, D, 1, generator=g) / math.sqrt(D)
    sd["masknet.model.output_fc.1.bias"] = 0.1 * torch.randn(dims.n_filters * dims.n_spk, generator=g)
    return sd


# ----------------------------------------------------------------------------------------------
# pyannote-style speaker networks [UPSTREAM-RECALL]: SincNet front end, XVectorSincNet, PyanNet
# ----------------------------------------------------------------------------------------------
def find_sepformer_checkpoint(cache_dir: Optional[str] = None) -> Optional[Dict[str, torch.Tensor]]:
    """SpeechBrain savedir layout the reference uses (back/api.py:713-717): `<cache>/resepformer/{encoder,masknet,decoder}.ckpt`,
    each the state_dict of that module, then the fine-tune overlay `<cache>/resepformer-ft/` the reference applies with
    `load_state_dict(strict=False)` when all of hyperparams.yaml / masknet.ckpt / encoder.ckpt / decoder.ckpt exist
    (back/api.py:729-746).  Returns one flat dict with `encoder.` / `masknet.` / `decoder.` prefixes, or None when the
    base checkpoint is absent.  Files are read with weights_only=True."""
    cache_dir = cache_dir or os.environ.get("MODEL_CACHE_DIR", "models")
    base = os.path.join(cache_dir, "resepformer")
    parts = ("encoder", "masknet", "decoder")
    if not all(os.path.exists(os.path.join(base, f"{p}.ckpt")) for p in parts):
        return None
    sd: Dict[str, torch.Tensor] = {}
    for p in parts:
        for k, v in torch.load(os.path.join(base, f"{p}.ckpt"), map_location="cpu", weights_only=True).items():
            sd[f"{p}.{k}"] = v.float()
    ft = os.path.join(cache_dir, "resepformer-ft")
    if all(os.path.exists(os.path.join(ft, f)) for f in ("hyperparams.yaml", "masknet.ckpt", "encoder.ckpt", "decoder.ckpt")):
        for p in parts:
            for k, v in torch.load(os.path.join(ft, f"{p}.ckpt"), map_location="cpu", weights_only=True).items():
                if f"{p}.{k}" in sd and sd[f"{p}.{k}"].shape == v.shape:      # strict=False: unknown keys are ignored
                    sd[f"{p}.{k}"] = v.float()
    return sd


def sinc_filters(low_hz_: torch.Tensor, band_hz_: torch.Tensor, sample_rate: int = 16000, min_low_hz: float = 50.0,
                 min_band_hz: float = 50.0, kernel: int = 251) -> torch.Tensor:
    """Expand the learnable (low_hz_, band_hz_) of asteroid's ParamSincFB into its 40 cos + 40 sin
    band-pass filters [80, kernel] (Hamming-windowed sinc differences); done once at load time."""
    low_hz_ = low_hz_.double().view(-1, 1)
    band_hz_ = band_hz_.double().view(-1, 1)
    half = kernel // 2
    n_lin = torch.linspace(0, kernel / 2 - 1, steps=half, dtype=torch.float64)
    window = 0.54 - 0.46 * torch.cos(2 * math.pi * n_lin / kernel)
    n_ = 2 * math.pi * torch.arange(-half, 0, dtype=torch.float64).view(1, -1) / sample_rate
    low = min_low_hz + low_hz_.abs()
    high = torch.clamp(low + min_band_hz + band_hz_.abs(), min_low_hz, sample_rate / 2)
    band = (high - low)[:, 0]
    ft_low, ft_high = low @ n_, high @ n_
    cos_l = ((torch.sin(ft_high) - torch.sin(ft_low)) / (n_ / 2)) * window
    sin_l = ((torch.cos(ft_low) - torch.cos(ft_high)) / (n_ / 2)) * window
    cos_f = torch.cat([cos_l, 2 * band.view(-1, 1), torch.flip(cos_l, dims=[1])], dim=1)
    sin_f = torch.cat([sin_l, torch.zeros_like(band.view(-1, 1)), -torch.flip(sin_l, dims=[1])], dim=1)
    return (torch.cat([cos_f, sin_f], dim=0) / (2 * band.repeat(2)[:, None])).float()


def _synthetic_sincnet(sd: Dict[str, torch.Tensor], g: torch.Generator, prefix: str = "sincnet."):
    import numpy as np
    to_mel = lambda hz: 2595 * np.log10(1 + hz / 700)
    to_hz = lambda mel: 700 * (10 ** (mel / 2595) - 1)
    hz = to_hz(np.linspace(to_mel(30.0), to_mel(16000 / 2 - 100.0), 41))
    sd[prefix + "conv1d.0.filterbank.low_hz_"] = torch.tensor(hz[:-1], dtype=torch.float32).view(-1, 1)
    sd[prefix + "conv1d.0.filterbank.band_hz_"] = torch.tensor(np.diff(hz), dtype=torch.float32).view(-1, 1)
    sd[prefix + "wav_norm1d.weight"] = torch.tensor([1.1])
    sd[prefix + "wav_norm1d.bias"] = torch.tensor([0.05])
    sd[prefix + "conv1d.1.weight"] = torch.randn(60, 80, 5, generator=g) / math.sqrt(400)
    sd[prefix + "conv1d.1.bias"] = 0.1 * torch.randn(60, generator=g)
    sd[prefix + "conv1d.2.weight"] = torch.randn(60, 60, 5, generator=g) / math.sqrt(300)
    sd[prefix + "conv1d.2.bias"] = 0.1 * torch.randn(60, generator=g)
    for i, c in enumerate((80, 60, 60)):
        sd[prefix + f"norm1d.{i}.weight"] = 1 + 0.1 * torch.randn(c, generator=g)
        sd[prefix + f"norm1d.{i}.bias"] = 0.1 * torch.randn(c, generator=g)


def synthetic_xvector_state_dict(seed: int = 0) -> Dict[str, torch.Tensor]:
    """`pyannote/embedding` (XVectorSincNet) key layout with seeded weights."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    _synthetic_sincnet(sd, g)
    cin = 60
    for i, (cout, k) in enumerate(zip((512, 512, 512, 512, 1500), (5, 3, 3, 1, 1))):
        sd[f"tdnns.{i}.0.weight"] = torch.randn(cout, cin, k, generator=g) / math.sqrt(cin * k)
        sd[f"tdnns.{i}.0.bias"] = 0.1 * torch.randn(cout, generator=g)
        sd[f"tdnns.{i}.2.weight"] = 1 + 0.1 * torch.randn(cout, generator=g)
        sd[f"tdnns.{i}.2.bias"] = 0.1 * torch.randn(cout, generator=g)
        sd[f"tdnns.{i}.2.running_mean"] = 0.1 * torch.randn(cout, generator=g)
        sd[f"tdnns.{i}.2.running_var"] = 0.5 + torch.rand(cout, generator=g)
        cin = cout
    sd["embedding.weight"] = torch.randn(512, 3000, generator=g) / math.sqrt(3000)
    sd["embedding.bias"] = 0.1 * torch.randn(512, generator=g)
    return sd


RESNET34_LAYERS = (3, 4, 6, 3)


def synthetic_resnet34_state_dict(seed: int = 0, m_channels: int = 32, feat_dim: int = 80, embed_dim: int = 256) -> Dict[str, torch.Tensor]:
    """`pyannote/wespeaker-voxceleb-resnet34-LM` key layout (resnet.conv1 / bn1 / layer{1..4}.{i}.conv{1,2} / bn{1,2} /
    shortcut.{0,1} / seg_1) with seeded weights (He-scaled convolutions, mild BatchNorm statistics)."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def bn(name, c):
        sd[name + ".weight"] = 1 + 0.1 * torch.randn(c, generator=g)
        sd[name + ".bias"] = 0.1 * torch.randn(c, generator=g)
        sd[name + ".running_mean"] = 0.1 * torch.randn(c, generator=g)
        sd[name + ".running_var"] = 0.5 + torch.rand(c, generator=g)

    def conv(name, cout, cin, k):
        sd[name] = torch.randn(cout, cin, k, k, generator=g) * math.sqrt(2.0 / (cin * k * k))

    conv("resnet.conv1.weight", m_channels, 1, 3)
    bn("resnet.bn1", m_channels)
    cin = m_channels
    for li, nblocks in enumerate(RESNET34_LAYERS, start=1):
        cout = m_channels * (1 << (li - 1))
        for bi in range(nblocks):
            p = f"resnet.layer{li}.{bi}."
            stride = 2 if (li > 1 and bi == 0) else 1
            conv(p + "conv1.weight", cout, cin, 3)
            bn(p + "bn1", cout)
            conv(p + "conv2.weight", cout, cout, 3)
            sd[p + "conv2.weight"] *= 0.5      # keeps the residual stream O(1..10) through 16 blocks
            bn(p + "bn2", cout)
            if stride != 1 or cin != cout:
                conv(p + "shortcut.0.weight", cout, cin, 1)
                bn(p + "shortcut.1", cout)
            cin = cout
    stats = cin * (feat_dim // 8) * 2
    sd["resnet.seg_1.weight"] = torch.randn(embed_dim, stats, generator=g) / math.sqrt(stats)
    sd["resnet.seg_1.bias"] = 0.1 * torch.randn(embed_dim, generator=g)
    return sd


def synthetic_pyannet_state_dict(n_classes: int = 7, seed: int = 0) -> Dict[str, torch.Tensor]:
    """PyanNet (pyannote/segmentation*) key layout with seeded weights: SincNet, 4 BiLSTM(128), 2 linear, classifier."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    _synthetic_sincnet(sd, g)
    for l in range(4):
        cin = 60 if l == 0 else 256
        for sfx in ("", "_reverse"):
            sd[f"lstm.weight_ih_l{l}{sfx}"] = torch.randn(512, cin, generator=g) / math.sqrt(cin)
            sd[f"lstm.weight_hh_l{l}{sfx}"] = torch.randn(512, 128, generator=g) / math.sqrt(128)
            sd[f"lstm.bias_ih_l{l}{sfx}"] = 0.1 * torch.randn(512, generator=g)
            sd[f"lstm.bias_hh_l{l}{sfx}"] = 0.1 * torch.randn(512, generator=g)
    sd["linear.0.weight"] = torch.randn(128, 256, generator=g) / math.sqrt(256)
    sd["linear.0.bias"] = 0.1 * torch.randn(128, generator=g)
    sd["linear.1.weight"] = torch.randn(128, 128, generator=g) / math.sqrt(128)
    sd["linear.1.bias"] = 0.1 * torch.randn(128, generator=g)
    sd["classifier.weight"] = torch.randn(n_classes, 128, generator=g) / math.sqrt(128)
    sd["classifier.bias"] = 0.1 * torch.randn(n_classes, generator=g)
    return sd
