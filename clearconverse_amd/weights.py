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
