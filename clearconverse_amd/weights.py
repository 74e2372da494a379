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
    reference applies, back/api.py:671-692) or None when no checkpoint is present.  Overlay semantics are those of
    `load_state_dict(strict=False)` as the reference runs it: unknown keys are ignored; a key whose shape does not match is not
    copied (torch raises afterwards, the reference logs the error and keeps the model -- with every matching key already applied)."""
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
        sd.update({k: v.float() for k, v in over.items() if k in sd and sd[k].shape == v.shape})  # strict=False; a size mismatch is skipped
    elif os.path.exists(pt_path):
        over = torch.load(pt_path, map_location="cpu", weights_only=True)
        sd.update({k: v.float() for k, v in over.items() if torch.is_tensor(v) and k in sd and sd[k].shape == v.shape})
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


def find_sepformer_checkpoint(cache_dir: Optional[str] = None, apply_ft_overlay: Optional[bool] = None) -> Optional[Dict[str, torch.Tensor]]:
    """SpeechBrain savedir layout the reference uses (back/api.py:713-717): `<cache>/resepformer/{encoder,masknet,decoder}.ckpt`,
    each the state_dict of that module.  Returns one flat dict with `encoder.` / `masknet.` / `decoder.` prefixes, or None when
    the base checkpoint is absent.  Files are read with weights_only=True.

    The fine-tune overlay `<cache>/resepformer-ft/`: the reference calls
    `self.separator.load_state_dict({'masknet': {...}, 'encoder': {...}, 'decoder': {...}}, strict=False)` (back/api.py:739-746).
    Those three top-level keys name no parameter of the SpeechBrain module (its keys are `mods.<part>.<name>`), and strict=False
    drops unexpected keys silently, so the reference KEEPS THE BASE WEIGHTS.  Default here = that effective behaviour (overlay
    ignored).  `apply_ft_overlay=True` (or CCX_APPLY_RESEPFORMER_FT=1) applies the overlay tensor by tensor -- what the reference's
    author evidently intended, and a deliberate deviation from what the reference does."""
    cache_dir = cache_dir or os.environ.get("MODEL_CACHE_DIR", "models")
    if apply_ft_overlay is None:
        apply_ft_overlay = os.environ.get("CCX_APPLY_RESEPFORMER_FT", "0") not in ("", "0")
    base = os.path.join(cache_dir, "resepformer")
    parts = ("encoder", "masknet", "decoder")
    if not all(os.path.exists(os.path.join(base, f"{p}.ckpt")) for p in parts):
        return None
    sd: Dict[str, torch.Tensor] = {}
    for p in parts:
        for k, v in torch.load(os.path.join(base, f"{p}.ckpt"), map_location="cpu", weights_only=True).items():
            sd[f"{p}.{k}"] = v.float()
    ft = os.path.join(cache_dir, "resepformer-ft")
    if apply_ft_overlay and all(os.path.exists(os.path.join(ft, f)) for f in ("hyperparams.yaml", "masknet.ckpt", "encoder.ckpt", "decoder.ckpt")):
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


# ----------------------------------------------------------------------------------------------
# pyannote-side checkpoints and pipeline hyper-parameters (reference back/api.py:776-792)
# ----------------------------------------------------------------------------------------------
# What the reference's three constructors pull from the Hugging Face hub [UPSTREAM-RECALL for the repository contents]:
#   Inference("pyannote/embedding")                            -> pyannote/embedding/pytorch_model.bin            (XVectorSincNet)
#   Pipeline.from_pretrained("pyannote/voice-activity-detection", cache_dir=<cache>/vad)
#        -> config.yaml {pipeline.params.segmentation: pyannote/segmentation, params: onset/offset/min_duration_on/off}
#        -> pyannote/segmentation/pytorch_model.bin                                                               (PyanNet, 3 classes)
#   Pipeline.from_pretrained("pyannote/speaker-diarization-3.1", cache_dir=<cache>/speaker-diarization)
#        -> config.yaml {params.clustering: method/min_cluster_size/threshold, params.segmentation.min_duration_off}
#        -> pyannote/segmentation-3.0/pytorch_model.bin (PyanNet, 7 powerset classes), pyannote/wespeaker-voxceleb-resnet34-LM
# A pytorch_model.bin is a Lightning checkpoint: {"state_dict": {...}, "hyper_parameters": ..., "pyannote.audio": ...}.
PYANNOTE_MODELS = {
    "xvector": ("pyannote/embedding", ("embedding",)),
    "pyannet_vad": ("pyannote/segmentation", ("vad",)),
    "pyannet_diar": ("pyannote/segmentation-3.0", ("speaker-diarization",)),
    "resnet34": ("pyannote/wespeaker-voxceleb-resnet34-LM", ("speaker-diarization",)),
}
PYANNOTE_PIPELINES = {
    "vad": ("pyannote/voice-activity-detection", "vad"),
    "diarization": ("pyannote/speaker-diarization-3.1", "speaker-diarization"),
}
# defaults = published values as recalled [UPSTREAM-RECALL]; a config.yaml found on disk overrides them.  The VAD numbers are the
# ones the `pyannote/segmentation` model card gives for voice activity detection on AMI (onset 0.767 / offset 0.377 / min_duration_on
# 0.136 / min_duration_off 0.067); whether `pyannote/voice-activity-detection`'s own config.yaml carries exactly these is NOT
# verifiable offline (parity unpinned) -- they only apply when that file is absent from MODEL_CACHE_DIR.
VAD_DEFAULTS = dict(onset=0.767, offset=0.377, min_duration_on=0.136, min_duration_off=0.067)
DIAR_DEFAULTS = dict(threshold=0.7045654963945799, min_cluster_size=12, min_duration_off=0.0, method="centroid")


def _hub_roots(cache_dir: str, subdirs) -> list:
    roots = [os.path.join(cache_dir, s) for s in subdirs] + [cache_dir]
    for env, tail in (("PYANNOTE_CACHE", ""), ("HF_HOME", "hub"), ("HUGGINGFACE_HUB_CACHE", ""), ("HF_HUB_CACHE", "")):
        v = os.environ.get(env)
        if v:
            roots.append(os.path.join(v, tail) if tail else v)
    home = os.path.expanduser("~")
    roots += [os.path.join(home, ".cache", "torch", "pyannote"), os.path.join(home, ".cache", "huggingface", "hub")]
    return roots


def _hub_file(roots, repo: str, filenames) -> Optional[str]:
    """First existing file among: the hub cache layout `<root>/models--<org>--<name>/snapshots/<rev>/<file>`, a plain checkout
    `<root>/<org>/<name>/<file>` or `<root>/<name>/<file>`."""
    import glob
    org, name = repo.split("/")
    for root in roots:
        for fn in filenames:
            cands = sorted(glob.glob(os.path.join(root, f"models--{org}--{name}", "snapshots", "*", fn)))
            cands += [os.path.join(root, org, name, fn), os.path.join(root, name, fn)]
            for c in cands:
                if os.path.isfile(c):
                    return c
    return None


class CheckpointRefused(RuntimeError):
    """The weights-only loader would not read a file (it holds pickled objects beyond tensors and plain containers)."""


def load_state_dict_file(path: str) -> Dict[str, torch.Tensor]:
    """Tensors of a checkpoint file WITHOUT executing anything from it: safetensors, or torch.load(weights_only=True) of either
    a plain state_dict or a Lightning checkpoint ({"state_dict": ...}).  Raises CheckpointRefused when the safe loader refuses
    the file (pyannote's published pytorch_model.bin files carry pickled task specifications; convert them once with
    `torch.save(ckpt["state_dict"], path)` or to safetensors in an environment that has pyannote.audio)."""
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return {k: v.float() for k, v in load_file(path, device="cpu").items()}
    try:
        ck = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:           # pickle.UnpicklingError: unsupported global ...
        raise CheckpointRefused(f"{path}: the weights-only loader refused this file ({type(e).__name__}: {str(e)[:200]})") from e
    if isinstance(ck, dict) and isinstance(ck.get("state_dict"), dict):
        ck = ck["state_dict"]
    if not isinstance(ck, dict):
        raise CheckpointRefused(f"{path}: not a state_dict")
    return {k: v.float() if torch.is_floating_point(v) else v for k, v in ck.items() if torch.is_tensor(v)}


def _conform(kind: str, sd: Dict[str, torch.Tensor], schema: Dict[str, torch.Tensor], path: str) -> Dict[str, torch.Tensor]:
    """Keep exactly the tensors the kernels take (names + shapes of the architecture built here); fail loudly on a missing key or
    a wrong shape -- a silently half-loaded network would run on noise."""
    out = {}
    for k, ref in schema.items():
        if k not in sd:
            raise ValueError(f"{path}: {kind} checkpoint lacks '{k}' (not the architecture built here)")
        if tuple(sd[k].shape) != tuple(ref.shape):
            # only size-1 dimensions may differ (a Conv1d weight saved as [out, in, 1] for a Linear, a squeezed scalar ...): the
            # element ORDER is then unchanged.  Any other same-numel mismatch (a transposed or permuted weight) is an error
            squeeze = lambda shp: tuple(int(d) for d in shp if int(d) != 1)
            if squeeze(sd[k].shape) == squeeze(ref.shape):
                out[k] = sd[k].reshape(ref.shape)
                continue
            raise ValueError(f"{path}: '{k}' has shape {tuple(sd[k].shape)}, expected {tuple(ref.shape)}")
        out[k] = sd[k]
    return out


def find_pyannote_checkpoint(kind: str, cache_dir: Optional[str] = None, log=None) -> Optional[Dict[str, torch.Tensor]]:
    """State dict of one of the pyannote-side networks (`kind` in PYANNOTE_MODELS) from the places the reference's loaders
    leave them (hub cache under MODEL_CACHE_DIR/<embedding|vad|speaker-diarization>, PYANNOTE_CACHE, HF_HOME) or None when no
    usable file exists.  A file the weights-only loader refuses is reported through `log` and skipped."""
    cache_dir = cache_dir or os.environ.get("MODEL_CACHE_DIR", "models")
    repo, subdirs = PYANNOTE_MODELS[kind]
    path = _hub_file(_hub_roots(cache_dir, subdirs), repo, ("model.safetensors", "pytorch_model.bin", "pytorch_model.pt"))
    if path is None:
        return None
    try:
        sd = load_state_dict_file(path)
    except CheckpointRefused as e:
        (log or print)(f"[ccx weights] {e}; using synthetic weights for {kind}")
        return None
    schema = {"xvector": synthetic_xvector_state_dict, "resnet34": synthetic_resnet34_state_dict,
              "pyannet_vad": lambda: synthetic_pyannet_state_dict(3), "pyannet_diar": lambda: synthetic_pyannet_state_dict(7)}[kind]()
    return _conform(kind, sd, schema, path)


def find_pipeline_config(which: str, cache_dir: Optional[str] = None) -> Dict[str, object]:
    """Hyper-parameters of `pyannote/voice-activity-detection` (which="vad") or `pyannote/speaker-diarization-3.1`
    (which="diarization") from the pipeline's config.yaml when it is on disk (yaml.safe_load), else the recalled defaults.
    The returned dict says where the values came from (`source`)."""
    import yaml
    cache_dir = cache_dir or os.environ.get("MODEL_CACHE_DIR", "models")
    repo, sub = PYANNOTE_PIPELINES[which]
    out: Dict[str, object] = dict(VAD_DEFAULTS if which == "vad" else DIAR_DEFAULTS)
    out["source"] = "defaults"
    path = _hub_file(_hub_roots(cache_dir, (sub,)), repo, ("config.yaml",))
    if path is None:
        return out
    with open(path, "r", encoding="utf-8") as f:
        cfg = yaml.safe_load(f) or {}
    params = cfg.get("params") or {}
    if which == "vad":
        for k in ("onset", "offset", "min_duration_on", "min_duration_off"):
            if k in params:
                out[k] = float(params[k])
    else:
        cl = params.get("clustering") or {}
        if "threshold" in cl:
            out["threshold"] = float(cl["threshold"])
        if "min_cluster_size" in cl:
            out["min_cluster_size"] = int(cl["min_cluster_size"])
        if "method" in cl:
            out["method"] = str(cl["method"])
        seg = params.get("segmentation") or {}
        if "min_duration_off" in seg:
            out["min_duration_off"] = float(seg["min_duration_off"])
        if out["method"] != "centroid":
            raise ValueError(f"{path}: clustering method '{out['method']}' is not implemented (only centroid linkage, the 3.1 default)")
    pp = (cfg.get("pipeline") or {}).get("params") or {}
    out["models"] = {k: v for k, v in pp.items() if isinstance(v, str)}
    out["source"] = path
    return out
