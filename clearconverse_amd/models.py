"""Model set of the hot path: builds the five libccx-backed objects the processor drives
(reference: `_load_resepformer_model`, `_load_whisper_model`, `_initialize_pyannote_models`,
/root/reference/back/api.py:657-797).  Real checkpoints are used when present under MODEL_CACHE_DIR;
there is no network in the build/bench environment, so otherwise seeded synthetic weights of the same
architectures are generated.  No CPU fallback: every object needs the HIP
library and a GPU."""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch

from . import _lib
from .denoise import SpectralGate
from .pipelines import SpeakerDiarization, VoiceActivityDetection
from .separator import SepformerSeparator
from .speaker import ResNetEmbedder, SegmentationNet, XVectorEmbedder
from .weights import (SepDims, WhisperDims, find_pipeline_config, find_pyannote_checkpoint, find_sepformer_checkpoint,
                      find_whisper_checkpoint, synthetic_pyannet_state_dict, synthetic_resnet34_state_dict,
                      synthetic_sepformer_state_dict, synthetic_whisper_state_dict, synthetic_xvector_state_dict)
from .whisper import WhisperModel


def build_state_dicts(config=None, whisper_dims: Optional[WhisperDims] = None, sep_dims: Optional[SepDims] = None,
                      seed: int = 0) -> Dict[str, object]:
    """Every weight of the path as host tensors: the real Whisper checkpoint when MODEL_CACHE_DIR holds one, seeded
    synthetic weights of the same architectures otherwise.  In a multi-GPU job rank 0 calls this and the result
    travels by ONE broadcast (batch.broadcast_weights)."""
    size = getattr(config, "whisper_model_size", "small.en") if config is not None else "small.en"
    ck = find_whisper_checkpoint(size) if whisper_dims is None else None
    if ck is not None:
        wd, wsd = ck
    else:
        wd = whisper_dims or WhisperDims.small_en()
        wsd = synthetic_whisper_state_dict(wd, seed=seed)
    sd_ = sep_dims or SepDims()
    sep_ck = find_sepformer_checkpoint() if sep_dims is None else None
    # pyannote-side networks (reference back/api.py:776-792): real checkpoints when the hub cache holds them, seeded otherwise
    synth = {"xvector": lambda: synthetic_xvector_state_dict(seed=seed + 2), "pyannet_diar": lambda: synthetic_pyannet_state_dict(7, seed=seed + 3),
             "pyannet_vad": lambda: synthetic_pyannet_state_dict(3, seed=seed + 4), "resnet34": lambda: synthetic_resnet34_state_dict(seed=seed + 5)}
    nets, sources = {}, {"whisper": "checkpoint" if ck is not None else f"synthetic-seed{seed}",
                         "sepformer": "checkpoint" if sep_ck is not None else f"synthetic-seed{seed + 1}"}
    for kind, make in synth.items():
        real = find_pyannote_checkpoint(kind)
        nets[kind] = real if real is not None else make()
        sources[kind] = "checkpoint" if real is not None else "synthetic"
    return {
        "whisper_dims": dict(wd.__dict__), "whisper": wsd, "whisper_source": sources["whisper"],
        "sep_dims": dict(sd_.__dict__),
        "sepformer": sep_ck or synthetic_sepformer_state_dict(sd_, seed=seed + 1),
        **nets,
        "weights_sources": sources,
        # pipeline hyper-parameters: the pipelines' own config.yaml when on disk, the published values otherwise
        "vad_params": find_pipeline_config("vad"), "diarization_params": find_pipeline_config("diarization"),
    }


def load_models(config=None, device=None, whisper_batch: int = 8, ctx: Optional[_lib.Context] = None, max_audio_seconds: float = 600.0,
                whisper_dims: Optional[WhisperDims] = None, sep_dims: Optional[SepDims] = None, seed: int = 0,
                sep_tokens: int = 160_000, max_crops: int = 256, state_dicts: Optional[Dict[str, object]] = None,
                seg_max_crops: Optional[int] = None, seg_max_seconds: float = 1200.0, emb_max_crops: Optional[int] = None,
                resnet_max_chunks: int = 96, whisper_instances: int = 1, share_encoder_scratch: bool = True,
                gate_max_clips: int = 32, gate_max_seconds: float = 30.0) -> Dict[str, object]:
    if not torch.cuda.is_available():
        raise _lib.CcxError("load_models needs a ROCm GPU: the HIP path has no CPU fallback")
    dev_index = device.index if isinstance(device, torch.device) and device.index is not None else (device if isinstance(device, int) else 0)
    ctx = ctx or _lib.Context(dev_index)
    W = state_dicts if state_dicts is not None else build_state_dicts(config, whisper_dims, sep_dims, seed)
    wd, wsd = WhisperDims(**W["whisper_dims"]), W["whisper"]
    if state_dicts is not None:         # the geometry travels WITH the weights: a contradicting argument is an error, not ignored
        if whisper_dims is not None and whisper_dims != wd:
            raise ValueError(f"load_models: whisper_dims {whisper_dims} contradicts state_dicts['whisper_dims'] {wd}")
        if sep_dims is not None and sep_dims != SepDims(**W["sep_dims"]):
            raise ValueError(f"load_models: sep_dims {sep_dims} contradicts state_dicts['sep_dims'] {W['sep_dims']}")
    vp, dp = W.get("vad_params") or {}, W.get("diarization_params") or {}
    sd_ = SepDims(**W["sep_dims"])      # the geometry the separator weights were built with (broadcast manifest included)
    # whisper_instances = 2: the software-pipelined batch driver (batch.py) encodes batch i + 1 into one instance while batch i
    # is still decoding out of the other (each holds its own cross-KV, workspaces and step graphs; the weights are 0.5 GB)
    # The further instances take the first one's log-mel / encoder workspaces (share_encoder_scratch; 32 GB at 768 windows): those
    # are only live inside log_mel / encode, and the pipelined driver orders every instance's log_mel / encode on one stream.
    whispers = [WhisperModel(wd, wsd, max_batch=whisper_batch, device=dev_index, ctx=ctx, max_audio_seconds=max_audio_seconds)]
    for _ in range(1, max(1, int(whisper_instances))):
        whispers.append(WhisperModel(wd, wsd, max_batch=whisper_batch, device=dev_index, ctx=ctx, max_audio_seconds=max_audio_seconds,
                                     share_encoder_scratch_with=whispers[0] if share_encoder_scratch else None))
    whisper = whispers[0]
    separator = SepformerSeparator(sd_, W["sepformer"], max_tokens=sep_tokens, max_utts=64,
                                   device=dev_index, ctx=ctx)
    embedder = XVectorEmbedder(W["xvector"], max_crops=int(emb_max_crops or max_crops), max_samples=16000 * 1200,
                               device=dev_index, ctx=ctx)
    # embedding model of the diarization pipeline (speaker-diarization-3.1 uses WeSpeaker ResNet-34, not pyannote/embedding)
    diar_embedder = ResNetEmbedder(W["resnet34"], max_chunks=int(resnet_max_chunks), max_samples=160000, max_masks=max(512, 3 * int(resnet_max_chunks)),
                                   device=dev_index, ctx=ctx)
    # one launch group of the segmentation nets holds seg_max_crops windows / seg_max_seconds of audio (≈60 B of device
    # buffers per sample): the batched pipeline raises both so that every window of a step shares one LSTM launch
    seg_crops = int(seg_max_crops or max_crops)
    seg_diar = SegmentationNet(W["pyannet_diar"], n_classes=7, powerset=True, max_crops=seg_crops,
                               max_samples=int(16000 * seg_max_seconds), device=dev_index, ctx=ctx)
    seg_vad = SegmentationNet(W["pyannet_vad"], n_classes=3, powerset=False, max_crops=seg_crops,
                              max_samples=int(16000 * seg_max_seconds), device=dev_index, ctx=ctx)
    # one launch of the spectral gate holds gate_max_clips rows of gate_max_seconds (longer inputs take the chunked path)
    gate = SpectralGate(max_samples=int(16000 * gate_max_seconds), max_clips=int(gate_max_clips), device=dev_index, ctx=ctx)
    return {
        "ctx": ctx,
        "weights_source": W.get("whisper_source", "given"),
        "whisper_model": whisper,
        "whisper_models": whispers,
        "separator": separator,
        "embedding_model": embedder,
        "vad_pipeline": VoiceActivityDetection(seg_vad, batch=seg_crops, **{k: vp[k] for k in ("onset", "offset", "min_duration_on", "min_duration_off") if k in vp}),
        "diarization": SpeakerDiarization(seg_diar, diar_embedder, batch=seg_crops,
                                          **{k: dp[k] for k in ("threshold", "min_cluster_size", "min_duration_off") if k in dp}),
        "weights_sources": W.get("weights_sources", {}),
        "diarization_embedder": diar_embedder,
        "denoiser": gate,
        "segmentation_vad": seg_vad,
        "segmentation_diar": seg_diar,
    }
