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
from .speaker import SegmentationNet, XVectorEmbedder
from .weights import (SepDims, WhisperDims, find_whisper_checkpoint, synthetic_pyannet_state_dict,
                      synthetic_sepformer_state_dict, synthetic_whisper_state_dict, synthetic_xvector_state_dict)
from .whisper import WhisperModel


def load_models(config=None, device=None, whisper_batch: int = 8, ctx: Optional[_lib.Context] = None, max_audio_seconds: float = 30.0,
                whisper_dims: Optional[WhisperDims] = None, sep_dims: Optional[SepDims] = None, seed: int = 0,
                sep_tokens: int = 160_000, max_crops: int = 256) -> Dict[str, object]:
    if not torch.cuda.is_available():
        raise _lib.CcxError("load_models needs a ROCm GPU: the HIP path has no CPU fallback")
    dev_index = device.index if isinstance(device, torch.device) and device.index is not None else (device if isinstance(device, int) else 0)
    ctx = ctx or _lib.Context(dev_index)
    size = getattr(config, "whisper_model_size", "small.en") if config is not None else "small.en"
    ck = find_whisper_checkpoint(size) if whisper_dims is None else None
    if ck is not None:
        wd, wsd = ck
    else:
        wd = whisper_dims or WhisperDims.small_en()
        wsd = synthetic_whisper_state_dict(wd, seed=seed)
    whisper = WhisperModel(wd, wsd, max_batch=whisper_batch, device=dev_index, ctx=ctx, max_audio_seconds=max_audio_seconds)
    sd_ = sep_dims or SepDims()
    separator = SepformerSeparator(sd_, synthetic_sepformer_state_dict(sd_, seed=seed + 1), max_tokens=sep_tokens, max_utts=64,
                                   device=dev_index, ctx=ctx)
    embedder = XVectorEmbedder(synthetic_xvector_state_dict(seed=seed + 2), max_crops=max_crops, max_samples=16000 * 1200,
                               device=dev_index, ctx=ctx)
    seg_diar = SegmentationNet(synthetic_pyannet_state_dict(7, seed=seed + 3), n_classes=7, powerset=True, max_crops=max_crops,
                               max_samples=16000 * 1200, device=dev_index, ctx=ctx)
    seg_vad = SegmentationNet(synthetic_pyannet_state_dict(3, seed=seed + 4), n_classes=3, powerset=False, max_crops=max_crops,
                              max_samples=16000 * 1200, device=dev_index, ctx=ctx)
    gate = SpectralGate(max_samples=480000, max_clips=32, device=dev_index, ctx=ctx)
    return {
        "ctx": ctx,
        "whisper_model": whisper,
        "separator": separator,
        "embedding_model": embedder,
        "vad_pipeline": VoiceActivityDetection(seg_vad, batch=max_crops),
        "diarization": SpeakerDiarization(seg_diar, embedder, batch=max_crops),
        "denoiser": gate,
        "segmentation_vad": seg_vad,
        "segmentation_diar": seg_diar,
    }
