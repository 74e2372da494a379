"""ctypes binding of libccx.so (the C ABI declared in include/ccx.h).

The product path has NO CPU fallback: if the HIP library is missing or a call fails, a
`CcxError` is raised.  Tensors are torch ROCm tensors; only `data_ptr()` and the current HIP
stream cross the boundary.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "libccx.so"


class CcxError(RuntimeError):
    pass


class WhisperDims(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer",
        "n_vocab", "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer")]


class SepformerDims(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "n_filters", "kernel", "stride", "d_model", "n_head", "d_ffn", "n_layers", "n_blocks", "segment", "n_spk")]


class DecodeRules(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "eot", "sot", "sot_prev", "no_speech", "no_timestamps", "timestamp_begin", "blank",
        "max_initial_timestamp_index", "n_suppress")] + [("suppress", C.POINTER(C.c_int))]


_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float
_ip = C.POINTER(C.c_int)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_fp = C.POINTER(C.c_float)

# name -> (restype, argtypes); every symbol declared in include/ccx.h
PROTOTYPES = {
    "ccx_version": (C.c_char_p, []),
    "ccx_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "ccx_ctx_destroy": (None, [_vp]),
    "ccx_last_error": (C.c_char_p, [_vp]),
    "ccx_prof_enable": (_i, [_vp, _i]),
    "ccx_prof_count": (_i, [_vp]),
    "ccx_prof_get": (_i, [_vp, _i, C.c_char_p, _i, C.POINTER(C.c_double), C.POINTER(C.c_double), _fp]),
    "ccx_gemm_bf16": (_i, [_vp, _i, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _i, _i, _i, _vp]),
    "ccx_layernorm": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "ccx_gather_rows": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i64, _vp]),
    "ccx_peak_normalize": (_i, [_vp, _vp, _vp, _i64, _vp, _i, _f, _vp]),
    "ccx_row_variance": (_i, [_vp, _vp, _i64, _vp, _i, _vp, _vp]),
    "ccx_cross_attention_xa": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "ccx_cosine_rows": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "ccx_speaker_profiles": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "ccx_resample_sinc": (_i, [_vp, _vp, _i64, _vp, _i, _i, _i, _i, _vp, _vp, _i64, _vp, _i, _vp]),
    "ccx_enc_attention": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ccx_whisper_create": (_i, [_vp, C.POINTER(WhisperDims), _i, C.POINTER(_vp)]),
    "ccx_whisper_destroy": (None, [_vp]),
    "ccx_whisper_set_tensor": (_i, [_vp, C.c_char_p, _vp, _i, _i, _i64p]),
    "ccx_whisper_set_max_audio": (_i, [_vp, C.c_double]),
    "ccx_whisper_share_encoder_scratch": (_i, [_vp, _vp]),
    "ccx_whisper_finalize": (_i, [_vp]),
    "ccx_whisper_set_rules": (_i, [_vp, C.POINTER(DecodeRules)]),
    "ccx_whisper_logmel": (_i, [_vp, _vp, _i64, _ip, _ip, _i, _vp, _vp]),
    "ccx_whisper_set_mel": (_i, [_vp, _vp, _i, _vp]),
    "ccx_whisper_encode": (_i, [_vp, _i, _vp, _vp]),
    "ccx_whisper_decoder_logits": (_i, [_vp, _i32p, _i, _i, _vp, _vp]),
    "ccx_whisper_decode_greedy": (_i, [_vp, _i32p, _i32p, _i, _i, _i, _i32p, _i32p, _fp, _fp, _vp]),
    "ccx_whisper_last_cross_path": (_i, [_vp]),
    "ccx_whisper_prepare_lanes": (_i, [_vp, _vp]),
    "ccx_whisper_trace_lanes": (_i, [_vp, C.c_char_p, _i]),
    "ccx_whisper_decode": (_i, [_vp, _i32p, _i32p, _i, _i, _i, _f, C.c_uint64, _i32p, _i32p, _fp, _fp, _vp]),
    "ccx_sepformer_create": (_i, [_vp, C.POINTER(SepformerDims), _i, _i, C.POINTER(_vp)]),
    "ccx_sepformer_destroy": (None, [_vp]),
    "ccx_sepformer_set_tensor": (_i, [_vp, C.c_char_p, _vp, _i64]),
    "ccx_sepformer_finalize": (_i, [_vp]),
    "ccx_sepformer_separate": (_i, [_vp, _vp, _i64, _ip, _i, _vp, _vp]),
    "ccx_speaker_create": (_i, [_vp, _i, _i, _i, _i, _i64, C.POINTER(_vp)]),
    "ccx_speaker_destroy": (None, [_vp]),
    "ccx_speaker_set_tensor": (_i, [_vp, C.c_char_p, _vp, _i64]),
    "ccx_speaker_finalize": (_i, [_vp]),
    "ccx_speaker_embed": (_i, [_vp, _vp, _i64p, _ip, _i, _vp, _i64p, _ip, _vp, _vp]),
    "ccx_speaker_segment": (_i, [_vp, _vp, _i64p, _ip, _i, _vp, _i64, _ip, _vp]),
    "ccx_resnet_create": (_i, [_vp, _i, _i64, _i, C.POINTER(_vp)]),
    "ccx_resnet_destroy": (None, [_vp]),
    "ccx_resnet_set_tensor": (_i, [_vp, C.c_char_p, _vp, _i64]),
    "ccx_resnet_finalize": (_i, [_vp]),
    "ccx_resnet_embed": (_i, [_vp, _vp, _i64, _i, _i, _vp, _i, _ip, _i, _vp, _vp]),
    "ccx_specgate_create": (_i, [_vp, _i64, _i, _i, C.POINTER(_vp)]),
    "ccx_specgate_destroy": (None, [_vp]),
    "ccx_specgate_reduce": (_i, [_vp, _vp, _i64, _ip, _i, _f, _vp, _vp]),
    "ccx_specgate_reduce_long": (_i, [_vp, _vp, _i64, _f, _vp, _vp]),
    "ccx_specgate_set_clip_noise": (_i, [_vp, _i]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """dlopen libccx.so and bind every prototype.  Raises CcxError when the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise CcxError(f"{LIB_PATH} is missing: run `python -m clearconverse_amd.build` "
                       "(there is no CPU fallback for the HIP path)")
    # torch first: libccx.so needs libamdhip64, and the process must end up with ONE HIP runtime.  PyTorch's ROCm wheels carry
    # their own copy; if libccx were loaded before torch, it would bind /opt/rocm's copy and torch would later load a second
    # runtime -- ccx_ctx_create then reports "no ROCm-capable device" (seen with build() and smoke() in one process).
    import torch  # noqa: F401
    try:
        lib = C.CDLL(str(LIB_PATH))
    except OSError as e:
        raise CcxError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise CcxError(f"libccx.so does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class Context:
    """One ccx_ctx per process/device (created after fork, reference back/api.py:2045-2049)."""

    def __init__(self, device: int = 0):
        self.lib = load()
        h = _vp()
        rc = self.lib.ccx_ctx_create(device, C.byref(h))
        if rc != 0:
            raise CcxError(f"ccx_ctx_create({device}) failed: {self.lib.ccx_last_error(None).decode()}")
        self.handle = h
        self.device = device

    def check(self, rc: int, what: str = ""):
        if rc != 0:
            raise CcxError(f"{what or 'ccx call'} failed ({rc}): {self.lib.ccx_last_error(self.handle).decode()}")

    prof_on = False

    def prof_enable(self, on: bool = True):
        self.check(self.lib.ccx_prof_enable(self.handle, 1 if on else 0), "ccx_prof_enable")
        self.prof_on = bool(on)

    def prof_records(self):
        """[(kernel name, algorithmic flops, algorithmic bytes, ms)] for every recorded launch."""
        out = []
        buf = C.create_string_buffer(64)
        fl, by, ms = C.c_double(), C.c_double(), C.c_float()
        for i in range(self.lib.ccx_prof_count(self.handle)):
            self.check(self.lib.ccx_prof_get(self.handle, i, buf, 64, C.byref(fl), C.byref(by), C.byref(ms)), "ccx_prof_get")
            out.append((buf.value.decode(), fl.value, by.value, ms.value))
        return out

    def close(self):
        if getattr(self, "handle", None):
            self.lib.ccx_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def current_stream_ptr() -> int:
    import torch
    return int(torch.cuda.current_stream().cuda_stream)


def ptr(t) -> int:
    """Device (or host) address of a contiguous torch tensor; None -> NULL."""
    if t is None:
        return 0
    if not t.is_contiguous():
        raise CcxError("tensor passed to libccx must be contiguous")
    return int(t.data_ptr())
