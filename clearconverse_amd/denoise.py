"""`SpectralGate`: libccx-backed replacement of `noisereduce.reduce_noise(y=, sr=, stationary=True,
prop_decrease=)` as the reference uses it (/root/reference/back/api.py:349, 832-833).  Calling the
object with (np1d, sr, prop_decrease) returns a float32 numpy array of the same length -- the
`denoiser` contract of clearconverse_amd.processor.  `reduce_batch` is the batched device entry."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib


class SpectralGate:
    def __init__(self, max_samples: int = 480000, max_clips: int = 32, sample_rate: int = 16000, device: int = 0,
                 ctx: Optional[_lib.Context] = None, clip_noise_stationary: bool = True):
        if not torch.cuda.is_available():
            raise _lib.CcxError("SpectralGate needs a ROCm GPU: the HIP path has no CPU fallback")
        self.device = torch.device("cuda", device)
        self.ctx = ctx or _lib.Context(device)
        self.lib = self.ctx.lib
        self.max_samples, self.max_clips, self.sr = int(max_samples), int(max_clips), int(sample_rate)
        h = C.c_void_p()
        self.ctx.check(self.lib.ccx_specgate_create(self.ctx.handle, self.max_samples, self.max_clips, self.sr, C.byref(h)),
                       "ccx_specgate_create")
        self.handle = h
        # signals beyond one 600000-sample chunk: noise statistics from the first chunk only (noisereduce's clip_noise_stationary=True,
        # its default) or from the whole signal (False).  One switch, parity unpinned -- see ccx.h
        self.clip_noise_stationary = bool(clip_noise_stationary)
        self.ctx.check(self.lib.ccx_specgate_set_clip_noise(self.handle, 1 if clip_noise_stationary else 0), "ccx_specgate_set_clip_noise")

    def close(self):
        if getattr(self, "handle", None):
            self.lib.ccx_specgate_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reduce_batch(self, y: torch.Tensor, n_samples: Sequence[int], prop_decrease: float) -> torch.Tensor:
        """y [B, stride] f32 on the GPU -> denoised [B, stride]."""
        assert y.is_cuda and y.dtype == torch.float32 and y.is_contiguous()
        B, stride = y.shape
        out = torch.empty_like(y)
        for b0 in range(0, B, self.max_clips):
            nb = min(self.max_clips, B - b0)
            ns = (C.c_int * nb)(*[int(v) for v in n_samples[b0:b0 + nb]])
            self.ctx.check(self.lib.ccx_specgate_reduce(self.handle, y[b0:b0 + nb].data_ptr(), stride, ns, nb, float(prop_decrease),
                                                        out[b0:b0 + nb].data_ptr(), _lib.current_stream_ptr()), "ccx_specgate_reduce")
        return out

    def reduce_long(self, y: torch.Tensor, prop_decrease: float) -> torch.Tensor:
        """One 1-D signal of any length (GPU tensor): noisereduce's chunked path -- threshold from the whole signal, 600000-sample
        chunks with 30000 samples of context (ccx_specgate_reduce_long).  Whole conversations go through this."""
        assert y.is_cuda and y.dtype == torch.float32 and y.dim() == 1 and y.is_contiguous()
        out = torch.empty_like(y)
        self.ctx.check(self.lib.ccx_specgate_reduce_long(self.handle, y.data_ptr(), int(y.numel()), float(prop_decrease), out.data_ptr(),
                                                         _lib.current_stream_ptr()), "ccx_specgate_reduce_long")
        return out

    def __call__(self, y, sr: int = 16000, prop_decrease: float = 1.0) -> np.ndarray:
        if int(sr) != self.sr:
            raise _lib.CcxError(f"SpectralGate was built for {self.sr} Hz, got {sr}")
        x = np.ascontiguousarray(np.asarray(y, dtype=np.float32).reshape(1, -1))
        d = torch.from_numpy(x).to(self.device)
        if x.shape[1] > self.max_samples:
            return self.reduce_long(d[0], prop_decrease).cpu().numpy()
        return self.reduce_batch(d, [x.shape[1]], prop_decrease)[0].cpu().numpy()
