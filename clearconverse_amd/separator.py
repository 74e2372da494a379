"""`SepformerSeparator`: drop-in for the `self.separator` object of the reference
(SepformerSeparation.from_hparams at /root/reference/back/api.py:713-717; called at 1077 as
`separated = self.separator.separate_batch(subsegment)` with a [1, T] tensor, result indexed
`separated[..., idx]`).  All arithmetic runs in libccx (csrc/sepformer.hip)."""
from __future__ import annotations

import ctypes as C
from dataclasses import asdict
from typing import Dict, Optional, Sequence

import torch

from . import _lib
from .weights import SepDims


class SepformerSeparator:
    def __init__(self, dims: SepDims, state_dict: Dict[str, torch.Tensor], max_tokens: int = 160_000, max_utts: int = 64,
                 device: int = 0, ctx: Optional[_lib.Context] = None):
        if not torch.cuda.is_available():
            raise _lib.CcxError("SepformerSeparator needs a ROCm GPU: the HIP path has no CPU fallback")
        self.dims = dims
        self.max_tokens, self.max_utts = int(max_tokens) // dims.segment * dims.segment, int(max_utts)
        self.device = torch.device("cuda", device)
        self.ctx = ctx or _lib.Context(device)
        self.lib = self.ctx.lib
        h = C.c_void_p()
        cd = _lib.SepformerDims(**asdict(dims))
        self.ctx.check(self.lib.ccx_sepformer_create(self.ctx.handle, C.byref(cd), int(max_tokens), int(max_utts), C.byref(h)),
                       "ccx_sepformer_create")
        self.handle = h
        for name, t in state_dict.items():
            t = t.detach().to("cpu", torch.float32).contiguous()
            self.ctx.check(self.lib.ccx_sepformer_set_tensor(self.handle, name.encode(), t.data_ptr(), t.numel()), f"set_tensor({name})")
        self.ctx.check(self.lib.ccx_sepformer_finalize(self.handle), "ccx_sepformer_finalize")

    def tokens_for(self, n_samples: int) -> int:
        """Encoder frames of one utterance after padding to whole chunks (capacity accounting)."""
        d = self.dims
        L = (int(n_samples) - d.kernel) // d.stride + 1
        return (L + (d.segment - L % d.segment))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.ccx_sepformer_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def separate_batch(self, mix: torch.Tensor, n_samples: Optional[Sequence[int]] = None) -> torch.Tensor:
        """mix [B, T] -> [B, T, 2].  Rows are independent utterances; `n_samples` gives their true
        lengths when the batch is ragged (zero padded to T)."""
        if mix.dim() == 1:
            mix = mix.unsqueeze(0)
        x = mix.to(self.device, torch.float32).contiguous()
        B, T = x.shape
        ns = (C.c_int * B)(*([T] * B if n_samples is None else [int(v) for v in n_samples]))
        out = torch.empty(B, T, 2, device=self.device, dtype=torch.float32)
        self.ctx.check(self.lib.ccx_sepformer_separate(self.handle, x.data_ptr(), T, ns, B, out.data_ptr(), _lib.current_stream_ptr()),
                       "ccx_sepformer_separate")
        return out
