"""libccx-backed speaker networks.

`XVectorEmbedder` is the drop-in for `self.embedding_model` of the reference
(Inference("pyannote/embedding", window="whole"), /root/reference/back/api.py:776-780): calling it
with {"waveform": Tensor[1,T], "sample_rate": 16000} returns a 1-D numpy embedding, exactly the call at
back/api.py:869-872.  `embed_batch` is the batched entry the reference lacks (ragged crops, one launch
chain).  `SegmentationNet` is the PyanNet the VAD / diarization pipelines run (see pipelines.py)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .weights import sinc_filters


class _SpeakerNet:
    def __init__(self, kind: int, state_dict: Dict[str, torch.Tensor], n_classes: int, powerset: bool, max_crops: int,
                 max_samples: int, device: int, ctx: Optional[_lib.Context]):
        if not torch.cuda.is_available():
            raise _lib.CcxError("speaker networks need a ROCm GPU: the HIP path has no CPU fallback")
        self.device = torch.device("cuda", device)
        self.ctx = ctx or _lib.Context(device)
        self.lib = self.ctx.lib
        self.max_crops, self.max_samples = int(max_crops), int(max_samples)
        h = C.c_void_p()
        self.ctx.check(self.lib.ccx_speaker_create(self.ctx.handle, kind, n_classes, 1 if powerset else 0, self.max_crops,
                                                   self.max_samples, C.byref(h)), "ccx_speaker_create")
        self.handle = h
        sd = dict(state_dict)
        lo, band = sd.pop("sincnet.conv1d.0.filterbank.low_hz_"), sd.pop("sincnet.conv1d.0.filterbank.band_hz_")
        sd["sincnet.conv1d.0.filters"] = sinc_filters(lo, band)
        for name, t in sd.items():
            if not torch.is_tensor(t) or name in ("powerset",):
                continue
            t = t.detach().to("cpu", torch.float32).contiguous()
            self.ctx.check(self.lib.ccx_speaker_set_tensor(self.handle, name.encode(), t.data_ptr(), t.numel()), f"set_tensor({name})")
        self.ctx.check(self.lib.ccx_speaker_finalize(self.handle), "ccx_speaker_finalize")

    def close(self):
        if getattr(self, "handle", None):
            self.lib.ccx_speaker_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _groups(self, crops: Sequence):
        """Split a crop list into launches that respect both capacities (crop count and total samples)."""
        i0, n = 0, len(crops)
        while i0 < n:
            i1, tot = i0, 0
            while i1 < n and i1 - i0 < self.max_crops:
                k = int(crops[i1].numel() if torch.is_tensor(crops[i1]) else np.asarray(crops[i1]).size)
                if tot + k > 0.9 * self.max_samples and i1 > i0:
                    break
                tot += k
                i1 += 1
            yield i0, i1
            i0 = i1

    def _pack(self, crops: Sequence):
        """Concatenate crops (1-D tensors/arrays, any device) into one device buffer + offset table.
        Host-resident crops are concatenated on the host first (ONE upload instead of one per crop)."""
        ts = [c.reshape(-1) if torch.is_tensor(c) else torch.from_numpy(np.asarray(c, dtype=np.float32)).reshape(-1) for c in crops]
        lens = [int(t.numel()) for t in ts]
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
        if all(not t.is_cuda for t in ts):
            buf = (torch.cat(ts) if len(ts) > 1 else ts[0]).to(torch.float32).to(self.device)
        else:
            ts = [t.to(self.device, torch.float32) for t in ts]
            buf = torch.cat(ts) if len(ts) > 1 else ts[0].contiguous()
        return buf, offs, np.asarray(lens, dtype=np.int32)


class XVectorEmbedder(_SpeakerNet):
    DIM = 512

    def __init__(self, state_dict, max_crops: int = 256, max_samples: int = 16000 * 600, device: int = 0, ctx=None):
        super().__init__(0, state_dict, 0, True, max_crops, max_samples, device, ctx)

    def embed_batch(self, crops: Sequence, weights: Optional[Sequence] = None) -> torch.Tensor:
        """crops: list of 1-D waveforms (16 kHz) -> [n, 512] f32 tensor on the GPU.  `weights`
        (optional): one 1-D frame-weight vector per crop (any resolution) for weighted statistics pooling."""
        out = torch.empty(len(crops), self.DIM, device=self.device, dtype=torch.float32)
        i64p, ip = C.POINTER(C.c_int64), C.POINTER(C.c_int)
        for i0, i1 in self._groups(crops):
            part = crops[i0:i1]
            buf, offs, lens = self._pack(part)
            o = out[i0:i1]
            if weights is None:
                wptr, woffs, wlens = None, None, None
            else:
                wbuf, wo, wl = self._pack(weights[i0:i1])
                wptr, woffs, wlens = wbuf.data_ptr(), wo.ctypes.data_as(i64p), wl.ctypes.data_as(ip)
            self.ctx.check(self.lib.ccx_speaker_embed(self.handle, buf.data_ptr(), offs.ctypes.data_as(i64p), lens.ctypes.data_as(ip),
                                                      len(part), wptr, woffs, wlens, o.data_ptr(), _lib.current_stream_ptr()),
                           "ccx_speaker_embed")
        return out

    def __call__(self, item: dict) -> np.ndarray:
        """pyannote Inference(window='whole') call shape: {'waveform': [1,T], 'sample_rate': sr} -> np [512]."""
        if int(item.get("sample_rate", 16000)) != 16000:
            raise _lib.CcxError("XVectorEmbedder expects 16 kHz input (the reference always passes target_sample_rate)")
        w = item["waveform"]
        return self.embed_batch([w.reshape(-1)])[0].cpu().numpy()


class ResNetEmbedder:
    """WeSpeaker ResNet-34 (`pyannote/wespeaker-voxceleb-resnet34-LM`), the embedding model inside the
    reference's `pyannote/speaker-diarization-3.1` pipeline (/root/reference/back/api.py:788-792, called at
    1056-1060 and 1124-1128).  `embed_chunks` embeds equal-length chunks; with `weights` it returns one
    embedding per (chunk, mask) pair while the convolutional trunk runs once per chunk."""
    DIM = 256

    def __init__(self, state_dict: Dict[str, torch.Tensor], max_chunks: int = 96, max_samples: int = 160000, max_masks: int = 512,
                 device: int = 0, ctx: Optional[_lib.Context] = None):
        if not torch.cuda.is_available():
            raise _lib.CcxError("speaker networks need a ROCm GPU: the HIP path has no CPU fallback")
        self.device = torch.device("cuda", device)
        self.ctx = ctx or _lib.Context(device)
        self.lib = self.ctx.lib
        self.max_chunks, self.max_samples, self.max_masks = int(max_chunks), int(max_samples), int(max_masks)
        h = C.c_void_p()
        self.ctx.check(self.lib.ccx_resnet_create(self.ctx.handle, self.max_chunks, self.max_samples, self.max_masks, C.byref(h)),
                       "ccx_resnet_create")
        self.handle = h
        for name, t in state_dict.items():
            if not torch.is_tensor(t) or name.endswith("num_batches_tracked"):
                continue
            t = t.detach().to("cpu", torch.float32).contiguous()
            self.ctx.check(self.lib.ccx_resnet_set_tensor(self.handle, name.encode(), t.data_ptr(), t.numel()), f"set_tensor({name})")
        self.ctx.check(self.lib.ccx_resnet_finalize(self.handle), "ccx_resnet_finalize")

    def close(self):
        if getattr(self, "handle", None):
            self.lib.ccx_resnet_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def embed_chunks(self, chunks: torch.Tensor, weights: Optional[torch.Tensor] = None,
                     mask_chunk: Optional[Sequence[int]] = None) -> torch.Tensor:
        """chunks [n, N] f32 (16 kHz, equal lengths).  weights None -> [n, 256]; else weights [m, n_w] frame weights
        and mask_chunk [m] (chunk index of each mask) -> [m, 256]."""
        chunks = chunks.to(self.device, torch.float32).contiguous()
        n, N = int(chunks.shape[0]), int(chunks.shape[1])
        if weights is None:
            out = torch.empty(n, self.DIM, device=self.device, dtype=torch.float32)
            for i0 in range(0, n, min(self.max_chunks, self.max_masks)):
                part = chunks[i0:i0 + min(self.max_chunks, self.max_masks)]
                self.ctx.check(self.lib.ccx_resnet_embed(self.handle, part.data_ptr(), N, N, int(part.shape[0]), None, 0, None, 0,
                                                         out[i0:].data_ptr(), _lib.current_stream_ptr()), "ccx_resnet_embed")
            return out
        weights = weights.to(self.device, torch.float32).contiguous()
        mc = np.asarray(mask_chunk, dtype=np.int64)
        if mc.shape[0] != weights.shape[0]:
            raise _lib.CcxError("embed_chunks: one mask_chunk entry per weight row is required")
        if mc.size and (np.any(np.diff(mc) < 0) or mc.min() < 0 or mc.max() >= n):
            raise _lib.CcxError("embed_chunks: mask_chunk must be sorted and index existing chunks")
        m, n_w = int(weights.shape[0]), int(weights.shape[1])
        out = torch.empty(m, self.DIM, device=self.device, dtype=torch.float32)
        ip = C.POINTER(C.c_int)
        c0 = 0
        while c0 < n:
            # chunks [c0, c1) and their masks [j0, j1), both within capacity
            c1 = min(n, c0 + self.max_chunks)
            j0 = int(np.searchsorted(mc, c0, "left"))
            while c1 > c0 + 1 and int(np.searchsorted(mc, c1, "left")) - j0 > self.max_masks:
                c1 -= 1
            j1 = int(np.searchsorted(mc, c1, "left"))
            if j1 - j0 > self.max_masks:
                raise _lib.CcxError("embed_chunks: a single chunk has more masks than max_masks")
            if j1 > j0:
                part = chunks[c0:c1]
                local = (mc[j0:j1] - c0).astype(np.int32)
                self.ctx.check(self.lib.ccx_resnet_embed(self.handle, part.data_ptr(), N, N, c1 - c0, weights[j0:j1].data_ptr(), n_w,
                                                         local.ctypes.data_as(ip), j1 - j0, out[j0:j1].data_ptr(),
                                                         _lib.current_stream_ptr()), "ccx_resnet_embed")
            c0 = c1
        return out


class SegmentationNet(_SpeakerNet):
    """PyanNet: frames of per-class scores for each crop (SincNet frame rate: 1 frame per 270 samples)."""

    def __init__(self, state_dict, n_classes: int = 7, powerset: bool = True, max_crops: int = 256,
                 max_samples: int = 16000 * 600, device: int = 0, ctx=None):
        super().__init__(1, state_dict, n_classes, powerset, max_crops, max_samples, device, ctx)
        self.n_classes = n_classes
        self.powerset = bool(powerset)

    def segment_launch(self, crops: Sequence) -> list:
        """Queue the network over all launch groups of `crops`; nothing is copied back yet (see segment_fetch)."""
        i64p, ip = C.POINTER(C.c_int64), C.POINTER(C.c_int)
        pending = []
        for i0, i1 in self._groups(crops):
            part = crops[i0:i1]
            buf, offs, lens = self._pack(part)
            cap = int(sum(lens) // 270 + 4 * len(part))
            out = torch.empty(cap, self.n_classes, device=self.device, dtype=torch.float32)
            frames = np.zeros(len(part), dtype=np.int32)
            self.ctx.check(self.lib.ccx_speaker_segment(self.handle, buf.data_ptr(), offs.ctypes.data_as(i64p), lens.ctypes.data_as(ip),
                                                        len(part), out.data_ptr(), cap, frames.ctypes.data_as(ip), _lib.current_stream_ptr()),
                           "ccx_speaker_segment")
            pending.append((out, frames, buf))          # buf stays referenced until its kernels have run
        return pending

    @staticmethod
    def segment_fetch(pending: list) -> List[np.ndarray]:
        """Device->host copies (one per launch group) of a segment_launch: per-crop [frames, classes] host arrays."""
        res: List[np.ndarray] = []
        for out, frames, _ in pending:
            host = out[: int(frames.sum())].cpu().numpy()
            r = 0
            for f in frames:
                res.append(host[r:r + int(f)])
                r += int(f)
        return res

    def segment_numpy(self, crops: Sequence) -> List[np.ndarray]:
        """Like segment_batch but returns host arrays.  All launch groups are queued first; the device->host copies
        (one per group) follow, so the GPU never waits for the host to pack the next group."""
        return self.segment_fetch(self.segment_launch(crops))

    def segment_batch(self, crops: Sequence) -> List[torch.Tensor]:
        outs: List[torch.Tensor] = []
        for i0, i1 in self._groups(crops):
            part = crops[i0:i1]
            buf, offs, lens = self._pack(part)
            cap = int(sum(lens) // 270 + 4 * len(part))
            out = torch.empty(cap, self.n_classes, device=self.device, dtype=torch.float32)
            frames = np.zeros(len(part), dtype=np.int32)
            self.ctx.check(self.lib.ccx_speaker_segment(self.handle, buf.data_ptr(), offs.ctypes.data_as(C.POINTER(C.c_int64)),
                                                        lens.ctypes.data_as(C.POINTER(C.c_int)), len(part), out.data_ptr(), cap,
                                                        frames.ctypes.data_as(C.POINTER(C.c_int)), _lib.current_stream_ptr()),
                           "ccx_speaker_segment")
            r = 0
            for f in frames:
                outs.append(out[r:r + int(f)])
                r += int(f)
        return outs
