"""Timeline of the software-pipelined schedule (batch.run_pinned_pipelined) at the bench's shape: per batch when its front end ran
(stream events and host issue time), per decode group when its encoder and its decode ran.
Usage on the GPU box: python tools/pipeline_trace.py [n_batches [span]]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from clearconverse_amd import _lib
from clearconverse_amd.audio import synthetic_clip
from clearconverse_amd.batch import BatchPipeline
from clearconverse_amd.models import build_state_dicts, load_models

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
span = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B = 32
grp = 6 * B * span
ctx = _lib.Context(0)
models = load_models(None, 0, whisper_batch=grp, ctx=ctx, seed=0, state_dicts=build_state_dicts(None, seed=0), seg_max_crops=52 * B + 16,
                     seg_max_seconds=300.0 * B, emb_max_crops=44 * B, resnet_max_chunks=21 * B, whisper_instances=2, max_audio_seconds=30.0,
                     gate_max_clips=B)
bp = BatchPipeline(models, whisper_group=grp, sample_len=224)
audio = torch.from_numpy(np.stack([synthetic_clip(i, 30.0) for i in range(B)])).cuda().contiguous()
bp.run_pinned(audio)
bp.run_pinned_pipelined([audio] * span, span=span)
if n % span:
    bp.run_pinned_pipelined([audio] * (n % span), span=span)
torch.cuda.synchronize()
t0 = time.perf_counter()
bp.run_pinned_pipelined([audio] * n, span=span)
torch.cuda.synchronize()
print(f"untraced: {n} batches in {(time.perf_counter() - t0) * 1e3:.0f} ms = {(time.perf_counter() - t0) * 1e3 / n:.1f} ms per batch")
bp.trace = []
t0 = time.perf_counter()
bp.run_pinned_pipelined([audio] * n, span=span)
torch.cuda.synchronize()
print(f"traced:   {n} batches in {(time.perf_counter() - t0) * 1e3:.0f} ms")
for what, u, a, b in sorted(bp.trace, key=lambda r: r[2]):
    print(f"{what:10s} unit {u:2d}: {(a - t0) * 1e3:8.1f} -> {(b - t0) * 1e3:8.1f}  ({(b - a) * 1e3:7.1f} ms)")
