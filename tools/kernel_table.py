"""One sequential front-end + encoder step of the pipeline workload with libccx's per-launch events on: calls, ms, TFLOP/s and
TB/s (algorithmic) per instrumented kernel.  Kernels without a prof scope do not show: tools/step_kernel_diff.sh lists those."""
import os, sys, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clearconverse_amd import _lib
from clearconverse_amd.audio import synthetic_clip
from clearconverse_amd.batch import BatchPipeline
from clearconverse_amd.models import load_models
ctx = _lib.Context(0)
B = 32
audio = torch.from_numpy(np.stack([synthetic_clip(i, 30.0) for i in range(B)])).cuda().contiguous()
models = load_models(None, 0, whisper_batch=192, ctx=ctx, seed=0, seg_max_crops=52 * 32 + 16, seg_max_seconds=300.0 * 32, emb_max_crops=44 * 32, resnet_max_chunks=21 * 32)
bp = BatchPipeline(models, whisper_group=192, sample_len=4)
bp.run_pinned(audio); torch.cuda.synchronize()
ctx.prof_enable(True)
bp.run_pinned(audio); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for name, fl, by, ms in ctx.prof_records():
    a = agg[name]; a[0] += 1; a[1] += fl; a[2] += by; a[3] += ms
print(f"{'kernel':42s} {'n':>6s} {'ms':>8s} {'TF/s':>8s} {'TB/s':>7s}")
for name, (n, fl, by, ms) in sorted(agg.items(), key=lambda kv: -kv[1][3]):
    if ms < 0.3: continue
    print(f"{name[:42]:42s} {n:6d} {ms:8.2f} {fl / ms / 1e9:8.1f} {by / ms / 1e9:7.2f}")
