"""Shape-level GEMM timing table of one pipeline step (CCX_PROF_SHAPES=1): which GEMMs the time goes to."""
import os, sys, collections
os.environ["CCX_PROF_SHAPES"] = "1"
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
bp = BatchPipeline(models, whisper_group=192, sample_len=8)
bp.run_pinned(audio)
torch.cuda.synchronize()
ctx.prof_enable(True)
bp.run_pinned(audio)
torch.cuda.synchronize()
recs = ctx.prof_records()
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for name, fl, by, ms in recs:
    if name.startswith("gemm<"):
        a = agg[name]; a[0] += 1; a[1] += fl; a[2] += ms
tot = sum(a[2] for a in agg.values())
print(f"total gemm ms {tot:.1f}")
for name, (n, fl, ms) in sorted(agg.items(), key=lambda kv: -kv[1][2])[:40]:
    print(f"{ms:8.2f} ms  {n:5d} x  {fl / ms / 1e9:7.1f} TF/s  {name}")
