"""Host-side profile of the VAD / diarization stages of one pinned pipeline step (cProfile, cumulative)."""
import cProfile, pstats, os, sys, io
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clearconverse_amd import _lib
from clearconverse_amd.audio import synthetic_clip
from clearconverse_amd.models import load_models
ctx = _lib.Context(0)
B = 32
clips = [synthetic_clip(i, 30.0) for i in range(B)]
models = load_models(None, 0, whisper_batch=8, ctx=ctx, seed=0)
items = [{"waveform": torch.from_numpy(c).cuda(), "sample_rate": 16000} for c in clips]
for name in ("vad_pipeline", "diarization"):
    kw = dict(min_speakers=1, max_speakers=2) if name == "diarization" else {}
    models[name].batch(items, **kw)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    models[name].batch(items, **kw)
    torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
    print("=====", name)
    print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:4500])
