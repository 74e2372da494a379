"""Host-side profile (cProfile) of one pinned pipeline step, or of the VAD / diarization stages alone:
python tools/stage_profile.py [step|stages]"""
import cProfile, pstats, os, sys, io
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clearconverse_amd import _lib
from clearconverse_amd.audio import synthetic_clip
from clearconverse_amd.batch import BatchPipeline
from clearconverse_amd.models import load_models

mode = sys.argv[1] if len(sys.argv) > 1 else "step"
ctx = _lib.Context(0)
B = 32
clips = [synthetic_clip(i, 30.0) for i in range(B)]


def report(pr, title, n=28):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(n)
    print("=====", title)
    print("\n".join(l[:150] for l in s.getvalue().splitlines() if l.strip()))


if mode == "stages":
    models = load_models(None, 0, whisper_batch=8, ctx=ctx, seed=0, seg_max_crops=52 * 32 + 16, seg_max_seconds=300.0 * 32, emb_max_crops=44 * 32, resnet_max_chunks=21 * 32)
    items = [{"waveform": torch.from_numpy(c).cuda(), "sample_rate": 16000} for c in clips]
    for name in ("vad_pipeline", "diarization"):
        kw = dict(min_speakers=1, max_speakers=2) if name == "diarization" else {}
        models[name].batch(items, **kw)
        torch.cuda.synchronize()
        pr = cProfile.Profile(); pr.enable()
        models[name].batch(items, **kw)
        torch.cuda.synchronize()
        pr.disable()
        report(pr, name, 18)
else:
    models = load_models(None, 0, whisper_batch=192, ctx=ctx, seed=0, seg_max_crops=52 * 32 + 16, seg_max_seconds=300.0 * 32, emb_max_crops=44 * 32, resnet_max_chunks=21 * 32)
    audio = torch.from_numpy(np.stack(clips)).cuda().contiguous()
    bp = BatchPipeline(models, whisper_group=192, sample_len=8)     # short decode: the host side of the other stages is the subject
    bp.run_pinned(audio)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    bp.run_pinned(audio)
    torch.cuda.synchronize()
    pr.disable()
    report(pr, "run_pinned (sample_len 8)")
