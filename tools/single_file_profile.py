"""cProfile of EnhancedAudioProcessor.run on one 30 s WAV (the single_file workload of bench.py): where the wall time of the
drop-in path goes, by function (calls into libccx are synchronous on this path, so host time = wall time)."""
import cProfile, os, pstats, sys, tempfile
import torch
from clearconverse_amd import _lib
from clearconverse_amd.audio import SCHEDULE_30S, synthetic_clip, write_wav
from clearconverse_amd.models import load_models
from clearconverse_amd.pipelines import Annotation
from clearconverse_amd.processor import Config, EnhancedAudioProcessor

ctx = _lib.Context(0)
models = dict(load_models(None, 0, whisper_batch=8, ctx=ctx, seed=0, max_audio_seconds=30.0))


class Pinned:
    def __init__(self, inner, tracks):
        self.inner, self.tracks = inner, tracks

    def __call__(self, path, **kw):
        self.inner(path, **kw)
        return Annotation(self.tracks)


models["vad_pipeline"] = Pinned(models["vad_pipeline"], [(0.0, 16.0, "SPEECH"), (18.0, 24.0, "SPEECH"), (26.0, 30.0, "SPEECH")])
models["diarization"] = Pinned(models["diarization"], [(s, e, "SPEAKER_00" if spk == "A" else "SPEAKER_01") for spk, s, e in SCHEDULE_30S])
work = tempfile.mkdtemp(prefix="ccx_sf_")
wav = os.path.join(work, "clip.wav")
write_wav(wav, synthetic_clip(0, 30.0))
proc = EnhancedAudioProcessor(Config(temperature=0.0), load_models_immediately=False, model_loader=lambda cfg, dev: models)
proc._initialize_models()
for _ in range(2):
    proc.run(wav, output_dir=os.path.join(work, "out"))
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(4):
    proc.run(wav, output_dir=os.path.join(work, "out"))
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats("cumulative").print_stats(45)
