"""-m gpu: the model set wired together -- pipelines, processor.run on a WAV file, and the clip-batched
pinned-schedule driver -- all on libccx objects (reduced Whisper/SepFormer depth for speed; same kernels)."""
import os

import numpy as np
import pytest
import torch

from clearconverse_amd.audio import synthetic_clip, write_wav
from clearconverse_amd.weights import SepDims, WhisperDims

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def models(ccx_ctx):
    from clearconverse_amd.models import load_models
    m = load_models(None, 0, whisper_batch=16, ctx=ccx_ctx, whisper_dims=WhisperDims.mini(2, 128), sep_dims=SepDims(n_layers=2),
                    sep_tokens=60_000, max_crops=128)
    yield m


def test_vad_and_diarization_return_annotations(models, tmp_path):
    clip = synthetic_clip(0, 30.0)
    path = str(tmp_path / "clip.wav")
    write_wav(path, clip)
    vad = models["vad_pipeline"](path)
    diar = models["diarization"](path, min_speakers=1, max_speakers=2)
    for ann in (vad, diar):
        last = -1.0
        for seg, track, label in ann.itertracks(yield_label=True):
            assert 0.0 <= seg.start < seg.end <= 30.0 + 1e-6 and isinstance(label, str)
            assert seg.start >= last
            last = seg.start
    assert len(set(l for _, _, l in diar.itertracks(yield_label=True))) <= 2
    # same result from an in-memory waveform (the batch driver's entry)
    vad2 = models["vad_pipeline"]({"waveform": torch.from_numpy(clip), "sample_rate": 16000})
    a = [(round(s.start, 3), round(s.end, 3)) for s, _, _ in vad.itertracks(yield_label=True)]
    b = [(round(s.start, 3), round(s.end, 3)) for s, _, _ in vad2.itertracks(yield_label=True)]
    assert len(a) == len(b) and all(abs(x[0] - y[0]) < 0.02 and abs(x[1] - y[1]) < 0.02 for x, y in zip(a, b))   # 16-bit WAV quantisation


@pytest.fixture(scope="module")
def scripted_models(ccx_ctx):
    """The same model set with SCRIPTED segmentation weights (tests/scripted_nets.py, fitted to synthetic_clip(1, 30 s)): VAD and
    diarization then follow the clip's schedule (A 0-9 s, B 7-16 s, A 18-24 s, B 26-30 s), so `process_file` is guaranteed
    speech regions, two diarization labels and an overlap -- a transcript must come out."""
    from clearconverse_amd.models import build_state_dicts, load_models
    from tests.scripted_nets import scripted_pyannet_state_dict
    sds = build_state_dicts(None, whisper_dims=WhisperDims.mini(2, 128), sep_dims=SepDims(n_layers=2), seed=0)
    sds["pyannet_diar"], _ = scripted_pyannet_state_dict(1, 7, True)
    sds["pyannet_vad"], _ = scripted_pyannet_state_dict(1, 3, False, window_s=5.0, seed=4)
    m = load_models(None, 0, whisper_batch=16, ctx=ccx_ctx, state_dicts=sds, sep_tokens=60_000, max_crops=128)
    yield m
    for k in ("whisper_model", "separator", "embedding_model", "diarization_embedder", "segmentation_vad", "segmentation_diar", "denoiser"):
        m[k].close()


@pytest.mark.parametrize("temperature", [0.0, 0.1])      # 0.1 = the reference's Config default (back/api.py:128)
def test_processor_run_on_wav_file(scripted_models, tmp_path, temperature):
    """`EnhancedAudioProcessor.run` on a 30 s WAV (reference back/api.py:1204-1280): the transcript format of 1252-1265 is asserted
    unconditionally -- the scripted segmentation weights guarantee speech, and min_speakers = max_speakers = 2 two diarization
    labels (num_clusters is then forced to 2 whatever the seeded ResNet embeddings look like)."""
    import re
    from clearconverse_amd.processor import Config, EnhancedAudioProcessor
    clip = synthetic_clip(1, 30.0)
    path = str(tmp_path / "clip.wav")
    write_wav(path, clip)
    p = EnhancedAudioProcessor(Config(temperature=temperature, min_speakers=2, max_speakers=2), load_models_immediately=False,
                               model_loader=lambda cfg, dev: scripted_models)
    seen = []
    out = p.run(path, output_dir=str(tmp_path / "out"), progress_callback=lambda pct, msg: seen.append(pct))
    assert seen[:1] == [5] and 30 in seen
    assert isinstance(out, tuple) and len(out) == 3
    assert out[0] == path and out[1] is not None and os.path.exists(out[2])
    pat = r"\[(SPEAKER_A|SPEAKER_B|UNKNOWN)\] (\d+\.\d\d)s - (\d+\.\d\d)s\n([^\n]*)\n\n"      # f"[{spk}] {start:.2f}s - {end:.2f}s\n{text}\n\n"
    assert re.fullmatch(f"(?:{pat})+", out[1]), out[1]
    blocks = re.findall(pat, out[1])
    assert len(blocks) >= 2 and any(t.strip() for _, _, _, t in blocks)
    speakers = set()
    for spk, t0, t1, _ in blocks:
        assert 0.0 <= float(t0) < float(t1) <= 31.0
        speakers.add(spk)
    assert {"SPEAKER_A", "SPEAKER_B"} <= speakers, speakers
    with open(out[2], encoding="utf-8") as f:
        assert f.read() == out[1]
    res = p.process_file(path)
    assert res["metadata"]["total_segments"] == len(res["segments"]) >= 2 and abs(res["metadata"]["duration"] - 30.0) < 1e-3
    audio, sr = p.load_audio(path)                  # A3 on the HIP spectral gate: peak-normalised
    assert sr == 16000 and audio.shape == (1, 480000) and abs(float(audio.abs().max()) - 1.0) < 1e-4


def test_two_pass_schedule_batched_whisper_calls_equal_the_serial_order(scripted_models, tmp_path, monkeypatch):
    """process_file records its Whisper calls and runs them in dependency waves through WhisperModel.transcribe_batch (one decode
    batch per wave; a prompt that carries the previous segment's text -- reference back/api.py:1425-1426, 1467-1468 -- waits for its
    wave).  At temperature 0 the transcript, every segment's text, speaker, times and audio must be exactly those of the serial order
    (batch_whisper_calls = False: one B = 1 decode per call, the reference's sequence): a window's tokens do not depend on its batch
    mates (tests/test_whisper_gpu.py), so this holds bit for bit."""
    from clearconverse_amd.processor import Config, EnhancedAudioProcessor
    # two clips: the pinned 30 s schedule (no call depends on another: ONE wave) and a dialogue whose turns follow each other within
    # a second (scripted segmentation weights fitted to clip 1: whatever they give on the second clip, both schedules must agree)
    clips = [synthetic_clip(1, 30.0), np.concatenate([synthetic_clip(1, 30.0)[16000 * 18:16000 * 24], synthetic_clip(1, 30.0)[16000 * 18:16000 * 30]])]
    for ci, clip in enumerate(clips):
        path = str(tmp_path / f"clip{ci}.wav")
        write_wav(path, clip)
        outs = []
        for batched in (False, True):
            p = EnhancedAudioProcessor(Config(temperature=0.0, min_speakers=1 + (ci == 0), max_speakers=2), load_models_immediately=False,
                                       model_loader=lambda cfg, dev: scripted_models)
            p._initialize_models()
            p.batch_whisper_calls = batched
            p.batch_embeddings = batched              # the sliding-window embeddings of an overlap segment as one batch / one by one
            calls = []
            wm = scripted_models["whisper_model"]
            orig = wm.transcribe_batch
            monkeypatch.setattr(wm, "transcribe_batch", lambda audios, prompts=None, **kw: (calls.append(len(audios)), orig(audios, prompts, **kw))[1])
            res = p.process_file(path)
            monkeypatch.undo()
            assert res is not None and len(res["segments"]) >= 1
            outs.append((res, calls))
        (a, ca), (b, cb) = outs
        assert EnhancedAudioProcessor.format_transcript(a["segments"]) == EnhancedAudioProcessor.format_transcript(b["segments"]), ci
        for x, y in zip(a["segments"], b["segments"]):
            assert (x.start, x.end, x.speaker_id, x.is_overlap, x.transcription, x.confidence) == (y.start, y.end, y.speaker_id, y.is_overlap, y.transcription, y.confidence)
            assert torch.equal(x.audio_tensor, y.audio_tensor)
        n_calls = sum(ca)
        assert all(c == 1 for c in ca) and sum(cb) == n_calls and len(cb) <= len(ca)       # WhisperModel.transcribe goes through transcribe_batch
        if ci == 0:
            assert len(cb) == 1 and cb[0] >= 4, cb                                        # the pinned schedule: one wave for all calls
        print(f"clip {ci}: {n_calls} Whisper calls, serial {len(ca)} decodes, two-pass schedule {cb}")


def test_pinned_batch_driver_counts(models):
    from clearconverse_amd.batch import BatchPipeline
    bp = BatchPipeline(models, whisper_group=16, sample_len=8)
    audio = torch.from_numpy(np.stack([synthetic_clip(i, 30.0) for i in range(2)])).cuda()
    r = bp.run_pinned(audio, timed=True)
    assert r["n_clips"] == 2 and r["audio_seconds"] == 60.0
    assert r["whisper_calls"] == 2 * (2 + 4)               # 2 regular segments + 4 overlap regions per clip
    assert r["separator_calls"] == 2 * 4
    n_win = 2 * 2 * 21                                     # (9 - 0.8) / 0.4 + 1 windows per overlap segment
    assert r["embeds"] == 2 * 4 + 2 * 2 + n_win + 2 * 2 * 4
    assert len(r["records"]) == r["whisper_calls"] and all(len(x["tokens"]) <= 8 for x in r["records"])
    assert all(np.isfinite(s) for s in r["sims"])
    assert set(bp.stage_ms) >= {"load_audio_gate", "vad", "diarization", "profiles", "separate", "whisper"}


def test_pipelined_schedule_equals_sequential(ccx_ctx):
    """batch.run_pinned_pipelined (front end of batch i + 1 on a second stream while batch i decodes on a worker thread, two
    Whisper instances alternating, several Whisper groups per batch) must return exactly what run_pinned returns batch by batch."""
    from clearconverse_amd.batch import BatchPipeline
    from clearconverse_amd.models import load_models
    m = load_models(None, 0, whisper_batch=8, ctx=ccx_ctx, whisper_dims=WhisperDims.mini(2, 128), sep_dims=SepDims(n_layers=2),
                    sep_tokens=60_000, max_crops=128, whisper_instances=2)
    bp = BatchPipeline(m, whisper_group=8, sample_len=6)            # 12 windows per 2-clip batch -> two groups per batch
    batches = [torch.from_numpy(np.stack([synthetic_clip(10 * k + i, 30.0) for i in range(2)])).cuda() for k in range(3)]
    seq = [bp.run_pinned(a, debug=True) for a in batches]
    for span in (1, 1, 2, 3, 4):                                    # span 1 twice: the second pass replays captured graphs
        # span 2: decode units of two batches (24 windows -> 3 groups of 8) + a trailing unit of one batch; span 3: one unit;
        # span 4 (bench.py's default): more than there are batches -> the same single unit
        pip = bp.run_pinned_pipelined(batches, debug=True, span=span)
        assert len(pip) == len(seq)
        for a, b in zip(seq, pip):
            assert [r["tokens"] for r in a["records"]] == [r["tokens"] for r in b["records"]]
            assert [r["sum_logprob"] for r in a["records"]] == [r["sum_logprob"] for r in b["records"]]
            assert a["sims"] == b["sims"] and a["pick"] == b["pick"] and a["prompt_ids"] == b["prompt_ids"]
            assert torch.equal(a["window_sims_full"], b["window_sims_full"]) and torch.equal(a["separated"], b["separated"])
            for k in ("whisper_calls", "separator_calls", "embeds", "vad_regions", "diar_turns"):
                assert a[k] == b[k]
    for k in ("separator", "embedding_model", "diarization_embedder", "segmentation_vad", "segmentation_diar", "denoiser"):
        m[k].close()
    for w in m["whisper_models"]:
        w.close()


def test_processor_run_on_a_45_second_wav(models, tmp_path):
    """A conversation longer than one Whisper window / one spectral-gate chunk through EnhancedAudioProcessor.run: load_audio gates
    the whole file by the chunked path (2 chunks), the pipelines slide over it, and a turn longer than 30 s is transcribed window
    by window.  (The reference handles arbitrary lengths: noisereduce chunks, whisper seeks.)"""
    from clearconverse_amd.processor import Config, EnhancedAudioProcessor
    clip = np.concatenate([synthetic_clip(3, 30.0), 0.8 * synthetic_clip(4, 30.0)[: 16000 * 15]])
    path = str(tmp_path / "long.wav")
    write_wav(path, clip)
    p = EnhancedAudioProcessor(Config(temperature=0.0), load_models_immediately=False, model_loader=lambda cfg, dev: models)
    p._initialize_models()
    audio, sr = p.load_audio(path)
    assert audio.shape == (1, 45 * 16000) and abs(float(audio.abs().max()) - 1.0) < 1e-4
    out = p.run(path, output_dir=str(tmp_path / "out"))
    assert isinstance(out, tuple) and len(out) == 3
    # a single 40 s turn straight through the Whisper call surface: two windows
    res = models["whisper_model"].transcribe(audio[0, : 16000 * 40].cpu().numpy(), initial_prompt="This is a conversation between two people.")
    assert len({s["seek"] for s in res["segments"]}) >= 2
