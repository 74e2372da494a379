"""CPU: clearconverse_amd.processor / intervals against fixtures produced by EXECUTING the
reference's own glue code (oracle/gen_glue_golden.py, reference back/api.py:294-343, 584-1549) with
the scripted stub models of tests/glue_stubs.py.  Same stubs here -> results must be identical."""
import json
import os
from pathlib import Path

import numpy as np
import pytest
import torch

from clearconverse_amd import intervals as iv
from clearconverse_amd.processor import AudioSegment, Config, EnhancedAudioProcessor
from tests.glue_stubs import (Annotation, Scenario, StubEmbedding, StubSeparator, StubWhisper, result_to_json, scenario_audio)

GOLDEN = Path(__file__).resolve().parent / "golden"
INTERVALS = json.loads((GOLDEN / "glue_intervals.json").read_text())
PROCESS = json.loads((GOLDEN / "glue_process_file.json").read_text())


@pytest.mark.parametrize("idx", range(len(INTERVALS["cases"])))
def test_interval_helpers_match_reference(idx):
    c = INTERVALS["cases"][idx]
    segs = [tuple(x) for x in c["input"]["segments"]]
    vad = [tuple(v) for v in c["input"]["vad"]]
    assert [list(m) for m in iv.merge_diarization_segments(list(segs), c["input"]["gap"])] == c["merged"]
    ov = iv.find_segment_overlaps(list(segs))
    assert sorted([[k[0], k[1], sorted(v)] for k, v in ov.items()]) == c["overlaps"]
    got = [iv.refine_segment_with_vad((s, e), vad) for s, e, _ in segs]
    assert [list(r) if r is not None else None for r in got] == c["refined"]


def _build(sc: Scenario):
    cfg = Config(auth_token="x", **sc.config)
    p = EnhancedAudioProcessor(cfg, load_models_immediately=False)
    p.device = torch.device("cpu")
    audio = scenario_audio(sc)
    p.load_audio = lambda path: (audio.clone(), 16000)
    p.whisper_model = StubWhisper()
    p.separator = StubSeparator()
    p.embedding_model = StubEmbedding()
    p.denoiser = lambda y, sr, prop_decrease: np.asarray(y)        # same identity stub the reference run used for nr.reduce_noise
    p.vad_pipeline = lambda path: Annotation([(s, e, "SPEECH") for s, e in sc.vad])
    calls = []

    def diar(path, min_speakers=None, max_speakers=None):
        calls.append(os.path.basename(str(path)))
        return Annotation(list(sc.secondary if os.path.basename(str(path)) == "temp_segment.wav" else sc.diarization))
    p.diarization = diar
    p.models_loaded = {k: True for k in p.models_loaded}
    return p, calls


def _close(a, b, path=""):
    if isinstance(a, float) or isinstance(b, float):
        assert a == pytest.approx(b, rel=1e-6, abs=1e-7), path
    elif isinstance(a, dict):
        assert set(a) == set(b), (path, set(a) ^ set(b))
        for k in a:
            _close(a[k], b[k], f"{path}.{k}")
    elif isinstance(a, list):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _close(x, y, f"{path}[{i}]")
    else:
        assert a == b, path


@pytest.mark.parametrize("name", sorted(PROCESS["scenarios"]))
def test_process_file_matches_reference(name):
    entry = PROCESS["scenarios"][name]
    sc = Scenario.from_json(entry["scenario"])
    p, diar_calls = _build(sc)
    res = p.process_file("clip.wav")
    got = result_to_json(res, p.whisper_model.calls, p.separator.calls, diar_calls)
    exp = dict(entry["expected"])
    transcript = exp.pop("transcript", None)
    _close(json.loads(json.dumps(got)), exp, name)
    if res is not None:
        assert EnhancedAudioProcessor.format_transcript(res["segments"]) == transcript


def test_config_mirrors_reference_defaults():
    c = Config(auth_token="t")
    assert (c.target_sample_rate, c.min_segment_duration, c.overlap_threshold, c.merge_gap_threshold) == (16000, 0.45, 0.5, 0.5)
    assert (c.max_embedding_segments, c.noise_reduction_amount, c.temperature, c.max_speakers, c.min_speakers) == (100, 0.5, 0.1, 2, 1)
    assert (c.sliding_window_size, c.sliding_window_step, c.secondary_diarization_threshold, c.whisper_model_size) == (0.8, 0.4, 0.3, "small.en")


def test_run_writes_transcript_and_segments(tmp_path):
    sc = Scenario.from_json(PROCESS["scenarios"]["two_speakers_overlap_10s"]["scenario"])
    p, _ = _build(sc)
    seen = []
    src, transcript, path = p.run("clip.wav", output_dir=str(tmp_path / "out"), progress_callback=lambda pct, msg: seen.append(pct))
    assert src == "clip.wav" and os.path.exists(path)
    assert open(path, encoding="utf-8").read() == transcript == PROCESS["scenarios"]["two_speakers_overlap_10s"]["expected"]["transcript"]
    assert seen[0] == 5 and seen[-1] == 100 and 30 in seen and 60 in seen and 80 in seen
    assert len(list((tmp_path / "out" / "overlap_segments").glob("overlap_*.wav"))) == 3


def test_run_returns_none_tuple_when_nothing_detected(tmp_path):
    sc = Scenario.from_json(PROCESS["scenarios"]["no_speakers"]["scenario"])
    p, _ = _build(sc)
    assert p.run("clip.wav", output_dir=str(tmp_path / "o")) == (None, None, None)


def test_extract_segment_edge_cases():
    p = EnhancedAudioProcessor(Config(), load_models_immediately=False)
    p.device = torch.device("cpu")
    a = torch.arange(32000, dtype=torch.float32)[None]
    assert p._extract_segment(a, -1.0, 0.5).shape == (1, 8000)
    assert p._extract_segment(a, 1.5, 9.0).shape == (1, 8000)          # clipped to the 2.0 s clip
    assert p._extract_segment(a, 1.0, 1.0).shape == (1, 100)           # invalid -> zeros(1,100), reference 855-858
    assert p._extract_embedding(torch.zeros(1, 7999)) is None          # < 0.5 s -> no embedding, reference 864-866


class BatchStubWhisper:
    """A whisper model that ALSO offers transcribe_batch (like clearconverse_amd.whisper.WhisperModel): the text is a function of the
    call's audio length and prompt only, so the order of the calls cannot show in the results."""

    def __init__(self):
        self.calls, self.batches = [], []

    @staticmethod
    def _text(n, prompt):
        import zlib
        return f" utt{zlib.crc32(repr((n, prompt)).encode()) % 100000} n{n}."

    def transcribe(self, audio, initial_prompt=None, word_timestamps=False, condition_on_previous_text=True, temperature=0.0, **kw):
        n = int(np.asarray(audio).reshape(-1).shape[0])
        self.calls.append(dict(n_samples=n, initial_prompt=initial_prompt, condition_on_previous_text=bool(condition_on_previous_text),
                               temperature=float(temperature)))
        return {"text": self._text(n, initial_prompt)}

    def transcribe_batch(self, audios, initial_prompts=None, condition_on_previous_text=True, temperature=0.0, **kw):
        self.batches.append(len(audios))
        return [self.transcribe(a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a, initial_prompt=p,
                                condition_on_previous_text=condition_on_previous_text, temperature=temperature)
                for a, p in zip(audios, initial_prompts)]


@pytest.mark.parametrize("name", sorted(PROCESS["scenarios"]))
def test_two_pass_schedule_batched_equals_serial(name):
    """The two-pass schedule of process_file: the Whisper calls recorded by the first pass, run in dependency waves through
    transcribe_batch, give the segments, texts, prompts and transcript of the serial order (reference back/api.py:1378-1530) -- only
    the ORDER of the calls differs.  Prompt-carrying calls (same speaker within 1.0 s: 1425-1426, 1467-1468) wait for their wave."""
    sc = Scenario.from_json(PROCESS["scenarios"][name]["scenario"])
    outs = []
    for batched in (False, True):
        p, diar_calls = _build(sc)
        p.whisper_model = BatchStubWhisper()
        p.batch_whisper_calls = batched
        res = p.process_file("clip.wav")
        outs.append((result_to_json(res, sorted(p.whisper_model.calls, key=repr), p.separator.calls, diar_calls),
                     None if res is None else EnhancedAudioProcessor.format_transcript(res["segments"]), p.whisper_model))
    (a, ta, wa), (b, tb, wb) = outs
    _close(json.loads(json.dumps(b)), json.loads(json.dumps(a)), name)
    assert ta == tb
    assert wa.batches == [] and (len(wb.calls) == 0 or len(wb.batches) >= 1)
    if name == "rapid_exchange_and_prompt_carry":
        carried = [c for c in wb.calls if c["initial_prompt"].startswith("utt")]
        assert len(carried) == 1 and wb.batches == [5, 1]                   # at least one dependent wave
    if name == "two_speakers_one_overlap_30s":
        assert wb.batches == [len(wb.calls)] and len(wb.calls) >= 4          # no prompt depends on a text: ONE batch


def test_two_pass_schedule_overlap_failure_becomes_a_marker_segment():
    """A failing transcription of an overlap region gives the reference's marker segment (back/api.py:1109-1117) in the batched
    schedule too: the batch is re-run call by call, and the failing call alone is marked."""
    sc = Scenario.from_json(PROCESS["scenarios"]["two_speakers_overlap_10s"]["scenario"])
    p, _ = _build(sc)

    class Failing(BatchStubWhisper):
        def transcribe(self, audio, initial_prompt=None, **kw):
            n = int(np.asarray(audio).reshape(-1).shape[0])
            if initial_prompt == "This is a single speaker talking." and n == self.bad:
                raise ValueError("scripted failure")
            return super().transcribe(audio, initial_prompt=initial_prompt, **kw)

        def transcribe_batch(self, audios, initial_prompts=None, **kw):
            if any(int(a.numel()) == self.bad for a in audios):
                raise RuntimeError("batch failed")
            return super().transcribe_batch(audios, initial_prompts, **kw)
    serial, _ = _build(sc)
    serial.whisper_model = BatchStubWhisper()
    serial.batch_whisper_calls = False
    ref = serial.process_file("clip.wav")
    overlap = [s_ for s_ in ref["segments"] if s_.is_overlap]
    p.whisper_model = Failing()
    p.whisper_model.bad = int(overlap[1].audio_tensor.shape[-1])
    res = p.process_file("clip.wav")
    marked = [s_ for s_ in res["segments"] if s_.transcription == "[Processing error]"]
    assert len(marked) >= 1 and all(s_.confidence == 0.0 and s_.is_overlap for s_ in marked)
    assert len(res["segments"]) == len(ref["segments"])
    assert sum(s_.transcription != "[Processing error]" for s_ in res["segments"]) >= 1


def test_sliding_window_embeddings_batched_equal_the_loop():
    """`_resegment_overlap` embeds its windows through `_embed_many`: one `embed_batch` call per overlap segment for a model that offers
    it, the reference's loop (back/api.py:974-977) otherwise.  Same regions, same labels, same separator / Whisper calls either way."""
    class BatchEmbedding(StubEmbedding):
        def __init__(self):
            super().__init__()
            self.batches = []

        def embed_batch(self, crops):
            self.batches.append(len(crops))
            return torch.stack([torch.from_numpy(self({"waveform": c.reshape(1, -1), "sample_rate": 16000})) for c in crops])
    for name in ("two_speakers_one_overlap_30s", "custom_thresholds", "short_overlap_segment"):
        sc = Scenario.from_json(PROCESS["scenarios"][name]["scenario"])
        outs = []
        for batched in (False, True):
            p, diar_calls = _build(sc)
            p.embedding_model = BatchEmbedding()
            p.batch_embeddings = batched
            res = p.process_file("clip.wav")
            outs.append((result_to_json(res, p.whisper_model.calls, p.separator.calls, diar_calls), p.embedding_model.batches))
        (a, ba), (b, bb) = outs
        _close(json.loads(json.dumps(b)), json.loads(json.dumps(a)), name)
        exp = dict(PROCESS["scenarios"][name]["expected"]); exp.pop("transcript", None)
        _close(json.loads(json.dumps(b)), exp, name)                       # ... and both equal the reference-generated fixture
        assert ba == [] and len(bb) >= 1 and max(bb) > 1, (name, bb)
