"""Scripted stand-ins for the five model objects + the scenario list, shared by
oracle/gen_glue_golden.py (which drives the REFERENCE's glue code with them) and
tests/test_processor_glue.py (which drives clearconverse_amd.processor with the very same stubs
and compares against the committed JSON).  Everything here is this repo's own code.

Audio encodes who is speaking so the stub embedder/separator can steer the control flow
deterministically: speaker A is a period-4 cosine, speaker B a period-8 cosine (absolute sample
phase), overlaps are their sum.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

SR = 16000


class _Seg:
    def __init__(self, s, e):
        self.start, self.end = float(s), float(e)


class Annotation:
    """Minimal pyannote.core.Annotation look-alike: only itertracks(yield_label=True) is used by the
    reference (back/api.py:310, 882-883, 896-897, 1130-1131, 1324-1325)."""

    def __init__(self, tracks: List[Tuple[float, float, str]]):
        self.tracks = list(tracks)

    def itertracks(self, yield_label: bool = False):
        for i, (s, e, lab) in enumerate(self.tracks):
            yield (_Seg(s, e), f"T{i}", lab) if yield_label else (_Seg(s, e), f"T{i}")


@dataclass
class Scenario:
    name: str
    duration: float
    truth: List[Tuple[str, float, float]]                  # who really speaks when (drives the audio)
    diarization: List[Tuple[float, float, str]]            # what the scripted diarizer reports
    vad: List[Tuple[float, float]]
    secondary: List[Tuple[float, float, str]] = field(default_factory=list)   # answer for temp_segment.wav
    config: Dict[str, Any] = field(default_factory=dict)

    def to_json(self):
        return dict(name=self.name, duration=self.duration, truth=[list(t) for t in self.truth],
                    diarization=[list(t) for t in self.diarization], vad=[list(t) for t in self.vad],
                    secondary=[list(t) for t in self.secondary], config=self.config)

    @staticmethod
    def from_json(d):
        return Scenario(d["name"], d["duration"], [tuple(t) for t in d["truth"]], [tuple(t) for t in d["diarization"]],
                        [tuple(t) for t in d["vad"]], [tuple(t) for t in d["secondary"]], dict(d["config"]))


def scenario_audio(sc: Scenario) -> torch.Tensor:
    n = int(round(sc.duration * SR))
    t = np.arange(n, dtype=np.float64)
    x = np.zeros(n, dtype=np.float64)
    for spk, s, e in sc.truth:
        i0, i1 = int(s * SR), min(n, int(e * SR))
        if spk == "A":
            x[i0:i1] += 0.4 * np.cos(np.pi * t[i0:i1] / 2.0)
        elif spk == "B":
            x[i0:i1] += 0.3 * np.cos(np.pi * t[i0:i1] / 4.0)
        else:  # a third voice: period 16
            x[i0:i1] += 0.35 * np.cos(np.pi * t[i0:i1] / 8.0)
    x += 0.001 * np.cos(0.013 * t)   # tiny floor so variances are never exactly zero
    return torch.tensor(x[None], dtype=torch.float32)


class StubEmbedding:
    """pyannote Inference(window='whole') look-alike: dict(waveform [1,T], sample_rate) -> np [8]."""

    def __init__(self):
        self.calls = 0

    def __call__(self, d):
        self.calls += 1
        w = d["waveform"]
        x = (w.detach().cpu().numpy() if isinstance(w, torch.Tensor) else np.asarray(w)).reshape(-1).astype(np.float64)
        t = np.arange(x.size, dtype=np.float64)
        out = np.zeros(8, dtype=np.float64)
        for j, per in enumerate((4.0, 8.0, 16.0)):
            c = np.dot(x, np.cos(2 * np.pi * t / per))
            s = np.dot(x, np.sin(2 * np.pi * t / per))
            out[j] = math.sqrt(c * c + s * s) / max(1, x.size)
        out[3] = 0.01
        return out.astype(np.float32)


class StubSeparator:
    """SepformerSeparation look-alike: separate_batch([1,T]) -> [1,T,2]."""

    def __init__(self):
        self.calls: List[int] = []

    def separate_batch(self, mix: torch.Tensor) -> torch.Tensor:
        self.calls.append(int(mix.shape[-1]))
        x = mix.to(torch.float32)
        pad = torch.nn.functional.pad(x, (3, 0))
        low = (pad[..., 3:] + pad[..., 2:-1] + pad[..., 1:-2] + pad[..., :-3]) / 4.0   # kills the period-4 voice
        return torch.stack([x - low, low], dim=-1)


class StubWhisper:
    """whisper model look-alike: records every call, returns a deterministic text."""

    def __init__(self):
        self.calls: List[dict] = []

    def transcribe(self, audio, initial_prompt=None, word_timestamps=False, condition_on_previous_text=True,
                   temperature=0.0, **kw):
        n = int(np.asarray(audio).reshape(-1).shape[0])
        k = len(self.calls)
        self.calls.append(dict(n_samples=n, initial_prompt=initial_prompt, word_timestamps=bool(word_timestamps),
                               condition_on_previous_text=bool(condition_on_previous_text), temperature=float(temperature)))
        return {"text": f" utt{k} n{n}."}


def result_to_json(res, whisper_calls, separator_calls, diar_calls):
    if res is None:
        return dict(result=None, whisper_calls=whisper_calls, separator_calls=separator_calls, diarization_calls=diar_calls)
    segs = []
    for s in res["segments"]:
        md = dict(s.metadata)
        if "overlap_speakers" in md:
            md["overlap_speakers"] = sorted(md["overlap_speakers"])   # came from a set in the reference (api.py:343)
        at = s.audio_tensor
        segs.append(dict(start=float(s.start), end=float(s.end), speaker_id=s.speaker_id, is_overlap=bool(s.is_overlap),
                         transcription=s.transcription, confidence=float(s.confidence), metadata=md,
                         audio_len=int(at.shape[-1]), audio_abs_sum=float(at.abs().sum())))
    md = dict(res["metadata"])
    return dict(result=dict(segments=segs, metadata=md), whisper_calls=whisper_calls, separator_calls=separator_calls,
                diarization_calls=diar_calls)


SCENARIOS: List[Scenario] = [
    Scenario("two_speakers_one_overlap_30s", 30.0,
             truth=[("A", 0, 9), ("B", 7, 16), ("A", 18, 24), ("B", 26, 30)],
             diarization=[(0.0, 9.0, "SPEAKER_00"), (7.0, 16.0, "SPEAKER_01"), (18.0, 24.0, "SPEAKER_00"), (26.0, 30.0, "SPEAKER_01")],
             vad=[(0.0, 16.0), (18.0, 24.0), (26.0, 30.0)]),
    Scenario("two_speakers_overlap_10s", 10.0,
             truth=[("A", 0, 6), ("B", 4, 10)],
             diarization=[(0.0, 6.0, "SPEAKER_00"), (4.0, 10.0, "SPEAKER_01")],
             vad=[(0.0, 10.0)]),
    Scenario("single_speaker", 12.0,
             truth=[("A", 0.5, 5.0), ("A", 5.6, 11.0)],
             diarization=[(0.5, 5.0, "SPEAKER_00"), (5.6, 11.0, "SPEAKER_00")],
             vad=[(0.4, 5.1), (5.5, 11.2)]),
    Scenario("overlap_below_threshold", 14.0,
             truth=[("A", 0, 5), ("B", 4.7, 9), ("A", 9.3, 13.5)],
             diarization=[(0.0, 5.0, "SPEAKER_00"), (4.7, 9.0, "SPEAKER_01"), (9.3, 13.5, "SPEAKER_00")],
             vad=[(0.0, 13.5)]),
    Scenario("short_overlap_segment", 8.0,
             truth=[("A", 0, 4.2), ("B", 3.0, 4.6), ("A", 5.0, 8.0)],
             diarization=[(0.0, 4.2, "SPEAKER_00"), (3.0, 4.6, "SPEAKER_01"), (5.0, 8.0, "SPEAKER_00")],
             vad=[(0.0, 4.6), (5.0, 8.0)]),
    Scenario("low_similarity_secondary_diarization", 16.0,
             truth=[("A", 0, 4), ("B", 4.8, 10), ("A", 11, 15.5)],
             # the diarizer wrongly gives the middle segment to speaker 00 -> similarity < 0.30 -> second pass
             diarization=[(0.0, 4.0, "SPEAKER_00"), (4.8, 10.0, "SPEAKER_00"), (11.0, 15.5, "SPEAKER_00"), (15.5, 15.9, "SPEAKER_01")],
             vad=[(0.0, 4.0), (4.8, 10.0), (11.0, 15.9)],
             secondary=[(0.0, 2.4, "SPEAKER_01"), (2.5, 5.2, "SPEAKER_01")]),
    Scenario("no_speakers", 5.0, truth=[], diarization=[], vad=[]),
    Scenario("vad_trims_and_drops", 20.0,
             truth=[("A", 1, 6), ("B", 8, 8.6), ("B", 12, 19)],
             diarization=[(1.0, 6.0, "SPEAKER_00"), (8.0, 8.6, "SPEAKER_01"), (12.0, 19.0, "SPEAKER_01")],
             vad=[(1.5, 5.0), (8.3, 8.5), (12.2, 15.0), (15.4, 18.0)]),
    Scenario("no_vad_refinement", 20.0,
             truth=[("A", 1, 6), ("B", 8, 8.6), ("B", 12, 19)],
             diarization=[(1.0, 6.0, "SPEAKER_00"), (8.0, 8.6, "SPEAKER_01"), (12.0, 19.0, "SPEAKER_01")],
             vad=[(1.5, 5.0)], config=dict(use_vad_refinement=False)),
    Scenario("rapid_exchange_and_prompt_carry", 24.0,
             truth=[("A", 0, 3), ("A", 3.6, 6), ("B", 6.3, 9), ("A", 9.2, 12), ("A", 14, 17), ("B", 20, 23.5)],
             diarization=[(0.0, 3.0, "SPEAKER_00"), (3.6, 6.0, "SPEAKER_00"), (6.3, 9.0, "SPEAKER_01"), (9.2, 12.0, "SPEAKER_00"),
                          (14.0, 17.0, "SPEAKER_00"), (20.0, 23.5, "SPEAKER_01")],
             vad=[(0.0, 12.0), (14.0, 17.0), (20.0, 23.5)], config=dict(merge_gap_threshold=0.3)),
    Scenario("no_speaker_embeddings_overlap_fails", 10.0,
             truth=[("A", 0, 6), ("B", 4, 10)],
             diarization=[(0.0, 6.0, "SPEAKER_00"), (4.0, 10.0, "SPEAKER_01")],
             vad=[(0.0, 10.0)], config=dict(use_speaker_embeddings=False)),
    Scenario("three_labels_third_is_unknown", 26.0,
             truth=[("A", 0, 5), ("B", 6, 11), ("C", 12, 15), ("A", 16, 20), ("B", 21, 25.5)],
             diarization=[(0.0, 5.0, "SPEAKER_00"), (6.0, 11.0, "SPEAKER_01"), (12.0, 15.0, "SPEAKER_02"), (16.0, 20.0, "SPEAKER_00"),
                          (21.0, 25.5, "SPEAKER_01")],
             vad=[(0.0, 25.5)]),
    Scenario("custom_thresholds", 30.0,
             truth=[("A", 0, 9), ("B", 7, 16), ("A", 18, 24), ("B", 26, 30)],
             diarization=[(0.0, 9.0, "SPEAKER_00"), (7.0, 16.0, "SPEAKER_01"), (18.0, 24.0, "SPEAKER_00"), (26.0, 30.0, "SPEAKER_01")],
             vad=[(0.0, 16.0), (18.0, 24.0), (26.0, 30.0)],
             config=dict(sliding_window_size=1.0, sliding_window_step=0.5, overlap_threshold=1.0, max_embedding_segments=2,
                         condition_on_previous_text=False, noise_reduction_amount=0.3)),
]


def interval_cases():
    rng = np.random.default_rng(20240607)
    cases = []
    for i in range(24):
        n = int(rng.integers(0, 9))
        segs = []
        t = 0.0
        for _ in range(n):
            s = t + float(rng.uniform(-1.0, 1.5)) if segs else float(rng.uniform(0, 2))
            s = max(0.0, round(s, 3))
            e = round(s + float(rng.uniform(0.2, 4.0)), 3)
            segs.append([s, e, f"SPEAKER_0{int(rng.integers(0, 3))}"])
            t = e
        rng.shuffle(segs)   # merge_diarization_segments sorts in place
        vad = []
        v = 0.0
        for _ in range(int(rng.integers(0, 5))):
            s = round(v + float(rng.uniform(0, 2)), 3)
            e = round(s + float(rng.uniform(0.1, 5)), 3)
            vad.append([s, e])
            v = e
        cases.append(dict(segments=[list(x) for x in segs], gap=[0.5, 0.0, 1.0][i % 3], vad=vad))
    # hand-written edge cases: touching intervals, identical starts, fully nested, same-time end/start
    cases.append(dict(segments=[[0.0, 1.0, "A"], [1.0, 2.0, "A"], [2.5, 3.0, "A"], [2.5, 2.7, "B"]], gap=0.5, vad=[[0.0, 1.0], [1.0, 3.0]]))
    cases.append(dict(segments=[[0.0, 10.0, "A"], [2.0, 3.0, "B"], [4.0, 5.0, "B"], [4.5, 6.0, "C"]], gap=0.5, vad=[[2.5, 2.6]]))
    cases.append(dict(segments=[[1.0, 2.0, "A"], [2.0, 3.0, "B"]], gap=0.5, vad=[[0.0, 0.5]]))
    return cases
