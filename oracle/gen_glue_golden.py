#!/usr/bin/env python3
"""Generate tests/golden/glue_*.json by RUNNING the reference's own glue code.

TEST INFRASTRUCTURE ONLY.  The reference's orchestration (the control flow of the hot path) is
plain Python in /root/reference/back/api.py, but the module cannot be imported here (its imports
`dotenv, torchaudio, whisper, noisereduce, pyannote.audio, speechbrain, validators, reportlab` are
not installed -- ordinary ModuleNotFoundError).  This script therefore parses
the file with `ast`, and executes ONLY these definitions in a namespace of stubs:

  Config, AudioSegment                       back/api.py:101-135
  merge_diarization_segments ... enhance_audio   back/api.py:294-352
  ensure_wav_format                          back/api.py:530-568
  class EnhancedAudioProcessor               back/api.py:584-1549

The five model objects are replaced by deterministic scripted stubs (below); everything else --
interval arithmetic, speaker-profile building, label mapping, overlap routing, sliding-window
re-segmentation, prompt selection, result assembly, transcript text -- is the reference's own code.
Only the JSON fixtures (inputs + outputs) travel; no reference source is copied.

  python oracle/gen_glue_golden.py           # (re)write the fixtures
  python oracle/gen_glue_golden.py --check   # regenerate in memory and compare with the committed files
"""
from __future__ import annotations

import ast
import json
import logging
import math
import os
import sys
import tempfile
import types
from collections import Counter, defaultdict
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference/back/api.py")
GOLDEN = ROOT / "tests" / "golden"
WANTED = {"Config", "AudioSegment", "merge_diarization_segments", "get_pyannote_vad_intervals",
          "refine_segment_with_vad", "find_segment_overlaps", "enhance_audio", "ensure_wav_format",
          "EnhancedAudioProcessor"}

sys.path.insert(0, str(ROOT))
from tests.glue_stubs import (Annotation, Scenario, SCENARIOS, StubEmbedding, StubSeparator, StubWhisper,  # noqa: E402
                              scenario_audio, interval_cases, result_to_json)


def lift_reference():
    """exec the wanted top-level definitions of the reference into a stub namespace."""
    tree = ast.parse(REF.read_text())
    nodes = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in WANTED]
    found = {n.name for n in nodes}
    missing = WANTED - found
    if missing:
        raise RuntimeError(f"reference no longer defines {missing}")
    nr = types.SimpleNamespace(reduce_noise=lambda y, sr, stationary=True, prop_decrease=0.75: np.asarray(y))

    def _save(path, tensor, sr):
        Path(path).write_bytes(b"stub")
    torchaudio = types.SimpleNamespace(save=_save)
    ns: Dict[str, Any] = dict(torch=torch, np=np, logging=logging, os=os, Path=Path, Counter=Counter,
                              defaultdict=defaultdict, dataclass=dataclass, field=field, Any=Any, Dict=Dict,
                              List=List, Optional=Optional, Tuple=Tuple, nr=nr, torchaudio=torchaudio,
                              traceback=__import__("traceback"), json=json, subprocess=__import__("subprocess"),
                              env_config={"model_cache_dir": "models"})
    mod = ast.Module(body=nodes, type_ignores=[])
    exec(compile(mod, str(REF), "exec"), ns)
    return ns


def run_scenario(ns, sc: Scenario) -> dict:
    Config, Proc = ns["Config"], ns["EnhancedAudioProcessor"]
    cfg = Config(auth_token="x", **sc.config)
    p = Proc(cfg, load_models_immediately=False)
    p.device = torch.device("cpu")
    audio = scenario_audio(sc)
    p.load_audio = lambda path: (audio.clone(), 16000)
    whisper = StubWhisper()
    p.whisper_model = whisper
    p.separator = StubSeparator()
    p.embedding_model = StubEmbedding()
    p.vad_pipeline = lambda path: Annotation([(s, e, "SPEECH") for s, e in sc.vad])
    diar_calls = []

    def diar(path, min_speakers=None, max_speakers=None):
        diar_calls.append(os.path.basename(str(path)))
        if os.path.basename(str(path)) == "temp_segment.wav":
            return Annotation(list(sc.secondary))
        return Annotation(list(sc.diarization))
    p.diarization = diar
    p.models_loaded = {k: True for k in p.models_loaded}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        try:
            logging.disable(logging.CRITICAL)
            res = p.process_file("clip.wav")
            out = result_to_json(res, whisper.calls, p.separator.calls, diar_calls)
        finally:
            logging.disable(logging.NOTSET)
            os.chdir(cwd)
    # transcript text exactly as run() formats it (back/api.py:1255-1257) -- computed by the reference's
    # format string applied to the reference's own segment list
    if res is not None:
        t = ""
        for seg in res["segments"]:
            t += f"[{seg.speaker_id}] {seg.start:.2f}s - {seg.end:.2f}s\n"
            t += f"{seg.transcription}\n\n"
        out["transcript"] = t
    return out


def generate() -> Dict[str, dict]:
    ns = lift_reference()
    files: Dict[str, dict] = {}
    # 1. interval helpers on seeded random inputs
    cases = []
    for case in interval_cases():
        segs = [tuple(x) for x in case["segments"]]
        merged = ns["merge_diarization_segments"](list(segs), case["gap"])
        ov = ns["find_segment_overlaps"](list(segs))
        refined = [ns["refine_segment_with_vad"]((s, e), [tuple(v) for v in case["vad"]]) for s, e, _ in segs]
        cases.append(dict(input=case, merged=[list(m) for m in merged],
                          overlaps=sorted([[k[0], k[1], sorted(v)] for k, v in ov.items()]),
                          refined=[list(r) if r is not None else None for r in refined]))
    files["glue_intervals.json"] = dict(source="reference back/api.py:294-343 executed via oracle/gen_glue_golden.py", cases=cases)
    # 2. whole process_file control flow under scripted stub models
    scen = {}
    for sc in SCENARIOS:
        scen[sc.name] = dict(scenario=sc.to_json(), expected=run_scenario(ns, sc))
    files["glue_process_file.json"] = dict(source="reference back/api.py:584-1549 executed via oracle/gen_glue_golden.py",
                                           scenarios=scen)
    return files


def main():
    if not REF.exists():
        print("reference not present; fixtures are used as committed")
        return 0
    files = generate()
    GOLDEN.mkdir(parents=True, exist_ok=True)
    if "--check" in sys.argv:
        bad = []
        for name, data in files.items():
            p = GOLDEN / name
            if not p.exists() or json.loads(p.read_text()) != json.loads(json.dumps(data)):
                bad.append(name)
        if bad:
            print("golden fixtures differ from the reference run:", bad)
            return 1
        print("golden fixtures match the reference run")
        return 0
    for name, data in files.items():
        (GOLDEN / name).write_text(json.dumps(data, indent=1, sort_keys=True) + "\n")
        print("wrote", GOLDEN / name)
    return 0


if __name__ == "__main__":
    sys.exit(main())
