"""Times ONE 30 s clip end to end through the CPU oracle pipeline at FULL size (the measured counterpart of bench.py's extrapolated
`cpu_baseline`): tests/pinned_oracle.run_clip (gate, profiles, segment / window embeddings, full-depth separator, source pick)
+ run_pipelines (VAD, diarization with the ResNet-34) + the six Whisper small.en calls of the pinned schedule (log-mel, encoder,
224 greedy tokens each).  fp32 torch / scipy on the host cores.  TEST / MEASUREMENT INFRASTRUCTURE (imports oracle/).
    python tools/cpu_full_clip.py [threads] [sample_len] > profiles/rNN_cpu_full_clip.json"""
import json
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else min(os.cpu_count() or 1, 32)
    sample_len = int(sys.argv[2]) if len(sys.argv) > 2 else 224
    torch.set_num_threads(threads)
    from clearconverse_amd.audio import synthetic_clip
    from clearconverse_amd.models import build_state_dicts
    from clearconverse_amd.tokenizer import DecodeRules
    from clearconverse_amd.weights import SepDims
    from oracle import whisper_ref as R
    from tests import pinned_oracle as O
    sds = build_state_dicts(None, seed=0)
    clip = synthetic_clip(0, 30.0)
    stages = {}
    t_all = time.perf_counter()
    t0 = time.perf_counter()
    with torch.no_grad():
        front = O.run_clip(clip, sds, SepDims(**sds["sep_dims"]))
    stages["gate_profiles_embeddings_separation_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    pipes = O.run_pipelines(clip, sds, min_speakers=1, max_speakers=2)
    stages["vad_diarization_s"] = time.perf_counter() - t0
    rules = DecodeRules()
    orules = R.Rules(suppress=tuple(rules.suppress))
    orc = R.WhisperRef(R.Dims(**sds["whisper_dims"]), sds["whisper"])
    waves = list(front["regular"]) + [front["sources"][k][int(front["source_sims"][k][1] > front["source_sims"][k][0])] for k in range(4)]
    t0 = time.perf_counter()
    n_tok = 0
    with torch.no_grad():
        for w in waves:
            mel = R.pad_or_trim(R.log_mel_spectrogram(torch.from_numpy(np.ascontiguousarray(w)))[:, : len(w) // 160], 3000)
            xa = orc.encode(mel[None])
            r = R.greedy_decode_cached(orc, xa, [rules.sot], orules, sample_len=sample_len)
            n_tok += len(r.tokens)
    stages["whisper_6_calls_s"] = time.perf_counter() - t0
    total = time.perf_counter() - t_all
    print(json.dumps({"what": "one 30 s clip through the CPU oracle pipeline at full size (small.en, full-depth RE-SepFormer, x-vector, PyanNet, ResNet-34)",
                      "threads": threads, "cpu_count": os.cpu_count(), "sample_len": sample_len, "whisper_calls": len(waves), "tokens_decoded": n_tok,
                      "stages": {k: round(v, 2) for k, v in stages.items()}, "total_s": round(total, 2), "xrt": round(30.0 / total, 4),
                      "vad_regions": len(pipes["vad"]), "diarization_turns": len(pipes["diarization"])}))


if __name__ == "__main__":
    main()
