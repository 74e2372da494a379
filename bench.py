#!/usr/bin/env python3
"""bench.py -- xRT (audio-seconds / wall-second) of the MI355X-native ClearConverse hot path.

  python bench.py --gpus N --steps K --warmup W [--workload pipeline|whisper] [--batch B]

One "step" = one pass of the hot path over one batch of synthetic 30 s / 16 kHz clips already resident
in HBM.
  * workload `pipeline` (default; BASELINE.json configs[3], the configuration the metric is quoted on):
    full VAD -> diarize -> separate -> transcribe pass over B=32 clips per GPU through
    clearconverse_amd.batch.BatchPipeline (every model call batched across clips; control flow pinned to
    the synthetic schedule because seeded random weights give arbitrary diarization / EOT -- stated in
    `config.schedule`; VAD and diarization are still computed).
  * workload `whisper` (configs[1]): log-mel -> small.en encoder -> greedy decode of 8 x 30 s clips.
N>1: one process per GPU (launched by torchrun, or by bench.py itself when called plainly with --gpus N), clips sharded across ranks (clip i -> rank i mod N) with no
data-path collective; the only collectives are the timing barrier/max and one all-gather of the token
records at the end (weak scaling: B clips per GPU).

The JSON line carries `roofline` for the dominant kernel (per-launch HIP events recorded by libccx on the
launch stream during one extra step right after the timed region) and `cpu_baseline` (oracle/, fp32 on the host
cores, bounded sample, rank 0 at N=1 only) -- DESIGN.md "Measurement".
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np
import torch

PEAK_MFMA_BF16_TFLOPS = 2500.0  # dense bf16, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0
MFMA_KERNELS = ("gemm_bf16_nt_kernel", "enc_attention_kernel")


def enc_flops_per_window(d):
    """SURVEY.md section 8d formula (encoder incl. conv stem); the second figure is the cross-K/V projection of the 12 decoder layers,
    which round 3's decode no longer computes for groups of more than 80 sequences (the cross attention reads the encoder output
    itself, csrc/cross_x.hip) -- reported, not counted."""
    D, L, S = d.n_audio_state, d.n_audio_layer, d.n_audio_ctx
    enc = 2 * (3000 * D * 3 * d.n_mels + S * D * 3 * D + L * (S * (4 * D * D + 2 * D * 4 * D) + 2 * S * S * D))
    cross = 2 * d.n_text_layer * 2 * S * D * D
    return enc, cross


def cross_bytes_per_sequence(d, xstream=True):
    """Algorithmic HBM bytes of ONE layer's cross attention for ONE sequence and decode step.  The reference's formulation reads the
    layer's K and V (2 x n_audio_ctx x D bf16); the X-stream formulation reads the encoder output once
    (n_audio_ctx x D bf16) plus the expanded queries (heads x D bf16) and writes the two key halves' partial contexts (2 x heads x D f32)
    -- the bytes of dec_xs_stream_kernel, the dominant launch (csrc/cross_x.hip)."""
    D, S, H = d.n_text_state, d.n_audio_ctx, d.n_text_head
    return S * D * 2 + H * D * 2 + 2 * H * D * 4 if xstream else 2 * S * D * 2


def decode_bytes_per_step(d, B, xstream=True):
    """Algorithmic HBM bytes of one decode step: bf16 decoder weights + tied embedding once, plus each sequence's cross attention
    (see cross_bytes_per_sequence)."""
    D, L = d.n_text_state, d.n_text_layer
    w = 2 * (L * (4 * D * D + 2 * D * 4 * D + 4 * D * D) + d.n_vocab * D)   # self (qkv+o) + mlp + cross (q, k, v, o) + logits
    return w + B * L * cross_bytes_per_sequence(d, xstream)


def cpu_baseline_whisper(dims, sd, clip, rules, threads, tokens):
    from oracle import whisper_ref as R
    torch.set_num_threads(threads)
    orc = R.WhisperRef(R.Dims(**dims.__dict__), sd)
    orules = R.Rules(suppress=tuple(rules.suppress))
    t0 = time.perf_counter()
    with torch.no_grad():
        mel = R.pad_or_trim(R.log_mel_spectrogram(torch.from_numpy(clip))[:, : len(clip) // 160], 3000)
        xa = orc.encode(mel[None])
        t_enc = time.perf_counter() - t0
        r = R.greedy_decode_cached(orc, xa, [rules.sot], orules, sample_len=tokens)
    dt = time.perf_counter() - t0
    return dict(t_enc=t_enc, t_tok=(dt - t_enc) / max(1, len(r.margins)), total=dt, n_tok=len(r.tokens))


def cpu_baseline_pipeline(models_sd, clip, rules, threads):
    """Oracle op timings on bounded samples, scaled by the op counts of ONE 30 s clip under the pinned
    schedule (6 Whisper calls x (encode + 224 tokens), 4 separator regions (18 s), 107.6 s of x-vector input,
    63 x 10 s ResNet-34 diarization crops (the reference embeds every (chunk, local speaker) pair separately),
    51 x 5 s VAD + 21 x 10 s segmentation chunks, 58 s through the gate)."""
    from oracle import pyannote_ref as P, sepformer_ref as S, spectral_gate_ref as G, wespeaker_ref as W, whisper_ref as R
    torch.set_num_threads(threads)
    wd, wsd, sdims, ssd, xsd, psd, rsd = models_sd
    w = cpu_baseline_whisper(wd, wsd, clip[: 16000 * 6], rules, threads, tokens=24)
    with torch.no_grad():
        t0 = time.perf_counter()
        S.SepformerRef(S.SepDims(**sdims.__dict__), ssd).separate(torch.from_numpy(clip[None, : 16000 * 2]))
        t_sep_per_s = (time.perf_counter() - t0) / 2.0
        t0 = time.perf_counter()
        for _ in range(4):
            P.xvector_forward(xsd, torch.from_numpy(clip[None, : 12800]))
        t_xvec_per_s = (time.perf_counter() - t0) / (4 * 0.8)
        t0 = time.perf_counter()
        W.resnet_embed(rsd, clip[None, : 16000 * 2])
        t_res_per_s = (time.perf_counter() - t0) / 2.0
        osd = dict(psd); osd["powerset"] = torch.tensor(1)
        t0 = time.perf_counter()
        P.pyannet_forward(osd, torch.from_numpy(clip[None, None, : 16000 * 5]))
        t_seg_per_s = (time.perf_counter() - t0) / 5.0
    t0 = time.perf_counter()
    G.reduce_noise(clip, 16000, 0.5)
    t_gate_per_s = (time.perf_counter() - t0) / 30.0
    t_clip = (6 * (w["t_enc"] + 224 * w["t_tok"]) + 18.0 * t_sep_per_s + 107.6 * t_xvec_per_s + 630.0 * t_res_per_s
              + (51 * 5 + 21 * 10) * t_seg_per_s + 58.0 * t_gate_per_s)
    sample = (f"timed samples (fp32 torch/scipy oracle): 1 Whisper window encode {w['t_enc']:.1f} s + 24 greedy tokens "
              f"({w['t_tok'] * 1e3:.0f} ms/token), 2 s separator, 4 x 0.8 s x-vector, 2 s ResNet-34, 5 s PyanNet, 30 s spectral gate; "
              f"scaled by one clip's op counts under the pinned schedule -> {t_clip:.0f} s per 30 s clip")
    return 30.0 / t_clip, sample


class PowerProbe:
    """Socket power and shader clock of rank 0's card over the timed region: hwmon power1_input / freq1_input read every 50 ms by a
    thread (no HIP, no subprocess once the GPU is initialised: the card is looked up BEFORE, with `rocm-smi --showbus` -- the host's other
    cards belong to other boxes).  Reported as `power` in the JSON line; null when the files are not there.  tools/power_sampler.py is
    the stand-alone form."""

    def __init__(self):
        import glob, re, subprocess
        self.h = None
        try:
            out = subprocess.run(["rocm-smi", "--showbus"], capture_output=True, text=True, timeout=30).stdout
            m = re.search(r"GPU\[0\].*?([0-9a-fA-F]{4}:[0-9a-fA-F]{2}:[0-9a-fA-F]{2}\.[0-9a-fA-F])", out)
            bdf = m.group(1).lower() if m else None
            for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
                if bdf and os.path.basename(os.path.realpath(os.path.join(h, "device"))).lower() == bdf:
                    self.h = h
        except Exception:
            self.h = None
        self.samples, self._stop, self._thr = [], False, None

    @staticmethod
    def _read(path):
        with open(path) as f:
            return int(f.read().strip())

    def start(self):
        if self.h is None:
            return
        import threading

        def loop():
            while not self._stop:
                try:
                    self.samples.append((self._read(os.path.join(self.h, "power1_input")) * 1e-6, self._read(os.path.join(self.h, "freq1_input")) * 1e-6))
                except (OSError, ValueError):
                    pass
                time.sleep(0.05)
        self._thr = threading.Thread(target=loop, daemon=True)
        self._thr.start()

    def stop(self):
        if self._thr is None:
            return None
        self._stop = True
        self._thr.join(timeout=2.0)
        if not self.samples:
            return None
        pw = sorted(x[0] for x in self.samples)
        ck = sorted(x[1] for x in self.samples)
        try:
            cap = self._read(os.path.join(self.h, "power1_cap")) * 1e-6
        except (OSError, ValueError):
            cap = None
        return {"mean_w": round(sum(pw) / len(pw), 1), "median_w": round(pw[len(pw) // 2], 1), "p95_w": round(pw[int(len(pw) * 0.95)], 1),
                "cap_w": cap, "sclk_mhz_mean": round(sum(ck) / len(ck)), "sclk_mhz_min": round(ck[0]), "samples": len(pw),
                "source": "hwmon power1_input / freq1_input of rank 0's card, every 50 ms over the timed region"}


def spawn_ranks(n: int) -> int:
    """One process per GPU on this node: `python -m torch.distributed.run --nnodes=1 --nproc-per-node n bench.py <same args>`
    on 127.0.0.1 with a free port (the container hostname may not resolve).  Returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL between processes needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def lane_cross_attention_in_situ(path, bytes_per_sequence):
    """Reduce the decode-lane trace of libccx (ccx_whisper_trace_lanes; the format tools/decode_stamps.py reads) of the LAST decode
    in `path`: per lane the median time between the stamp in front of a layer's cross attention and the one behind it, while the
    other lanes run their own steps (graph replay, real concurrency -- what neither HIP events nor the tracer can see)."""
    blocks, cur = [], None
    for line in open(path):
        q = line.split()
        if q and q[0] == "decode":
            cur = {"B": int(q[2]), "lanes": {}}
            blocks.append(cur)
        elif q and q[0] == "lane" and cur is not None:
            v = np.array([int(x) for x in q[3:]], dtype=np.uint64)
            cur["lanes"][int(q[1])] = ((v >> np.uint64(8)).astype(np.int64) * 10, (v & np.uint64(255)).astype(np.int64))   # ns, tag
    if not blocks or not blocks[-1]["lanes"]:
        return None
    blk = blocks[-1]
    nl, B = len(blk["lanes"]), blk["B"]
    per = -(-(-(-B // nl)) // 16) * 16          # lane sizes as ccx_whisper_decode cuts them: multiples of one MFMA row tile
    out = []
    for lane, (t, tag) in sorted(blk["lanes"].items()):
        d = (t[1:] - t[:-1])[tag[1:] == 2] / 1e3                 # us, intervals that END at a tag-2 stamp = one cross attention
        d = d[len(d) // 4:]                                        # skip the eager first step and the capture pass
        rows = min(per, B - lane * per)
        ends = t[tag == 3]                                         # one stamp at the end of every decode step of the lane
        st = np.diff(ends)[len(ends) // 4:] / 1e3
        if len(d) and rows > 0:
            out.append(dict(lane=lane, sequences=rows, launches=int(len(d)), median_us=float(np.median(d)), p90_us=float(np.percentile(d, 90)),
                            gbs=rows * bytes_per_sequence / (float(np.median(d)) * 1e-6) / 1e9,
                            step_median_us=float(np.median(st)) if len(st) else None))
    return out or None


def single_file(args, rank, local_rank, world):
    """The drop-in path as the reference calls it (back/api.py:1204-1280, one task = one file = one process): a step = one
    `EnhancedAudioProcessor.run` on one 30 s / 16 kHz WAV -- file read, gate, VAD, diarization, profiles, per-segment embeddings,
    overlap re-segmentation, separation and one B = 1 Whisper decode per segment (prompt-chained), segment WAVs and transcript
    written.  Weights are seeded (no checkpoint offline), so VAD / diarization are computed and then replaced by the pinned
    synthetic schedule, exactly as the pipeline workload does.  Latency mode: no roofline entry."""
    import tempfile
    from clearconverse_amd import _lib
    from clearconverse_amd.audio import SCHEDULE_30S, synthetic_clip, write_wav
    from clearconverse_amd.models import load_models
    from clearconverse_amd.pipelines import Annotation
    from clearconverse_amd.processor import Config, EnhancedAudioProcessor
    ctx = _lib.Context(local_rank)
    models = dict(load_models(None, local_rank, whisper_batch=8, ctx=ctx, seed=0, max_audio_seconds=30.0))

    class Pinned:
        """Runs the real pipeline (its cost stays in the step), returns the scheduled Annotation."""
        def __init__(self, inner, tracks):
            self.inner, self.tracks = inner, tracks

        def __call__(self, path, **kw):
            self.inner(path, **kw)
            return Annotation(self.tracks)
    speech = [(0.0, 16.0, "SPEECH"), (18.0, 24.0, "SPEECH"), (26.0, 30.0, "SPEECH")]
    turns = [(s, e, "SPEAKER_00" if spk == "A" else "SPEAKER_01") for spk, s, e in SCHEDULE_30S]
    models["vad_pipeline"] = Pinned(models["vad_pipeline"], speech)
    models["diarization"] = Pinned(models["diarization"], turns)
    work = tempfile.mkdtemp(prefix="ccx_single_file_")
    n_files = max(1, min(8, args.steps))
    wavs = []
    for i in range(n_files):
        wavs.append(os.path.join(work, f"clip{i}.wav"))
        write_wav(wavs[-1], synthetic_clip(rank + world * i, 30.0))
    proc = EnhancedAudioProcessor(Config(temperature=0.0), load_models_immediately=False, model_loader=lambda cfg, dev: models)
    proc._initialize_models()
    outs = []

    def step(i):
        out = proc.run(wavs[i % n_files], output_dir=os.path.join(work, f"out{i % n_files}"))
        outs.append(out)
        return out
    for i in range(max(1, args.warmup)):
        step(i)
    torch.cuda.synchronize()
    del outs[:]
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        segs = [o[1].count("\n\n") if o[1] else 0 for o in outs]
        print(json.dumps({
            "metric": "xRT (audio-sec/wall-sec) end-to-end, 30 s 16 kHz clips", "value": round(30.0 * args.steps / dt, 2), "unit": "xRT",
            "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3 / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic clips (seed 1234+i) as 16-bit WAV files, weights synthetic-seed0, greedy T=0",
            "config": {"workload": "single_file: EnhancedAudioProcessor.run, one 30 s WAV per step (the reference's calling pattern, back/api.py:1204-1280)",
                       "clips_per_gpu": 1, "clip_seconds": 30, "transcript_segments_per_file": segs[:8],
                       "schedule": "VAD and diarization computed, then replaced by the pinned synthetic schedule (seeded weights); every "
                                   "Whisper call is a B = 1 decode chained by its prompt; WAV read and segment / transcript files written inside the step"},
            "roofline": None, "cpu_baseline": None}), flush=True)


def main():
    # every CCX_* switch of libccx found in the environment goes into the JSON line; the diagnostic ones that make results garbage
    # are refused (tools/README.md, "Measurement switches")
    ccx_env = {k: v for k, v in sorted(os.environ.items()) if k.startswith("CCX_")}
    if "CCX_ABLATE" in ccx_env:
        raise SystemExit("bench.py: CCX_ABLATE drops kernels from the decode step (diagnostic, results are garbage) -- unset it")
    os.environ.setdefault("CCX_PROF_SHAPES", "1")      # per-shape GEMM / LayerNorm labels in the profiled step (labels only; folded back below)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: two 4-batch decode groups timed behind one warm-up group -- a single group has no earlier decode for its front
    # end to overlap with and reads ~5 % slower than the steady state the driver's --steps 20 --warmup 5 shows
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", choices=("pipeline", "whisper", "single_file"), default="pipeline",
                    help="pipeline = BASELINE configs[3] (default); whisper = configs[1]; single_file = the reference's own calling pattern: "
                         "EnhancedAudioProcessor.run on one 30 s WAV at a time (B = 1 decodes chained by the prompt)")
    ap.add_argument("--batch", type=int, default=None, help="30 s clips per GPU per step (default 32 pipeline / 8 whisper)")
    ap.add_argument("--sample-len", type=int, default=224)
    ap.add_argument("--whisper-group", type=int, default=192, help="sequences decoded together in the pipeline workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stage-times", action="store_true", help="one extra (untimed) step with a sync after every stage")
    ap.add_argument("--no-comparisons", action="store_true",
                    help="skip the span-1 and sequential comparison legs after the timed region (profiling runs: fewer launch shapes per symbol)")
    ap.add_argument("--decode-span", type=int, default=4,
                    help="pipelined schedule: batches whose Whisper windows are encoded and decoded together (1 = one decode group per batch)")
    ap.add_argument("--schedule", choices=("pipelined", "sequential"), default="pipelined",
                    help="pipeline workload: overlap batch i's Whisper decode with batch i+1's front end (default) or run each batch start to finish")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher.  Nothing has touched the GPU yet (importing torch does not
        # initialise HIP), so the N ranks are ordinary child processes; rank 0's JSON line goes straight to our stdout and a
        # failing rank makes torch.distributed.run -- and therefore this process -- exit non-zero.
        raise SystemExit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} or without a launcher")
    probe = PowerProbe() if (rank == 0 and local_rank == 0) else None      # looks its card up before anything touches the GPU
    if os.environ.get("CCX_BENCH_SHARE_GPU"):
        local_rank = 0      # rehearsal of the N > 1 code path on a one-GPU box (with CCX_BENCH_BACKEND=gloo): all ranks on cuda:0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("CCX_BENCH_BACKEND", "nccl")      # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from clearconverse_amd import _lib
    from clearconverse_amd.audio import synthetic_clip
    from clearconverse_amd.batch import BatchPipeline, gather_transcripts
    from clearconverse_amd.tokenizer import DecodeRules
    from clearconverse_amd.weights import (SepDims, WhisperDims, find_whisper_checkpoint, synthetic_pyannet_state_dict,
                                           synthetic_resnet34_state_dict, synthetic_sepformer_state_dict, synthetic_whisper_state_dict,
                                           synthetic_xvector_state_dict)

    if args.workload == "single_file":
        return single_file(args, rank, local_rank, world)
    pipeline = args.workload == "pipeline"
    B = args.batch or (32 if pipeline else 8)
    rules = DecodeRules()
    ctx = _lib.Context(local_rank)
    ck = find_whisper_checkpoint("small.en")
    weights = "checkpoint" if ck is not None else "synthetic-seed0"
    dims = ck[0] if ck is not None else WhisperDims.small_en()

    # clip i of the global corpus -> rank i % world; B clips per rank
    clips = [synthetic_clip(rank + world * i, 30.0) for i in range(B)]
    audio = torch.from_numpy(np.stack(clips)).cuda(local_rank).contiguous()   # resident in HBM before timing

    t_load0 = time.perf_counter()
    bcast_ms = None
    if pipeline:
        from clearconverse_amd.batch import broadcast_weights
        from clearconverse_amd.models import build_state_dicts, load_models
        # C1: rank 0 reads / generates the weights, ONE RCCL broadcast of the packed blob, every rank
        # uploads its replica.  Outside the timed region; reported as weight_broadcast_ms / model_load_ms.
        sds = build_state_dicts(None, seed=0) if rank == 0 else None
        if dist is not None:
            torch.cuda.synchronize(); dist.barrier()
            tb = time.perf_counter()
            sds = broadcast_weights(sds, src=0, device=torch.device("cuda", local_rank))
            torch.cuda.synchronize()
            bcast_ms = (time.perf_counter() - tb) * 1e3
        # every VAD (51 x 5 s) / diarization (21 x 10 s) window of the rank's clips goes through the segmentation net in ONE launch group
        if args.schedule == "pipelined" and args.decode_span > 1:
            args.whisper_group = max(args.whisper_group, 6 * B * args.decode_span)
        models = load_models(None, local_rank, whisper_batch=args.whisper_group, ctx=ctx, seed=0, state_dicts=sds,
                             seg_max_crops=52 * B + 16, seg_max_seconds=300.0 * B, emb_max_crops=44 * B,
                             resnet_max_chunks=21 * B, whisper_instances=2 if args.schedule == "pipelined" else 1, max_audio_seconds=30.0,
                             gate_max_clips=B)
        del sds
        sd = None
        bp = BatchPipeline(models, whisper_group=args.whisper_group, sample_len=args.sample_len)

        def step():
            return bp.run_pinned(audio)
    else:
        from clearconverse_amd.whisper import WhisperModel
        sd = ck[1] if ck is not None else synthetic_whisper_state_dict(dims, seed=0)
        model = WhisperModel(dims, sd, max_batch=B, device=local_rank, ctx=ctx)
        n = [len(c) for c in clips]
        prompts = [[rules.sot] for _ in range(B)]

        def step():
            model.log_mel(audio, n)
            model.encode(B)
            recs = model.decode_greedy(prompts, sample_len=args.sample_len)
            return dict(records=recs, tokens=sum(len(r["tokens"]) for r in recs), whisper_calls=B)

    pipelined = pipeline and args.schedule == "pipelined"

    def run_steps(k):
        """k steps of the hot path; pipelined schedule: the k batches go through the software pipeline together (every batch
        is processed completely inside the call, the decode of batch i overlapping the front end of batch i + 1)."""
        if pipelined:
            return bp.run_pinned_pipelined([audio] * k, span=args.decode_span) if k > 0 else []
        return [step() for _ in range(k)]

    torch.cuda.synchronize()
    load_ms = (time.perf_counter() - t_load0) * 1e3
    if pipelined:
        step()                        # one sequential pass first: one-time kernel set-up and graph capture happen unoverlapped
        # every decode-group shape of the timed region is captured before it: K timed steps form full groups of `span` batches
        # plus one group of K % span, and a warm-up shorter than a group would leave the full group's graphs to the timed region
        shapes = {min(args.decode_span, args.steps)} | ({args.steps % args.decode_span} - {0})
        covered = {min(args.decode_span, args.warmup)} | ({args.warmup % args.decode_span} - {0}) if args.warmup > 0 else set()
        for n_ in sorted(shapes - covered):
            run_steps(n_)
    run_steps(args.warmup)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    if probe is not None:
        probe.start()
    t0 = time.perf_counter()
    n_tokens = n_calls = 0
    res = None
    for res in run_steps(args.steps):
        n_tokens += res["tokens"]
        n_calls += res["whisper_calls"]
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    power = probe.stop() if probe is not None else None
    seq_ms = span1_ms = None
    if pipelined and args.decode_span > 1 and not args.no_comparisons:
        # the same pipeline with one decode group per batch, for comparison (untimed for `value`)
        n1 = min(4, max(2, args.steps))
        torch.cuda.synchronize()
        ts = time.perf_counter()
        bp.run_pinned_pipelined([audio] * n1, span=1)
        torch.cuda.synchronize()
        span1_ms = (time.perf_counter() - ts) * 1e3 / n1
    if pipelined and not args.no_comparisons:
        # the sequential schedule of the same batch, for comparison (untimed for `value`)
        nseq = min(3, max(1, args.steps))
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for _ in range(nseq):
            step()
        torch.cuda.synchronize()
        seq_ms = (time.perf_counter() - ts) * 1e3 / nseq
    # per-launch HIP events (~11 k pairs per pipeline step) are kept OUT of the timed region: one more, untimed, step of the same
    # batch is recorded for the roofline entries
    prof_steps = 1
    ctx.prof_enable(True)
    for _ in range(prof_steps):
        step()
    torch.cuda.synchronize()
    recs = ctx.prof_records()
    ctx.prof_enable(False)
    if pipeline and args.stage_times:
        bp.stage_ms = {}
        bp.run_pinned(audio, timed=True)
    # which cross attention the decode groups run (CCX_CROSS_X=0 at model creation: round 2's per-layer K/V caches)
    xstream = os.environ.get("CCX_CROSS_X", "1") != "0" and dims.n_text_state in (128, 256, 384, 512, 768) and dims.n_text_head * 64 == dims.n_text_state
    # ---- decode probe: the decode chain runs inside hipGraphs in the timed region, where single launches
    # cannot be bracketed by events.  Re-run a few decode steps of the SAME batch eagerly (CCX_NO_GRAPH) with the
    # per-launch HIP events on, to get the average launch duration of the graph-resident kernels.
    probe_steps = 6
    wm = models["whisper_model"] if pipeline else model
    span = args.decode_span if (pipeline and args.schedule == "pipelined") else 1
    Bd = min(args.whisper_group, 6 * B * span) if pipeline else B
    # Two probes.  (1) a group 16 rows per lane SMALLER than the timed one (768 -> 720 sequences = 3 lanes of 240): the same kernels and
    # lane count but a launch shape of its own, so that rocprofv3's per-grid summary of this command (tools/kernel_trace_by_grid.py,
    # profiles/) shows exactly these launches in a row of their own next to `roofline.probe_distinct_grid.avg_launch_us`.  (2) the timed
    # region's own group size: `roofline.achieved` is measured AT the launch shape the timed region runs (a cross-attention launch takes
    # as long as the CU with the most blocks: 240 rows = 480 blocks leave the chip unevenly loaded, 256 rows = 512 blocks do not).
    def n_lanes(b):                       # as ccx_whisper_decode cuts a group into lanes
        if xstream:
            return 3 if b >= 320 else 1
        return 2 if b >= 640 else (3 if b >= 144 else (2 if b >= 96 else 1))
    nl_real = n_lanes(Bd)
    Bp = Bd - 16 * nl_real
    if n_lanes(Bp) != nl_real or nl_real == 1:
        Bp = Bd
    os.environ["CCX_NO_GRAPH"] = "1"
    probe_x = None
    if Bp != Bd:
        ctx.prof_enable(True)
        wm.decode_greedy([[rules.sot]] * Bp, sample_len=probe_steps)
        torch.cuda.synchronize()
        probe_x = ctx.prof_records()
        ctx.prof_enable(False)
    ctx.prof_enable(True)
    wm.decode_greedy([[rules.sot]] * Bd, sample_len=probe_steps)
    torch.cuda.synchronize()
    probe = ctx.prof_records()
    ctx.prof_enable(False)
    del os.environ["CCX_NO_GRAPH"]
    # ---- the same launch IN SITU: graph replay, all lanes of the group stepping concurrently (in-graph time stamps around every
    # layer's cross attention; the solo figure above times one launch at a time)
    in_situ = None
    if rank == 0:
        import tempfile
        tr = os.path.join(tempfile.gettempdir(), f"ccx_lane_trace_{os.getpid()}.txt")
        if os.path.exists(tr):
            os.remove(tr)
        wm.trace_lanes(tr, 1)
        for _ in range(2):                                   # the first decode captures the stamped step graphs, the second replays them
            wm.decode_greedy([[rules.sot]] * Bd, sample_len=24)
        torch.cuda.synchronize()
        wm.trace_lanes(None)
        if os.path.exists(tr):
            in_situ = lane_cross_attention_in_situ(tr, cross_bytes_per_sequence(dims, xstream))
            os.remove(tr)
    # decode launches per pipeline step: a unit of `span` batches is decoded in ceil(6 B span / Bd) groups
    decode_steps_per_step = (args.sample_len + 1) * (((6 * B * span + Bd - 1) // Bd) / span if pipeline else 1)

    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        gathered = gather_transcripts(res["records"], args.sample_len, rules.eot, torch.device("cuda", local_rank))   # C2
        cnt = torch.tensor([n_tokens, n_calls], device="cuda", dtype=torch.int64)
        dist.all_reduce(cnt)
        n_tokens, n_calls = int(cnt[0]), int(cnt[1])
        assert gathered.shape[0] == len(res["records"]) * world

    audio_s = 30.0 * B * world * args.steps
    value = audio_s / dt

    if rank == 0:
        # ---- roofline of the kernel with the largest time per step (HIP events on the launch stream: eager kernels during one extra
        # step right after the timed region, graph-resident decode kernels during an eager re-run of a few decode steps) ----
        def aggregate_records(rs):
            out_ = {}
            for name, fl, by, ms in rs:
                a = out_.setdefault(name, [0, 0.0, 0.0, 0.0])
                a[0] += 1; a[1] += fl; a[2] += by; a[3] += ms
            return out_

        def roof_entry(name, cnt_, fl, by, ms, where):
            pmc = {}
            pfs = sorted((ROOT / "profiles").glob("r*_pmc_traffic.json"))       # the newest round's counter passes
            if pfs:
                pmc = json.loads(pfs[-1].read_text()).get(name, {})
            if name in MFMA_KERNELS:
                ach = fl / (ms * 1e-3) / 1e12
                return dict(kernel=name, bound="mfma", achieved=round(ach, 2), peak=PEAK_MFMA_BF16_TFLOPS, unit="TFLOP/s",
                            frac=round(ach / PEAK_MFMA_BF16_TFLOPS, 4), traffic=pmc.get("hbm_bytes_per_launch"), launches=cnt_,
                            avg_launch_us=round(ms * 1e3 / cnt_, 2), flops_per_launch=fl / cnt_, measured=where)
            ach = by / (ms * 1e-3) / 1e9
            traffic = pmc.get("hbm_bytes_per_launch")
            if traffic is not None and pmc.get("sequences_per_launch"):
                # measured at `sequences_per_launch` per launch (counter passes serialise the lanes); equal to this run's launch
                # shape when the newest committed pass was taken at it, scaled by the sequence count otherwise
                traffic = traffic / pmc["sequences_per_launch"] * (by / cnt_) / cross_bytes_per_sequence(dims, xstream)
            return dict(kernel=name, bound="hbm", achieved=round(ach, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                        frac=round(ach / PEAK_HBM_GBS, 4), traffic=traffic,
                        traffic_source=(f"{pfs[-1].name}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, {pmc.get('sequences_per_launch')} sequences per launch there"
                                        if traffic is not None and pfs else None), launches=cnt_,
                        avg_launch_us=round(ms * 1e3 / cnt_, 2), bytes_per_launch=by / cnt_, measured=where)

        # GEMM launches carry per-shape labels ("gemm<epi1,256x256> M=288000 N=3072 K=768 taps=1"): the encoder's are the ones whose
        # M is the group's windows x 1500 / 1501 / 3000 rows (layers, conv2, conv1); all of them fold back into the family entry
        nwin = min(args.whisper_group, 6 * B) if pipeline else B
        enc_rows = {f" M={nwin * r} " for r in (1500, 1501, 3000)}
        enc_fl = enc_ms = 0.0
        enc_n = 0
        folded = []
        recs0 = recs
        for name, fl, by, ms in recs:
            if name.startswith("gemm<"):
                if any(t in name for t in enc_rows):
                    enc_fl += fl; enc_ms += ms; enc_n += 1
                name = "gemm_bf16_nt_kernel"
            elif name.startswith("layernorm_kernel"):
                name = "layernorm_kernel"
            folded.append((name, fl, by, ms))
        recs = folded
        agg = aggregate_records(recs)
        pagg = aggregate_records([("gemm_bf16_nt_kernel" if n.startswith("gemm<") else n, f, b_, m) for n, f, b_, m in probe])
        # per-step totals: eager kernels as recorded over the timed steps; graph-resident decode kernels = probe average
        # launch time x their launches per step
        where_eager = "HIP events on the launch stream, one extra step of the same batch right after the timed region"
        per_step = {k: (v[3] / prof_steps, where_eager) for k, v in agg.items() if v[1] > 0 or v[2] > 0}
        for k, v in pagg.items():
            if k.startswith("dec_") and (v[1] > 0 or v[2] > 0):
                launches = v[0] / (probe_steps) * decode_steps_per_step
                # (scaled from the probe's lane size to the timed group's: these kernels' time is proportional to the rows they stream)
                per_step[k] = (v[3] / v[0] * launches,
                               f"HIP events on the lanes' streams, eager re-run of {probe_steps} decode steps of the {Bd}-sequence group of the "
                               f"timed region on the encoded batch ({nl_real} lane(s); graph-resident in the timed region)")
        roof = roof_mfma = None
        if per_step:
            name = max(per_step, key=lambda k: per_step[k][0])
            src = pagg if name.startswith("dec_") else agg
            cnt_, fl, by, ms = src[name]
            roof = roof_entry(name, cnt_, fl, by, ms, per_step[name][1])
            roof["ms_per_step"] = round(per_step[name][0], 2)
            if probe_x is not None and name.startswith("dec_"):
                xr = [(fl_, by_, ms_) for n_, fl_, by_, ms_ in probe_x if n_ == name]
                if xr:
                    roof["probe_distinct_grid"] = dict(sequences=Bp, launches=len(xr), avg_launch_us=round(sum(r_[2] for r_ in xr) * 1e3 / len(xr), 2),
                                                       bytes_per_launch=sum(r_[1] for r_ in xr) / len(xr),
                                                       what=f"the same eager probe on a {Bp}-sequence group: a grid no other launch of this process has "
                                                            "(the row to look for in profiles/*_default_kernel_stats_by_grid.csv)")
            if name.startswith(("dec_cross_stream_kernel", "dec_xs_stream_kernel")) and in_situ:
                # the same kernel inside the replayed step graphs with every lane of the group running: bytes of a lane's launch /
                # median stamp-to-stamp time, averaged over the lanes
                gbs = float(np.mean([l["gbs"] for l in in_situ]))
                roof["achieved_in_situ"] = round(gbs, 1)
                roof["frac_in_situ"] = round(gbs / PEAK_HBM_GBS, 4)
                # time per pipeline step of this kernel at the in-situ launch time: launches per step x median lane launch (the per-grid
                # rocprofv3 summary of the same command times the same launches, with the lanes serialised by the tracer)
                launches_step = pagg[name][0] / probe_steps * decode_steps_per_step
                roof["ms_per_step_in_situ"] = round(float(np.mean([l["median_us"] for l in in_situ])) * launches_step * 1e-3, 2)
                steps = [l["step_median_us"] for l in in_situ if l.get("step_median_us")]
                if steps:
                    # the WHOLE decode step of the group: every lane's cross-KV + the decoder weights each lane streams once per step
                    # (self-KV is < 1 % at the trace's <= 24 positions), all lanes concurrent, over the median lane step time
                    by_step = float(sum(decode_bytes_per_step(dims, l["sequences"], xstream) for l in in_situ))
                    roof["decode_step_in_situ"] = dict(bytes=by_step, median_us=round(float(np.mean(steps)), 1),
                                                       achieved=round(by_step / (float(np.mean(steps)) * 1e-6) / 1e9, 1),
                                                       frac=round(by_step / (float(np.mean(steps)) * 1e-6) / 1e9 / PEAK_HBM_GBS, 4), unit="GB/s",
                                                       what="all kernels of one decode step of the whole group (cross attention, self attention, "
                                                            "linears, logits, select), lanes concurrent, graph replay")
                roof["in_situ"] = dict(lanes=[{k: (round(v, 1) if isinstance(v, float) else v) for k, v in l.items()} for l in in_situ],
                                       measured="in-graph s_memrealtime stamps around every layer's cross attention (ccx_whisper_trace_lanes), "
                                                f"graph replay of a {Bd}-sequence group, all lanes concurrent; includes the ~2 us stamp nodes"
                                                + ("; with the cross attention against the encoder output the stamps bracket its THREE launches (LayerNorm + "
                                                   "query projection + expansion, the streaming kernel, merge + value projection: 17 + 110 + 8 us alone), and "
                                                   "`achieved_in_situ` divides the streaming kernel's bytes by the whole bracket" if xstream else ""))
        if "gemm_bf16_nt_kernel" in agg and (roof is None or roof["kernel"] != "gemm_bf16_nt_kernel"):
            cnt_, fl, by, ms = agg["gemm_bf16_nt_kernel"]
            roof_mfma = roof_entry("gemm_bf16_nt_kernel", cnt_, fl, by, ms, where_eager)
            roof_mfma["scope"] = "every GEMM launch of a step (encoder, ResNet-34, SepFormer, TDNN, heads, logits)"
            if enc_ms > 0:
                ach = enc_fl / (enc_ms * 1e-3) / 1e12
                roof_mfma["whisper_encoder_gemms"] = dict(achieved=round(ach, 2), unit="TFLOP/s", frac=round(ach / PEAK_MFMA_BF16_TFLOPS, 4),
                                                          launches=enc_n, ms_per_step=round(enc_ms / prof_steps, 2))
                # the WHOLE encoder: its GEMMs + attention + LayerNorms (every launch between log-mel and the decoder) against the
                # algorithmic GFLOP per window (encoder 344.2; + cross-KV projection 42.5 only where it is still computed)
                att_ms = sum(ms for n_, _, _, ms in recs if n_ == "enc_attention_kernel")
                ln_ms = sum(ms for n_, _, _, ms in recs0 if n_.startswith("layernorm_kernel M=") and f" M={nwin * dims.n_audio_ctx} " in n_ + " ")
                wins = (6 * B if pipeline else B) * prof_steps
                tot_ms = enc_ms + att_ms + ln_ms
                gfl_win = float(enc_flops_per_window(dims)[0] + (0 if xstream else enc_flops_per_window(dims)[1]))
                fl_tot = wins * gfl_win
                ach_t = fl_tot / (tot_ms * 1e-3) / 1e12
                roof_mfma["whisper_encoder_total"] = dict(achieved=round(ach_t, 2), unit="TFLOP/s", frac=round(ach_t / PEAK_MFMA_BF16_TFLOPS, 4),
                                                          gemm_ms=round(enc_ms / prof_steps, 2), attention_ms=round(att_ms / prof_steps, 2),
                                                          layernorm_ms=round(ln_ms / prof_steps, 2), windows_per_step=wins // prof_steps,
                                                          gflop_per_window=round(gfl_win / 1e9, 1))
        if roof_mfma is not None:
            # what the matrix cores of this part sustain at the phased GEMM's footprint (tools/microbench_mfma_ceiling.hip, committed once
            # per round): nothing but MFMAs on random operands, and the same with the main loop's LDS operand reads beside them
            cf = sorted((ROOT / "profiles").glob("r*_mfma_ceiling.json"))
            if cf:
                cj = json.loads(cf[-1].read_text())
                ceil = cj["mfma_with_lds_operand_reads"]["tflops"]
                roof_mfma["measured_ceiling"] = dict(file=f"profiles/{cf[-1].name}", bare_mfma_tflops=cj["bare_mfma"]["tflops"],
                                                     with_lds_operand_reads_tflops=ceil, shader_clock_mhz=cj["mfma_with_lds_operand_reads"]["shader_clock_mhz"],
                                                     what="v_mfma_f32_16x16x32_bf16 only, 8 waves per CU, 128 accumulator registers, random bf16 operands")
                roof_mfma["frac_of_measured_ceiling"] = round(roof_mfma["achieved"] / ceil, 4)
                for k in ("whisper_encoder_gemms", "whisper_encoder_total"):
                    if k in roof_mfma:
                        roof_mfma[k]["frac_of_measured_ceiling"] = round(roof_mfma[k]["achieved"] / ceil, 4)
        stage_ms = {k: round(v[3] / prof_steps, 3) for k, v in agg.items()}
        stage_ms.update({k: round(v[0], 3) for k, v in per_step.items()})

        cpu = None
        if not args.no_cpu_baseline and world == 1:
            threads = min(os.cpu_count() or 1, 32)
            if pipeline:
                sdims = SepDims()
                msd = (dims, ck[1] if ck is not None else synthetic_whisper_state_dict(dims, seed=0), sdims,
                       synthetic_sepformer_state_dict(sdims, seed=1), synthetic_xvector_state_dict(seed=2),
                       synthetic_pyannet_state_dict(7, seed=3), synthetic_resnet34_state_dict(seed=5))
                xrt, sample = cpu_baseline_pipeline(msd, clips[0], rules, threads)
            else:
                w = cpu_baseline_whisper(dims, sd, clips[0], rules, threads, tokens=args.sample_len)
                xrt = 30.0 / w["total"]
                sample = f"1 x 30 s clip: log-mel + small.en encoder + {w['n_tok']}-token greedy decode, fp32 torch, {w['total']:.1f} s"
            cpu = dict(value=round(xrt, 3), unit="xRT (audio-sec/wall-sec)", cores=threads, kind="port",
                       method="extrapolated: bounded op samples x one clip's op counts" if pipeline else "measured end to end on one clip",
                       sample=sample)
            if pipeline:
                # one full clip WAS timed end to end through the same oracle pipeline on a GPU box's host cores (tools/cpu_full_clip.py,
                # ~100 s of CPU work: once per round, committed).  That MEASURED figure is `value`; this run's bounded-sample
                # extrapolation stays beside it as `extrapolated`
                full = sorted((ROOT / "profiles").glob("r*_cpu_full_clip.json"))
                if full:
                    fc = json.loads(full[-1].read_text())
                    cpu = dict(value=fc["xrt"], unit="xRT (audio-sec/wall-sec)", cores=fc["threads"], kind="port",
                               method=f"measured end to end: ONE full 30 s clip through the oracle pipeline on {fc['threads']} host threads of a GPU box "
                                      f"(tools/cpu_full_clip.py, committed as profiles/{full[-1].name})",
                               sample=f"one full 30 s clip under the pinned schedule (6 Whisper calls x 224 tokens, 4 separator regions, 62 x-vector crops, "
                                      f"VAD + diarization): {fc['total_s']} s of CPU work = {fc['xrt']} xRT",
                               extrapolated=dict(value=round(xrt, 3), cores=threads, sample=sample, over_measured=round(xrt / fc["xrt"], 3),
                                                 method="this run: bounded op samples x one clip's op counts"))

        enc_f, cross_f = enc_flops_per_window(dims)
        cfg = {"workload": "full_pipeline_vad_diarize_separate_transcribe (BASELINE configs[3])" if pipeline
               else "whisper_small_en_logmel_encode_greedy_decode (BASELINE configs[1])",
               "clips_per_gpu": B, "clip_seconds": 30, "sample_len": args.sample_len, "whisper_calls": n_calls,
               "tokens_decoded": n_tokens, "parallelism": f"clip-sharded x{world}",
               "encoder_gflop_per_window": round((enc_f + (0 if xstream else cross_f)) / 1e9, 1),
               "cross_path": sorted({r.get("cross_path", "unknown") for r in res["records"]}) if res else None,
               "cross_attention": ("against the encoder output: one pass over xa per layer and sequence serves all heads, no K/V caches "
                                   "(csrc/cross_x.hip; decodes of <= 80 sequences keep per-layer K/V)" if xstream else "per-layer K/V caches (CCX_CROSS_X=0)")}
        if pipeline:
            cfg["schedule"] = ("pinned synthetic schedule (SURVEY.md 8d): per clip 2 regular + 2 overlap-bearing segments -> "
                               "6 Whisper windows, 4 separator regions, 62 x-vector crops; VAD (51 x 5 s chunks) and diarization "
                               "(21 x 10 s segmentation chunks + ResNet-34 embeddings of their local speakers) computed but not steering")
            cfg["schedule"] += ("; batches software-pipelined: the Whisper decode of batch i (HBM-bound) overlaps the front end + encoder "
                                "of batch i+1 (MFMA-bound) on a second stream, two Whisper instances alternating" if pipelined
                                else "; sequential: each batch start to finish")
            cfg["batch_schedule"] = args.schedule
            cfg["decode_span_batches"] = args.decode_span if pipelined else 1
            if pipelined and args.decode_span > 1:
                cfg["schedule"] += (f"; the Whisper windows of {args.decode_span} consecutive batches ({6 * B * args.decode_span} sequences) are encoded and "
                                    "decoded as one group: with that many sequences per decode lane the latency-bound step chain hides under the "
                                    "other lanes' HBM-bound cross attention")
            if span1_ms is not None:
                cfg["pipelined_span1_ms_per_step"] = round(span1_ms, 3)
                cfg["pipelined_span1_xrt"] = round(30.0 * B * world / (span1_ms * 1e-3), 2)
            if seq_ms is not None:
                cfg["sequential_ms_per_step"] = round(seq_ms, 3)
                cfg["sequential_xrt"] = round(30.0 * B * world / (seq_ms * 1e-3), 2)
            cfg["whisper_group"] = args.whisper_group
            cfg["stage_ms_per_step"] = {k: round(v, 2) for k, v in bp.stage_ms.items()} if bp.stage_ms else None
        out = {
            "metric": "xRT (audio-sec/wall-sec) end-to-end, 30 s 16 kHz clips",
            "value": round(value, 2),
            "unit": "xRT",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt * 1e3 / args.steps, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": f"synthetic clips (seed 1234+i), weights {weights}, greedy T=0",
            "config": cfg,
            "roofline": roof,
            "roofline_mfma": roof_mfma,
            "cpu_baseline": cpu,
            "kernel_ms_per_step": stage_ms,
            "ccx_env": dict(ccx_env, **({} if "CCX_PROF_SHAPES" in ccx_env else {"CCX_PROF_SHAPES": "1 (set by bench.py: labels of the profiled step only)"})),
            "model_load_ms": round(load_ms, 1),
            "weight_broadcast_ms": None if bcast_ms is None else round(bcast_ms, 1),
            "power": power,
            "hbm_used_gb": round((torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 1e9, 1),
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
