#!/usr/bin/env python3
"""bench.py -- xRT (audio-seconds / wall-second) of the MI355X-native ClearConverse hot path.

  python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic 30 s / 16 kHz clips that are
already resident in HBM: log-mel -> Whisper small.en encoder -> greedy
decode (<= 224 tokens, hipGraph-captured step chain).  Workload at N=1 is BASELINE.json configs[1]
("Whisper small.en encoder+greedy-decode only, batch=8x30 s clips") until the separator and the
speaker nets land; it is named in config.workload.  N>1: one process per GPU (torchrun), clips are
sharded across ranks with no data-path collective (weak scaling); the only collectives are the
timing barrier/max and one all-gather of the token records at the end of the timed region.

The JSON line carries `roofline` for the dominant kernel (per-launch HIP-event timing recorded by
libccx over the timed steps) and `cpu_baseline` (oracle/whisper_ref.py, fp32 torch on the host
cores, bounded sample) -- see DESIGN.md "Measurement".
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


def enc_flops_per_window(d):
    """SURVEY.md section 8d formula (encoder incl. conv stem) + cross-KV projection."""
    D, L, S = d.n_audio_state, d.n_audio_layer, d.n_audio_ctx
    enc = 2 * (3000 * D * 3 * d.n_mels + S * D * 3 * D + L * (S * (4 * D * D + 2 * D * 4 * D) + 2 * S * S * D))
    cross = 2 * d.n_text_layer * 2 * S * D * D
    return enc, cross


def cpu_baseline(dims, sd, clip, rules, threads, tokens):
    """Oracle (CPU restatement of the reference path, fp32 torch) on ONE 30 s clip: log-mel +
    encoder + `tokens` greedy steps with KV cache.  Returns audio-seconds per wall-second."""
    from oracle import whisper_ref as R
    torch.set_num_threads(threads)
    orc = R.WhisperRef(R.Dims(**dims.__dict__), sd)
    orules = R.Rules(suppress=tuple(rules.suppress))
    t0 = time.perf_counter()
    with torch.no_grad():
        mel = R.pad_or_trim(R.log_mel_spectrogram(torch.from_numpy(clip))[:, : len(clip) // 160], 3000)
        xa = orc.encode(mel[None])
        r = R.greedy_decode_cached(orc, xa, [rules.sot], orules, sample_len=tokens)
    dt = time.perf_counter() - t0
    return len(clip) / 16000.0 / dt, dt, len(r.tokens)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8, help="30 s clips per GPU per step")
    ap.add_argument("--sample-len", type=int, default=224)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torchrun with WORLD_SIZE={args.gpus} (got {world})")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from clearconverse_amd import _lib
    from clearconverse_amd.audio import synthetic_clip
    from clearconverse_amd.tokenizer import DecodeRules
    from clearconverse_amd.weights import WhisperDims, find_whisper_checkpoint, synthetic_whisper_state_dict
    from clearconverse_amd.whisper import WhisperModel

    B = args.batch
    ck = find_whisper_checkpoint("small.en")
    if ck is not None:
        dims, sd = ck
        weights = "checkpoint"
    else:
        dims, sd = WhisperDims.small_en(), synthetic_whisper_state_dict(WhisperDims.small_en(), seed=0)
        weights = "synthetic-seed0"
    ctx = _lib.Context(local_rank)
    model = WhisperModel(dims, sd, max_batch=B, device=local_rank, ctx=ctx)
    rules = DecodeRules()

    # clip i of the global corpus -> rank i % world; B clips per rank
    clips = [synthetic_clip(rank + world * i, 30.0) for i in range(B)]
    n = [len(c) for c in clips]
    audio = torch.from_numpy(np.stack(clips)).cuda(local_rank).contiguous()   # resident in HBM before timing
    prompts = [[rules.sot] for _ in range(B)]

    def step():
        model.log_mel(audio, n)
        model.encode(B)
        return model.decode_greedy(prompts, sample_len=args.sample_len)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ctx.prof_enable(True)
    t0 = time.perf_counter()
    n_tokens = 0
    for _ in range(args.steps):
        res = step()
        n_tokens += sum(len(r["tokens"]) for r in res)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    recs = ctx.prof_records()
    ctx.prof_enable(False)

    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # transcripts (token ids) gathered on every rank: the one data-path collective (C2)
        rec = torch.full((B, args.sample_len), rules.eot, dtype=torch.int32, device="cuda")
        for b, r in enumerate(res):
            rec[b, : len(r["tokens"])] = torch.tensor(r["tokens"], dtype=torch.int32)
        gathered = [torch.empty_like(rec) for _ in range(world)]
        dist.all_gather(gathered, rec)
        tk = torch.tensor([n_tokens], device="cuda", dtype=torch.int64)
        dist.all_reduce(tk)
        n_tokens = int(tk.item())

    audio_s = 30.0 * B * world * args.steps
    value = audio_s / dt

    if rank == 0:
        # ---- roofline of the dominant eagerly-launched kernel (HIP events, timed region) ----
        agg = {}
        for name, fl, by, ms in recs:
            a = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
            a[0] += 1; a[1] += fl; a[2] += by; a[3] += ms
        roof = None
        if agg:
            name = max(agg, key=lambda k: agg[k][3])
            cnt, fl, by, ms = agg[name]
            if name in ("gemm_bf16_nt_kernel", "enc_attention_kernel"):
                ach = fl / (ms * 1e-3) / 1e12
                roof = dict(kernel=name, bound="mfma", achieved=round(ach, 2), peak=PEAK_MFMA_BF16_TFLOPS, unit="TFLOP/s",
                            frac=round(ach / PEAK_MFMA_BF16_TFLOPS, 4), traffic=None, launches=cnt,
                            avg_launch_us=round(ms * 1e3 / cnt, 2), flops_per_launch=fl / cnt)
            else:
                ach = by / (ms * 1e-3) / 1e9
                roof = dict(kernel=name, bound="hbm", achieved=round(ach, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                            frac=round(ach / PEAK_HBM_GBS, 4), traffic=None, launches=cnt,
                            avg_launch_us=round(ms * 1e3 / cnt, 2), bytes_per_launch=by / cnt)
        stage_ms = {k: round(v[3] / args.steps, 3) for k, v in agg.items()}

        cpu = None
        if not args.no_cpu_baseline:
            threads = min(os.cpu_count() or 1, 32)
            xrt, secs, ntok = cpu_baseline(dims, sd, clips[0], rules, threads, tokens=args.sample_len)
            cpu = dict(value=round(xrt, 3), unit="xRT (audio-sec/wall-sec)", cores=threads, kind="port",
                       sample=f"1 x 30 s clip: log-mel + small.en encoder + {ntok}-token greedy decode, fp32 torch, {secs:.1f} s")

        enc_f, cross_f = enc_flops_per_window(dims)
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
            "config": {"workload": "whisper_small_en_logmel_encode_greedy_decode (BASELINE configs[1])",
                       "clips_per_gpu": B, "clip_seconds": 30, "sample_len": args.sample_len,
                       "tokens_decoded": n_tokens, "parallelism": f"clip-sharded x{world}",
                       "encoder_gflop_per_window": round((enc_f + cross_f) / 1e9, 1)},
            "roofline": roof,
            "cpu_baseline": cpu,
            "kernel_ms_per_step": stage_ms,
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
