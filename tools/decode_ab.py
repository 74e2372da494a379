"""A/B of decode-chain switches inside ONE process: a full-size Whisper instance, B windows encoded once, then for every
configuration (a dict of CCX_* variables that libccx reads per decode) `reps` decodes of `steps` tokens, timed with a device
synchronisation on both sides.  Usage: PYTHONPATH=. python tools/decode_ab.py B steps "K=V,K2=V2" "K=V" ... ("-" = default env)."""
import os, sys, time
import numpy as np, torch
from clearconverse_amd.audio import synthetic_clip
from clearconverse_amd.tokenizer import DecodeRules
from clearconverse_amd.weights import WhisperDims, synthetic_whisper_state_dict
from clearconverse_amd.whisper import WhisperModel

B, steps = int(sys.argv[1]), int(sys.argv[2])
configs = sys.argv[3:] or ["-"]
dims = WhisperDims.small_en()
had = os.environ.get("CCX_DEC_LNFREE")
os.environ.setdefault("CCX_DEC_LNFREE", "3")      # present at creation: the instance also builds the folded weights of the chain experiments
m = WhisperModel(dims, synthetic_whisper_state_dict(dims, seed=0), max_batch=B)
if had is None:
    del os.environ["CCX_DEC_LNFREE"]              # "-" then means the library's default
rules = DecodeRules()
clips = [synthetic_clip(i, 30.0) for i in range(8)]
dev = torch.from_numpy(np.stack(clips)).cuda().repeat(B // 8, 1).contiguous()
m.log_mel(dev, [480000] * B); m.encode(B)
prompts = [[rules.sot]] * B
base = None
for rnd in range(2):                          # two rounds: box drift shows as a difference between the rounds of one configuration
    for cfg in configs:
        env = dict(kv.split("=") for kv in cfg.split(",")) if cfg != "-" else {}
        for k, v in env.items():
            os.environ[k] = v
        r = m.decode_greedy(prompts, sample_len=steps)          # capture / warm-up
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(2):
            r = m.decode_greedy(prompts, sample_len=steps)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
        for k in env:
            del os.environ[k]
        toks = [x["tokens"] for x in r]
        if base is None:
            base = toks
        print(f"round {rnd} {cfg:40s} {dt * 1e3:8.1f} ms per decode  {dt * 1e6 / (steps + 1):7.1f} us per step  tokens {'same' if toks == base else 'DIFFERENT'} as the first configuration", flush=True)
