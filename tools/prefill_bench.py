"""How long is the prompt prefill of a 384-sequence decode group?  Times decode_greedy(sample_len=1) (prefill + the first token) for
the given prompt lengths (default 4, 10, 16; B=<sequences> in the environment), prefilled in passes of 16 positions (default) and fed
step by step (CCX_PREFILL=0)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clearconverse_amd.weights import WhisperDims, synthetic_whisper_state_dict
from clearconverse_amd.whisper import WhisperModel

B = int(os.environ.get("B", "384"))
dims = WhisperDims()
m = WhisperModel(dims, synthetic_whisper_state_dict(dims, seed=0), max_batch=B)
audio = torch.randn(B, 16000 * 30, device="cuda") * 0.05
m.log_mel(audio, [16000 * 30] * B)
m.encode(B)
torch.cuda.synchronize()
rng = np.random.default_rng(0)
for P in [int(a) for a in sys.argv[1:]] or (4, 10, 16):
    prompts = [[m.rules.sot_prev] + list(rng.integers(1000, 20000, P - 2)) + [m.rules.sot] for _ in range(B)]
    for mode in ("1", "0"):
        os.environ["CCX_PREFILL"] = mode
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            m.decode_greedy(prompts, sample_len=1)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        print(f"B={B} prompt length {P:2d} CCX_PREFILL={mode}: {dt:7.2f} ms (prefill + first token)", flush=True)
