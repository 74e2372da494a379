"""Stand-alone timing of ccx_gemm_bf16 on the Whisper encoder's shapes (192 windows: M = 288 000), random operands.
CCX_GEMM_PHASED=0 selects the two-stage 256 x 256 kernel for an A/B run in a second process; GEMM_ITERS / GEMM_ONLY=<qkv|out|fc1|fc2>
for long single-shape runs (tools/power_sampler.py)."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clearconverse_amd import _lib

ctx = _lib.Context(0)
lib = ctx.lib
M = int(os.environ.get("GEMM_M", "288000"))
shapes = [("qkv", 0, 2304, 768), ("out", 2, 768, 768), ("fc1", 1, 3072, 768), ("fc2", 2, 768, 3072)]
st = int(torch.cuda.current_stream().cuda_stream)
only = os.environ.get("GEMM_ONLY")
for name, epi, N, K in shapes:
    if only and name != only:
        continue
    g = torch.Generator().manual_seed(N)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).cuda()
    bias = torch.randn(N, generator=g).cuda()
    out = torch.empty((M, N), dtype=torch.float32 if epi == 2 else torch.bfloat16, device="cuda")
    if epi == 2:
        out.normal_()
    def run():
        rc = lib.ccx_gemm_bf16(ctx.handle, epi, A.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), out.data_ptr(), N,
                               out.data_ptr() if epi == 2 else None, N, M, N, K, st)
        assert rc == 0, lib.ccx_last_error(ctx.handle)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = int(os.environ.get("GEMM_ITERS", "20"))
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{name:4s} epi{epi} M={M} N={N} K={K}: {ms:7.3f} ms  {2.0 * M * N * K / ms / 1e9:7.1f} TFLOP/s", flush=True)
