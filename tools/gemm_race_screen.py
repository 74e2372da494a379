"""Race screen of the phased GEMM main loop (LDS-DMA kept in flight across barriers): 300 launches per shape must be bit-identical.
Shapes cover 1, 2, 3, 12 and 48 K tiles, ragged M, a ragged last column tile and four epilogues."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clearconverse_amd import _lib
ctx = _lib.Context(0); lib = ctx.lib
st = int(torch.cuda.current_stream().cuda_stream)
bad = 0
for (M, N, K, epi) in [(73728, 3072, 768, 1), (73728, 768, 3072, 2), (65536, 1024, 128, 0), (70000, 1280, 192, 3), (100000, 768, 768, 2), (66000, 1104, 64, 5)]:
    g = torch.Generator().manual_seed(N + K)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).cuda()
    bias = torch.randn(N, generator=g).cuda()
    resid = torch.randn(M, N, generator=g).cuda()
    odt = torch.bfloat16 if epi in (0, 1, 5) else torch.float32
    first = None
    for it in range(300):
        out = torch.full((M, N), float("nan"), dtype=odt, device="cuda")
        rc = lib.ccx_gemm_bf16(ctx.handle, epi, A.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), out.data_ptr(), N,
                               resid.data_ptr() if epi == 2 else None, N, M, N, K, st)
        assert rc == 0
        if first is None: first = out
        elif not torch.equal(out, first):
            bad += 1; print("MISMATCH", M, N, K, epi, "launch", it, int((out != first).sum()))
    print("shape", M, N, K, "epi", epi, "300 launches identical" if bad == 0 else "BAD", flush=True)
print("total mismatching launches", bad)
