"""Can the HBM-bound decode stream and the MFMA-bound GEMMs run on DISJOINT sets of CUs at the same time?  hipExtStreamCreateWithCUMask
streams (mask bit i -> XCD i % 8, CU i / 8 of it: the first n bits are n / 8 CUs of every XCD): the X-stream cross attention
(ccx_cross_attention_xa, 256 rows) on the first n CUs, the encoder's fc1 GEMM (ccx_gemm_bf16) on the other 256 - n -- each alone on its
mask, then both together.  usage: PYTHONPATH=. python tools/cu_mask_probe.py [n ...]"""
import ctypes as C, math, sys, time
import numpy as np
import torch
from clearconverse_amd import _lib

hip = C.CDLL("libamdhip64.so")


def masked_stream(lo, hi):
    """stream restricted to CUs [lo, hi) of the 256 (bit order as above)"""
    words = (C.c_uint32 * 8)()
    for i in range(lo, hi):
        words[i // 32] |= 1 << (i % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, f"hipExtStreamCreateWithCUMask: {rc}"
    return s.value


def main():
    ns = [int(a) for a in sys.argv[1:]] or [256, 192, 160, 128, 96, 64]
    ctx = _lib.Context(0)
    lib = _lib.load()
    H, S, rows = 12, 1500, 256
    D = 64 * H
    g = torch.Generator().manual_seed(0)
    wk = (torch.randn(D, D, generator=g) / D ** 0.5).numpy().astype(np.float32)
    wv = (torch.randn(D, D, generator=g) / D ** 0.5).numpy().astype(np.float32)
    bv = np.zeros(D, np.float32)
    xa = torch.randn(rows, S, D, generator=g).to(torch.bfloat16).cuda()
    q = torch.randn(rows, D, generator=g).cuda()
    out = torch.empty(rows, D, device="cuda")
    M, N, K = 288000, 3072, 768
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).cuda()
    bias = torch.randn(N, generator=g).cuda()
    gout = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")

    def xs(st, n):
        for _ in range(n):
            ctx.check(lib.ccx_cross_attention_xa(ctx.handle, q.data_ptr(), wk.ctypes.data, wv.ctypes.data, bv.ctypes.data, xa.data_ptr(), None,
                                                 0, rows, rows, H, S, out.data_ptr(), st))

    def gemm(st, n):
        for _ in range(n):
            rc = lib.ccx_gemm_bf16(ctx.handle, 1, A.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), gout.data_ptr(), N, None, N, M, N, K, st)
            assert rc == 0

    def timed(fn):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    NX, NG = 60, 12
    # the cross attention converts its weights on the host per call (~ms): time the kernels through the per-launch profile instead
    def xs_kernel_us(st):
        xs(st, 2); torch.cuda.synchronize()
        ctx.prof_enable(True); xs(st, 3); torch.cuda.synchronize()
        recs = ctx.prof_records(); ctx.prof_enable(False)
        v = [ms * 1e3 for name, fl, by, ms in recs if "xs_stream" in name]
        return sum(v) / len(v)

    for n in ns:
        sd = masked_stream(0, n)
        t_xs = xs_kernel_us(sd)
        line = f"decode mask {n:3d} CUs: stream kernel {t_xs:6.1f} us ({rows * 2.396e6 / t_xs / 1e6:.2f} TB/s)"
        if n < 256:
            sg = masked_stream(n, 256)
            gemm(sg, 2)
            t_g = timed(lambda: gemm(sg, NG)) / NG
            line += f" | fc1 on the other {256 - n}: {t_g:6.3f} ms ({2.0 * M * N * K / t_g / 1e9:.0f} TFLOP/s)"
            # together: the GEMMs are queued first (enqueue only), then cross attentions run until the GEMMs have finished; the stream
            # kernel is timed per launch (profile events on its own stream), the GEMMs by events on theirs
            tg = torch.cuda.ExternalStream(sg)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            ctx.prof_enable(True)
            e0.record(tg); gemm(sg, NG * 3); e1.record(tg)
            k = 0
            while not e1.query() and k < 400:
                xs(sd, 1); k += 1
            torch.cuda.synchronize()
            recs = ctx.prof_records(); ctx.prof_enable(False)
            v = [ms * 1e3 for name, fl, by, ms in recs if "xs_stream" in name]
            t_g_both = e0.elapsed_time(e1) / (NG * 3)
            line += f" | together: stream kernel {sum(v) / max(1, len(v)):6.1f} us over {len(v)} launches, fc1 {t_g_both:6.3f} ms"
        print(line, flush=True)


if __name__ == "__main__":
    main()
