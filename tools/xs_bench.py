"""Stand-alone timing of the X-stream cross attention (csrc/cross_x.hip) through ccx_cross_attention_xa: per-launch HIP-event times
of its three kernels for a range of row counts (one row per sequence), full small.en width.
usage: python tools/xs_bench.py [rows ...]"""
import sys
import numpy as np
import torch
from clearconverse_amd import _lib

def main():
    rows_list = [int(a) for a in sys.argv[1:]] or [128, 256, 368, 384, 512, 768]
    ctx = _lib.Context(0)
    lib = _lib.load()
    H, S = 12, 1500
    D = 64 * H
    g = torch.Generator().manual_seed(0)
    wk = (torch.randn(D, D, generator=g) / D ** 0.5).numpy().astype(np.float32)
    wv = (torch.randn(D, D, generator=g) / D ** 0.5).numpy().astype(np.float32)
    bv = np.zeros(D, np.float32)
    nmax = max(rows_list)
    xa = torch.randn(nmax, S, D, generator=g).to(torch.bfloat16).cuda()
    q = torch.randn(nmax, D, generator=g).cuda()
    out = torch.empty(nmax, D, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for rows in rows_list:
        for it in range(3):
            if it == 2:
                ctx.prof_enable(True)
            ctx.check(lib.ccx_cross_attention_xa(ctx.handle, q.data_ptr(), wk.ctypes.data, wv.ctypes.data, bv.ctypes.data, xa.data_ptr(), None,
                                                 0, rows, rows, H, S, out.data_ptr(), st))
        torch.cuda.synchronize()
        recs = ctx.prof_records()
        ctx.prof_enable(False)
        line = f"rows {rows:4d}:"
        for name, fl, by, ms in recs[-3:]:
            line += f"  {name.split('<')[0][4:]} {ms * 1e3:7.1f} us"
            if "stream" in name:
                line += f" ({by / (ms * 1e-3) / 1e12:.2f} TB/s)"
        print(line, flush=True)

if __name__ == "__main__":
    main()
