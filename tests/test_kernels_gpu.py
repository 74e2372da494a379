"""-m gpu: each hand-written HIP kernel, called through the C ABI, against a plain fp32 torch
reference of the same op computed on the CPU from the SAME bf16-rounded inputs."""
import math

import numpy as np
import pytest
import torch

from tests.conftest import within

pytestmark = pytest.mark.gpu


def _stream():
    return int(torch.cuda.current_stream().cuda_stream)


def _rel(a, b):
    a = a.double().flatten(); b = b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


# the last four rows take the phased 256 x 256 kernel (>= 224 tiles) with 1, 2, 3 and 12 K tiles, ragged M and a ragged last column tile
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 256, 128), (1500, 768, 768), (12000, 2304, 768), (77, 384, 3072),
                                   (16384, 1024, 64), (16300, 1024, 128), (14500, 1280, 192), (15000, 1104, 768)])
@pytest.mark.parametrize("epi", [0, 1, 2, 3, 5])
def test_gemm_bf16(ccx_ctx, M, N, K, epi):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + epi)
    A = (torch.randn(M, K, generator=g)).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn(N, generator=g)
    resid = torch.randn(M, N, generator=g)
    ref = A.float() @ W.float().T + bias
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    elif epi == 5:
        ref = torch.relu(ref)
    elif epi == 2:
        ref = ref + resid
    Ad, Wd, bd, rd = A.cuda(), W.cuda(), bias.cuda(), resid.cuda()
    out_dtype = torch.bfloat16 if epi in (0, 1, 5) else torch.float32
    out = torch.full((M, N), float("nan"), dtype=out_dtype, device="cuda")
    lib = ccx_ctx.lib
    rc = lib.ccx_gemm_bf16(ccx_ctx.handle, epi, Ad.data_ptr(), K, Wd.data_ptr(), K, bd.data_ptr(), out.data_ptr(), N,
                           rd.data_ptr() if epi == 2 else None, N, M, N, K, _stream())
    ccx_ctx.check(rc, "gemm")
    torch.cuda.synchronize()
    got = out.float().cpu()
    assert torch.isfinite(got).all()
    tol = 6e-3 if out_dtype == torch.bfloat16 else 2e-5   # bf16 output rounding: 2^-9 relative per element
    assert _rel(got, ref) < tol, _rel(got, ref)
    # element-wise: identical inputs, fp32 accumulate -> only summation order / output rounding differ
    atol = 0.05 if out_dtype == torch.bfloat16 else 2e-3
    assert float((got - ref).abs().max()) < atol


@pytest.mark.parametrize("epi", [1, 2])
def test_gemm_phased_repeatable(ccx_ctx, epi):
    """The phased main loop keeps LDS-DMA in flight across barriers: a misplaced wait shows as a tile that differs from run to
    run.  40 launches of an encoder-shaped problem (K = 768 and 3072) must be bit-identical, and right."""
    for (M, N, K) in [(36000, 3072, 768), (36000, 768, 3072)]:
        g = torch.Generator().manual_seed(N + epi)
        A = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
        W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).cuda()
        bias = torch.randn(N, generator=g).cuda()
        resid = torch.randn(M, N, generator=g).cuda()
        out_dtype = torch.bfloat16 if epi == 1 else torch.float32
        ref = A.float() @ W.float().T + bias
        ref = torch.nn.functional.gelu(ref) if epi == 1 else ref + resid
        first = None
        for it in range(40):
            out = torch.full((M, N), float("nan"), dtype=out_dtype, device="cuda")
            rc = ccx_ctx.lib.ccx_gemm_bf16(ccx_ctx.handle, epi, A.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), out.data_ptr(), N,
                                           resid.data_ptr() if epi == 2 else None, N, M, N, K, _stream())
            ccx_ctx.check(rc, "gemm")
            if first is None:
                first = out
                assert _rel(out.float(), ref) < (6e-3 if epi == 1 else 2e-5)
            else:
                assert torch.equal(out, first), f"launch {it} differs from launch 0"


def test_gemm_rejects_bad_k(ccx_ctx):
    a = torch.zeros(128, 96, dtype=torch.bfloat16, device="cuda")
    w = torch.zeros(128, 96, dtype=torch.bfloat16, device="cuda")
    o = torch.zeros(128, 128, dtype=torch.bfloat16, device="cuda")
    rc = ccx_ctx.lib.ccx_gemm_bf16(ccx_ctx.handle, 0, a.data_ptr(), 96, w.data_ptr(), 96, None, o.data_ptr(), 128, None, 0,
                                   128, 128, 96, _stream())
    assert rc != 0 and b"multiple of 64" in ccx_ctx.lib.ccx_last_error(ccx_ctx.handle)


@pytest.mark.parametrize("M,D", [(5, 128), (1500, 768), (12000, 768), (33, 1024)])
def test_layernorm(ccx_ctx, M, D):
    g = torch.Generator().manual_seed(M + D)
    x = torch.randn(M, D, generator=g) * 3 + 0.5
    gamma = 1 + 0.1 * torch.randn(D, generator=g)
    beta = 0.1 * torch.randn(D, generator=g)
    ref = torch.nn.functional.layer_norm(x, (D,), gamma, beta, 1e-5)
    ob = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
    of = torch.empty(M, D, dtype=torch.float32, device="cuda")
    xd, gd, bd = x.cuda(), gamma.cuda(), beta.cuda()   # keep the device copies alive across the call
    rc = ccx_ctx.lib.ccx_layernorm(ccx_ctx.handle, xd.data_ptr(), gd.data_ptr(), bd.data_ptr(),
                                   ob.data_ptr(), of.data_ptr(), M, D, 1e-5, _stream())
    ccx_ctx.check(rc, "layernorm")
    torch.cuda.synchronize()
    within("layernorm_kernel: fp32 output max abs error", float((of.cpu() - ref).abs().max()), 3e-6)
    assert float((ob.float().cpu() - ref).abs().max()) < 0.03


@pytest.mark.parametrize("B,H,S", [(1, 2, 64), (2, 3, 200), (1, 12, 1500)])
def test_enc_attention(ccx_ctx, B, H, S):
    g = torch.Generator().manual_seed(B * 100 + H * 10 + S)
    Spad = (S + 127) // 128 * 128
    q = torch.randn(B, H, S, 64, generator=g).to(torch.bfloat16)
    k = torch.randn(B, H, S, 64, generator=g).to(torch.bfloat16)
    v = torch.randn(B, H, S, 64, generator=g).to(torch.bfloat16)
    # a few large scores so the online-softmax rescale path is exercised at a late key tile
    k[:, :, S - 3] *= 6.0
    sc = (q.float() @ k.float().transpose(-1, -2)) * 0.125
    ref = torch.softmax(sc, dim=-1) @ v.float()          # [B,H,S,64]
    ref = ref.permute(0, 2, 1, 3).reshape(B * S, H * 64)
    qp = torch.zeros(B, H, Spad, 64, dtype=torch.bfloat16); qp[:, :, :S] = q
    kp = torch.zeros(B, H, Spad, 64, dtype=torch.bfloat16); kp[:, :, :S] = k
    vt = torch.zeros(B, H, 64, Spad, dtype=torch.bfloat16); vt[:, :, :, :S] = v.transpose(-1, -2)
    o = torch.full((B * S, H * 64), float("nan"), dtype=torch.bfloat16, device="cuda")
    qd, kd, vd = qp.cuda(), kp.cuda(), vt.cuda()
    rc = ccx_ctx.lib.ccx_enc_attention(ccx_ctx.handle, qd.data_ptr(), kd.data_ptr(), vd.data_ptr(),
                                       o.data_ptr(), B, H, S, Spad, _stream())
    ccx_ctx.check(rc, "enc_attention")
    torch.cuda.synchronize()
    got = o.float().cpu()
    assert torch.isfinite(got).all()
    # P is rounded to bf16 before P.V and the output is bf16: ~2^-8 relative
    within("enc_attention_kernel: output rel-L2", _rel(got, ref), 4e-3)
    assert float((got - ref).abs().max()) < 0.06


def test_gather_rows_ragged(ccx_ctx):
    """ccx_gather_rows: ragged crops (arbitrary 4-byte-aligned starts, lengths 1 .. 9000) into a padded batch, bit-exact;
    columns past a crop's length stay untouched."""
    g = torch.Generator().manual_seed(5)
    src = torch.randn(3, 20000, generator=g).cuda()
    spans = [(0, 0, 9000), (1, 3, 1), (2, 19999, 1), (0, 1234, 4097), (1, 7, 1024), (2, 5000, 1023)]
    crops = [src[b, s:s + n] for b, s, n in spans]
    ptrs = torch.tensor([c.data_ptr() for c in crops], dtype=torch.int64).cuda()
    lens = torch.tensor([n for _, _, n in spans], dtype=torch.int32).cuda()
    dst = torch.full((len(spans), 9000), -7.0, device="cuda")
    ccx_ctx.check(ccx_ctx.lib.ccx_gather_rows(ccx_ctx.handle, ptrs.data_ptr(), lens.data_ptr(), len(spans), 9000, dst.data_ptr(), 9000,
                                              _stream()), "ccx_gather_rows")
    torch.cuda.synchronize()
    for i, (b, s, n) in enumerate(spans):
        assert torch.equal(dst[i, :n], src[b, s:s + n])
        assert bool((dst[i, n:] == -7.0).all())


def test_row_variance_cosine_rows_and_speaker_profiles(ccx_ctx):
    """The small reductions of the speaker-profile stage (reference back/api.py:939 torch.var, 946-953 weighted sum, 878-879 cosine
    similarity) through the C ABI, against torch on the host in fp64; a row's result must not depend on its batch mates."""
    g = torch.Generator().manual_seed(5)
    lens = [480000, 2, 12345, 144000, 1025, 64000, 7]
    stride = 480000
    x = torch.zeros(len(lens), stride)
    for i, n in enumerate(lens):
        x[i, :n] = torch.randn(n, generator=g) * (0.05 + 0.1 * i) + (0.3 if i == 2 else 0.0)      # one row with a large mean
    xd = x.cuda()
    nd = torch.tensor(lens, dtype=torch.int32, device="cuda")
    var = torch.empty(len(lens), device="cuda")
    ccx_ctx.check(ccx_ctx.lib.ccx_row_variance(ccx_ctx.handle, xd.data_ptr(), stride, nd.data_ptr(), len(lens), var.data_ptr(), _stream()), "ccx_row_variance")
    torch.cuda.synchronize()
    for i, n in enumerate(lens):
        ref = float(x[i, :n].double().var(unbiased=True))
        within("row_variance: relative error vs fp64 torch.var", abs(float(var[i]) - ref) / ref, 1.2e-7, i)
    one = torch.empty(1, device="cuda")
    ccx_ctx.check(ccx_ctx.lib.ccx_row_variance(ccx_ctx.handle, xd[3:4].contiguous().data_ptr(), stride, nd[3:4].contiguous().data_ptr(), 1, one.data_ptr(), _stream()), "ccx_row_variance")
    assert float(one[0]) == float(var[3])
    # cosine similarity: own rows, and against a repeating pair of rows
    a = torch.randn(37, 512, generator=g); b = torch.randn(37, 512, generator=g) * 3.0
    a[5] = 0.0                                                              # a zero row: 0 / eps -> 0
    ad, bd = a.cuda(), b.cuda()
    out = torch.empty(37, device="cuda")
    ccx_ctx.check(ccx_ctx.lib.ccx_cosine_rows(ccx_ctx.handle, ad.data_ptr(), bd.data_ptr(), 37, 512, 37, out.data_ptr(), _stream()), "ccx_cosine_rows")
    ref = torch.nn.functional.cosine_similarity(a.double(), b.double(), dim=1)
    within("cosine_rows: max abs error vs fp64", float((out.cpu().double() - ref).abs().max()), 6e-8)
    assert float(out[5]) == 0.0
    out2 = torch.empty(37, device="cuda")
    ccx_ctx.check(ccx_ctx.lib.ccx_cosine_rows(ccx_ctx.handle, ad.data_ptr(), bd.data_ptr(), 37, 512, 2, out2.data_ptr(), _stream()), "ccx_cosine_rows")
    ref2 = torch.nn.functional.cosine_similarity(a.double(), b[:2].double().repeat(19, 1)[:37], dim=1)
    assert float((out2.cpu().double() - ref2).abs().max()) < 5e-7
    sub = torch.empty(3, device="cuda")
    ccx_ctx.check(ccx_ctx.lib.ccx_cosine_rows(ccx_ctx.handle, ad[10:13].contiguous().data_ptr(), bd[10:13].contiguous().data_ptr(), 3, 512, 3, sub.data_ptr(), _stream()), "ccx_cosine_rows")
    assert torch.equal(sub, out[10:13])                                     # bit-identical whatever the batch
    # speaker profiles: 3 clips x 4 turns (A B A B) x 512
    emb = torch.randn(3, 4, 512, generator=g); w = torch.rand(3, 4, generator=g) + 0.1
    spk = torch.tensor([0, 1, 0, 1], dtype=torch.int32)
    ed, wd, sd = emb.cuda(), w.cuda(), spk.cuda()
    prof = torch.empty(3, 2, 512, device="cuda")
    ccx_ctx.check(ccx_ctx.lib.ccx_speaker_profiles(ccx_ctx.handle, ed.data_ptr(), wd.data_ptr(), sd.data_ptr(), 3, 4, 512, 2, prof.data_ptr(), _stream()), "ccx_speaker_profiles")
    for c in range(3):
        for s_ in range(2):
            cols = [t for t in range(4) if int(spk[t]) == s_]
            ref = sum(emb[c, t].double() * (w[c, t].double() / w[c, cols].double().sum()) for t in cols)
            assert float((prof[c, s_].cpu().double() - ref).abs().max()) < 1e-6
