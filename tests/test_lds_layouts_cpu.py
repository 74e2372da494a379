"""The XOR swizzle keys the kernels apply to their LDS images, checked against the bank model of tools/lds_bank_sim.py
(ds_read_b128 is serviced in four 16-lane groups; a 16-byte access occupies one of 16 slots of a 256-byte bank row).
Each case rebuilds the per-lane byte addresses of one fragment read exactly as the kernel computes them."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from lds_bank_sim import conflicts_b128  # noqa: E402


def _perm_row(l15, j):
    """MFMA operand row i = l15 of column tile j -> LDS row (a lane then owns 16 consecutive outputs)."""
    return 16 * (l15 >> 2) + 4 * j + (l15 & 3)


def test_gemm_operand_reads_are_conflict_free():
    # csrc/gemm_bf16.hip: 128-byte rows; A rows natural with keyA; W rows in the map of the output type (w_row<CM>) with
    # keyW<CM>; chunk = 4 ks + h
    key_a = lambda r: (r >> 1) & 7
    w_row = {1: lambda j, i: 16 * j + i,
             2: lambda j, i: 32 * (j >> 1) + 8 * (i >> 2) + 4 * (j & 1) + (i & 3)}
    key_w = {1: lambda r: (r >> 1) & 7,
             2: lambda r: ((r >> 1) & 1) | (((r >> 3) & 1) << 1) | (((r >> 4) & 1) << 2)}
    for ks in range(2):
        a = [(l & 15) * 128 + (((4 * ks + (l >> 4)) ^ key_a(l & 15)) << 4) for l in range(64)]
        assert conflicts_b128(a)[0] == 1
        for cm in (1, 2):
            for j in range(4):
                rows = [w_row[cm](j, l & 15) for l in range(64)]
                # the kernel computes the key once, from the rows of j = 0
                assert all(key_w[cm](rows[l]) == key_w[cm](w_row[cm](0, l & 15)) for l in range(64))
                b = [rows[l] * 128 + (((4 * ks + (l >> 4)) ^ key_w[cm](rows[l])) << 4) for l in range(64)]
                assert conflicts_b128(b)[0] == 1
    # every column of a wave's 64 is owned exactly once, and the phased kernel's W half b holds the rows of j in {2b, 2b+1}
    for cm in (1, 2):
        assert sorted(w_row[cm](j, i) for j in range(4) for i in range(16)) == list(range(64))
        for j in range(4):
            assert all((w_row[cm](j, i) >> 5) == (j >> 1) for i in range(16))
        col4 = (lambda j, h: 16 * j + 4 * h) if cm == 1 else (lambda j, h: 32 * (j >> 1) + 8 * h + 4 * (j & 1))
        for j in range(4):
            for h in range(4):
                # accumulator register r of a lane in quarter h = MFMA row 4h + r of block j
                assert [w_row[cm](j, 4 * h + r) for r in range(4)] == [col4(j, h) + r for r in range(4)]


def test_fused_ffn_weight_reads_are_conflict_free():
    # csrc/sepformer.hip sep_ffn_kernel: W1 part 256-byte rows (ff_key1), W2 part 128-byte rows (ff_key2)
    key1 = lambda r: (r & 3) | (((r >> 4) & 3) << 2)
    key2 = lambda r: ((r >> 1) & 1) | (((r >> 5) & 1) << 2)
    for ks in range(4):
        for j in range(4):
            rows = [_perm_row(l & 15, j) for l in range(64)]
            chunk = [8 * (ks >> 1) + 2 * (l >> 4) + (ks & 1) for l in range(64)]
            a = [rows[l] * 256 + ((chunk[l] ^ key1(rows[l])) << 4) for l in range(64)]
            assert conflicts_b128(a)[0] == 1
    for sx in range(2):
        for jo in range(8):
            rows = [64 * (jo >> 2) + _perm_row(l & 15, jo & 3) for l in range(64)]
            a = [rows[l] * 128 + (((2 * (l >> 4) + sx) ^ key2(rows[l])) << 4) for l in range(64)]
            assert conflicts_b128(a)[0] == 1
    # the key carried over from the GEMM (chunk laid out differently) was two-way conflicted here: keep the regression visible
    old = lambda r: ((r >> 1) & 1) | (((r >> 4) & 3) << 1)
    rows = [_perm_row(l & 15, 0) for l in range(64)]
    assert conflicts_b128([rows[l] * 128 + (((2 * (l >> 4)) ^ old(rows[l])) << 4) for l in range(64)])[0] == 2


def test_lstm_state_rows_are_conflict_free():
    # csrc/speaker.hip lstm_recurrent_kernel: H rows of 288 bytes, lane (row l15, k group hq) reads 16 bytes at 64 kk + 16 hq
    for kk in range(4):
        assert conflicts_b128([(l & 15) * 288 + 64 * kk + 16 * (l >> 4) for l in range(64)])[0] == 1
        assert conflicts_b128([(l & 15) * 272 + 64 * kk + 16 * (l >> 4) for l in range(64)])[0] == 2   # the first pitch tried


def test_fused_attention_block_reads_are_conflict_free():
    # csrc/sepformer.hip sep_attn_block_kernel: 256-byte rows (128 bf16), chunk c of row r stored at chunk c ^ (r & 15);
    # fragment reads: lane (l15, h4) reads row 16 t + l15, chunk 4 ks + h4
    off = lambda row, chunk: row * 256 + ((chunk ^ (row & 15)) << 4)
    for t in range(10):
        for ks in range(4):
            a = [off(16 * t + (l & 15), 4 * ks + (l >> 4)) for l in range(64)]
            assert conflicts_b128(a)[0] == 1
