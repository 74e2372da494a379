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
    # csrc/gemm_bf16.hip: 128-byte rows; A rows natural with keyA, W rows permuted with keyB; chunk = 4 ks + h
    key_a = lambda r: (r >> 1) & 7
    key_b = lambda r: ((r >> 1) & 1) | (((r >> 4) & 3) << 1)
    for ks in range(2):
        a = [(l & 15) * 128 + (((4 * ks + (l >> 4)) ^ key_a(l & 15)) << 4) for l in range(64)]
        assert conflicts_b128(a)[0] == 1
        for j in range(4):
            rows = [_perm_row(l & 15, j) for l in range(64)]
            b = [rows[l] * 128 + (((4 * ks + (l >> 4)) ^ key_b(rows[l])) << 4) for l in range(64)]
            assert conflicts_b128(b)[0] == 1


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
