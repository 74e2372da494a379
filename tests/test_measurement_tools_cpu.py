"""CPU: the measurement tooling that turns profiler output into the numbers of the bench line -- on synthetic inputs with known
answers (a wrong reduction here would misprice `roofline` silently)."""
import csv
import importlib
import json
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_kernel_trace_by_grid_groups_one_symbol_by_launch_shape(tmp_path):
    src = tmp_path / "ks_kernel_trace.csv"
    rows = [("void dec_cross_stream_kernel<true, 12, false>(DecAttnParams)", 1179648, 256, 0, 300_000),
            ("void dec_cross_stream_kernel<true, 12, false>(DecAttnParams)", 1179648, 256, 400_000, 690_000),
            ("void dec_cross_stream_kernel<true, 12, false>(DecAttnParams)", 1130496, 256, 700_000, 955_000),
            ("enc_attention_kernel(unsigned short const*, int)", 7077888, 256, 1_000_000, 2_700_000),
            ("void (anonymous namespace)::sg_stft_kernel(float const*)", 1024, 256, 0, 10)]
    with open(src, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kind", "Kernel_Name", "Start_Timestamp", "End_Timestamp", "Workgroup_Size_X", "Grid_Size_X"])
        for name, grid, wg, t0, t1 in rows:
            w.writerow(["KERNEL_DISPATCH", name, t0, t1, wg, grid])
    dst = tmp_path / "by_grid.csv"
    subprocess.run([sys.executable, str(ROOT / "tools" / "kernel_trace_by_grid.py"), str(src), str(dst), "0"], check=True, capture_output=True)
    got = {(r["kernel"], int(r["grid_x"])): r for r in csv.DictReader(open(dst))}
    a = got[("dec_cross_stream_kernel<true,12,false>", 1179648)]
    assert int(a["calls"]) == 2 and float(a["avg_us"]) == pytest.approx(295.0) and float(a["min_us"]) == 290.0 and float(a["max_us"]) == 300.0
    b = got[("dec_cross_stream_kernel<true,12,false>", 1130496)]
    assert int(b["calls"]) == 1 and float(b["avg_us"]) == 255.0
    assert ("enc_attention_kernel", 7077888) in got and ("sg_stft_kernel", 1024) in got          # argument lists and namespaces are cut
    assert sum(float(r["share_of_gpu_time"]) for r in got.values()) == pytest.approx(1.0, abs=1e-3)


def test_lane_trace_reduction_prices_the_cross_attention_per_lane(tmp_path, monkeypatch):
    """bench.lane_cross_attention_in_situ on a synthetic libccx lane trace: two lanes of 384 sequences, stamps (10 ns ticks << 8 | tag)
    with tag 1 before and tag 2 behind every cross attention (300 us / 320 us), tag 3 at the end of a step."""
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    bench = importlib.import_module("bench")
    def lane(cross_us, chain_us, steps=8, layers=12):
        t, out = 1000, []
        for _ in range(steps):
            for _l in range(layers):
                t += int(chain_us * 100); out.append((t << 8) | 1)
                t += int(cross_us * 100); out.append((t << 8) | 2)
            t += 500; out.append((t << 8) | 3)
        return out
    p = tmp_path / "trace.txt"
    with open(p, "w") as f:
        f.write("decode B 64 lanes 1\nlane 0 3 256 513 770\n")                  # an earlier decode: ignored
        f.write("decode B 768 lanes 2\n")
        for i, (c, ch) in enumerate(((300.0, 280.0), (320.0, 260.0))):
            st = lane(c, ch)
            f.write(f"lane {i} {len(st)} " + " ".join(map(str, st)) + "\n")
    per_seq = 12 * 1500 * 64 * 2 * 2
    got = bench.lane_cross_attention_in_situ(str(p), per_seq)
    assert [l["sequences"] for l in got] == [384, 384]
    assert got[0]["median_us"] == pytest.approx(300.0) and got[1]["median_us"] == pytest.approx(320.0)
    assert got[0]["gbs"] == pytest.approx(384 * per_seq / 300e-6 / 1e9, rel=1e-6)
    assert got[0]["step_median_us"] == pytest.approx(12 * 580.0 + 5.0)
    assert bench.lane_cross_attention_in_situ(str(tmp_path / "trace.txt"), per_seq) is not None
    empty = tmp_path / "empty.txt"; empty.write_text("")
    assert bench.lane_cross_attention_in_situ(str(empty), per_seq) is None


def test_bench_refuses_diagnostic_switches(monkeypatch):
    bench = importlib.import_module("bench")
    monkeypatch.setenv("CCX_ABLATE", "cross")
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "CCX_ABLATE" in str(e.value)


def test_pmc_reduction_doubles_fetch_and_reads_sequences_off_the_grid(tmp_path):
    def write(path, counter, rows):
        with open(path, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Dispatch_Id", "Kernel_Name", "Grid_Size", "Counter_Name", "Counter_Value"])
            for d, name, grid, val in rows:
                w.writerow([d, name, grid, counter, val])
    name = "void dec_cross_stream_kernel<true, 12, false>(DecAttnParams)"
    # two dispatches, counter split over 8 XCD rows each (rocprofv3 prints one row per instance)
    # ... plus ONE launch of another shape (bench.py's probe decodes a smaller group): it must not be averaged into the main entry
    write(tmp_path / "f.csv", "FETCH_SIZE", [(d, name, 384 * 12 * 256, 864000.0 / 8) for d in (1, 2) for _ in range(8)]
          + [(3, name, 336 * 12 * 256, 756000.0 / 8) for _ in range(8)])
    write(tmp_path / "w.csv", "WRITE_SIZE", [(d, name, 384 * 12 * 256, 576.0 / 8) for d in (1, 2) for _ in range(8)]
          + [(3, name, 336 * 12 * 256, 504.0 / 8) for _ in range(8)])
    out = tmp_path / "pmc.json"
    subprocess.run([sys.executable, str(ROOT / "tools" / "pmc_to_json.py"), str(tmp_path / "f.csv"), str(tmp_path / "w.csv"), str(out), "test note"],
                   check=True, capture_output=True)
    all_ = json.loads(out.read_text())
    d = all_["dec_cross_stream_kernel<true,12,false>"]
    assert d["launches"] == 2 and d["sequences_per_launch"] == 384 and d["sequences_per_launch_seen"] == [336, 384]
    assert all_["dec_cross_stream_kernel<true,12,false> @336"]["hbm_bytes_per_launch"] == pytest.approx((2 * 756000.0 + 504.0) * 1024.0)
    assert d["hbm_bytes_per_launch"] == pytest.approx((2 * 864000.0 + 576.0) * 1024.0)        # FETCH_SIZE counts half of a wide read on gfx950
    assert d["note"].startswith("test note")


def test_pmc_reduction_reads_the_rows_of_the_xstream_kernel_off_its_grid(tmp_path):
    """dec_xs_stream_kernel runs TWO blocks per row (the key halves): 256 rows = grid 131072; the probe's 240 rows stay beside it."""
    def write(path, counter, rows):
        with open(path, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Dispatch_Id", "Kernel_Name", "Grid_Size", "Counter_Name", "Counter_Value"])
            for d, name, grid, val in rows:
                w.writerow([d, name, grid, counter, val])
    name = "void dec_xs_stream_kernel<768, 1>(XsParams)"
    write(tmp_path / "f.csv", "FETCH_SIZE", [(d, name, 256 * 2 * 256, 292000.0 / 8) for d in (1, 2, 3) for _ in range(8)]
          + [(4, name, 240 * 2 * 256, 274000.0 / 8) for _ in range(8)])
    write(tmp_path / "w.csv", "WRITE_SIZE", [(d, name, 256 * 2 * 256, 18400.0 / 8) for d in (1, 2, 3) for _ in range(8)]
          + [(4, name, 240 * 2 * 256, 17250.0 / 8) for _ in range(8)])
    out = tmp_path / "pmc.json"
    subprocess.run([sys.executable, str(ROOT / "tools" / "pmc_to_json.py"), str(tmp_path / "f.csv"), str(tmp_path / "w.csv"), str(out), "test note"],
                   check=True, capture_output=True)
    all_ = json.loads(out.read_text())
    d = all_["dec_xs_stream_kernel<768,1>"]
    assert d["launches"] == 3 and d["sequences_per_launch"] == 256 and d["sequences_per_launch_seen"] == [240, 256]
    assert d["hbm_bytes_per_launch"] == pytest.approx((2 * 292000.0 + 18400.0) * 1024.0)
    assert all_["dec_xs_stream_kernel<768,1> @240"]["launches"] == 1


def test_bench_prices_the_cross_attention_with_the_bytes_of_the_formulation_it_runs():
    bench = importlib.import_module("bench")
    from clearconverse_amd.weights import WhisperDims
    d = WhisperDims.small_en()
    assert bench.cross_bytes_per_sequence(d, xstream=False) == 2 * 1500 * 768 * 2                       # K and V of one layer
    assert bench.cross_bytes_per_sequence(d, xstream=True) == 1500 * 768 * 2 + 12 * 768 * 2 + 2 * 12 * 768 * 4   # xa + q' + two partials
    w = 2 * (12 * (4 * 768 * 768 + 2 * 768 * 3072 + 4 * 768 * 768) + 51864 * 768)
    assert bench.decode_bytes_per_step(d, 768, True) == w + 768 * 12 * bench.cross_bytes_per_sequence(d, True)
    enc, cross = bench.enc_flops_per_window(d)
    assert round(enc / 1e9, 1) == 344.2 and round(cross / 1e9, 1) == 42.5
