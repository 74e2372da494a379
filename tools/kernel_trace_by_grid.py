"""rocprofv3 --kernel-trace CSV -> per (kernel symbol, grid size) statistics.

`rocprofv3 --stats` aggregates by symbol only, but one symbol runs at several launch shapes in a bench.py process (the decode
lanes' cross attention covers 384 sequences per launch in the timed pipelined region, 64 in the sequential comparison steps), and
`roofline.achieved` is priced per launch shape.  Usage:
    python tools/kernel_trace_by_grid.py <..._kernel_trace.csv> <out.csv> [min_total_ms]
Columns: kernel, grid_x, workgroup_x, calls, total_ms, avg_us, min_us, max_us, share_of_gpu_time."""
import collections
import csv
import sys


def norm(name: str) -> str:
    for pre in ("void ", "(anonymous namespace)::"):
        name = name.replace(pre, "")
    depth, out = 0, []
    for ch in name:                      # cut the argument list: the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).replace(", ", ",").strip()


def main():
    src, dst = sys.argv[1], sys.argv[2]
    floor_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
    agg = collections.defaultdict(lambda: [0, 0, 1 << 62, 0])
    total = 0
    with open(src, newline="") as f:
        for r in csv.DictReader(f):
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            k = (norm(r["Kernel_Name"]), int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0), int(r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or 0))
            a = agg[k]
            a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
            total += d
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "grid_x", "workgroup_x", "calls", "total_ms", "avg_us", "min_us", "max_us", "share_of_gpu_time"])
        for (name, g, wg), (n, t, lo, hi) in rows:
            if t / 1e6 < floor_ms:
                continue
            w.writerow([name, g, wg, n, f"{t / 1e6:.3f}", f"{t / n / 1e3:.2f}", f"{lo / 1e3:.2f}", f"{hi / 1e3:.2f}", f"{t / max(total, 1):.4f}"])
    for (name, g, wg), (n, t, lo, hi) in rows[:12]:
        print(f"{name[:70]:70s} grid {g:9d} calls {n:7d} total {t / 1e6:9.2f} ms avg {t / n / 1e3:9.2f} us")


if __name__ == "__main__":
    main()
