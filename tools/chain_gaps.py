"""A `rocprofv3 --kernel-trace` CSV of a small-batch decode (e.g. `bench.py --workload whisper`) -> for every kernel symbol of the decode
chain: launches, average duration, and the average GAP between the end of the previous kernel and its start (same queue, gaps above
50 us dropped as host pauses).  Splits a latency-bound chain into execution and launch-boundary time.
    python tools/chain_gaps.py <..._kernel_trace.csv> [name filter, default dec_]"""
import collections
import csv
import sys


def norm(name):
    for pre in ("void ", "(anonymous namespace)::"):
        name = name.replace(pre, "")
    depth, out = 0, []
    for ch in name:
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).replace(", ", ",").strip()


def main():
    flt = sys.argv[2] if len(sys.argv) > 2 else "dec_"
    rows = []
    with open(sys.argv[1], newline="") as f:
        for r in csv.DictReader(f):
            rows.append((r.get("Queue_Id", "0"), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), norm(r["Kernel_Name"]), int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0)))
    rows.sort(key=lambda t: (t[0], t[1]))
    agg = collections.defaultdict(lambda: [0, 0, 0, 0])       # calls, dur, gap, gap count
    prev_q, prev_end = None, 0
    for q, t0, t1, name, grid in rows:
        key = (name, grid)
        a = agg[key]
        a[0] += 1; a[1] += t1 - t0
        if q == prev_q and 0 <= t0 - prev_end < 50000:
            a[2] += t0 - prev_end; a[3] += 1
        prev_q, prev_end = q, t1
    tot_d = tot_g = 0
    print(f"{'kernel':62s} {'grid':>9s} {'calls':>7s} {'avg us':>8s} {'gap us':>7s}")
    for (name, grid), (n, d, g, gn) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if flt not in name:
            continue
        print(f"{name[:62]:62s} {grid:9d} {n:7d} {d / n / 1e3:8.2f} {(g / gn / 1e3) if gn else float('nan'):7.2f}")
        tot_d += d; tot_g += g
    print(f"decode-chain kernels: execution {tot_d / 1e6:.1f} ms, gaps in front of them {tot_g / 1e6:.1f} ms")


if __name__ == "__main__":
    main()
