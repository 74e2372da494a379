"""Timeline of the decode lanes from a rocprofv3 kernel trace taken with CCX_NO_GRAPH=1 (eager launches: the tracer serialises
hipGraph replays on different streams, eager launches keep their overlap).

  rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tl -- python3 bench.py --schedule sequential --steps 1 --warmup 1 \
      --sample-len 24 --no-cpu-baseline        (with CCX_NO_GRAPH=1 in the environment)
  python3 tools/lane_timeline.py /tmp/tl/.../tl_kernel_trace.csv

Prints, for the decode kernels only: per-queue busy share, per-kernel mean duration alone vs while a cross-attention kernel of
ANOTHER queue is running, and the mean gap a queue leaves between two kernels in both situations."""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r.get("Kernel_Name") or r.get("Name")
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        q = r.get("Stream_Id") or r.get("Queue_Id")
        rows.append((s, e, name, q))
rows.sort()
is_dec = lambda n: n.startswith("void dec_") or n.startswith("dec_") or "gemm_bf16_nt_kernel<3" in n
dec = [r for r in rows if is_dec(r[2])]
if not dec:
    sys.exit("no decode kernels in the trace")
# restrict to the longest run of decode kernels (the timed step's decode)
lo, hi = dec[len(dec) // 4][0], dec[-1][1]
dec = [r for r in dec if r[0] >= lo]
cross = [r for r in dec if "dec_attention_kernel<false>" in r[2] or "dec_cross_stream_kernel" in r[2]]
print(f"decode window {(hi - lo) / 1e6:.1f} ms, {len(dec)} kernels, {len(cross)} cross-attention launches")
byq = defaultdict(list)
for r in dec:
    byq[r[3]].append(r)
for q, rr in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e, _, _ in rr)
    print(f"queue {q}: {len(rr)} kernels, busy {busy / 1e6:.1f} ms = {busy / (hi - lo):.2f} of the window")
import bisect
cs = sorted((s, e, q) for s, e, _, q in cross)
starts = [c[0] for c in cs]

def overlapped(s, e, q):
    """share of [s, e) during which a cross-attention kernel of another queue runs"""
    i = bisect.bisect_left(starts, s) - 8
    cov = 0
    for j in range(max(0, i), len(cs)):
        cs_, ce_, cq_ = cs[j]
        if cs_ >= e:
            break
        if cq_ == q:
            continue
        a, b = max(s, cs_), min(e, ce_)
        if b > a:
            cov += b - a
    return cov / max(1, e - s)

stat = defaultdict(lambda: [0, 0.0, 0, 0.0])
for s, e, n, q in dec:
    ov = overlapped(s, e, q)
    k = n.split("(")[0][:70]
    st = stat[k]
    if ov > 0.5:
        st[2] += 1; st[3] += e - s
    elif ov < 0.05:
        st[0] += 1; st[1] += e - s
print(f"{'kernel':72s} {'alone n':>8s} {'us':>7s} {'beside cross n':>15s} {'us':>7s}")
for k, (n0, t0, n1, t1) in sorted(stat.items(), key=lambda kv: -(kv[1][1] + kv[1][3])):
    print(f"{k:72s} {n0:8d} {t0 / max(1, n0) / 1e3:7.2f} {n1:15d} {t1 / max(1, n1) / 1e3:7.2f}")
g = [0, 0.0, 0, 0.0]
for q, rr in byq.items():
    for a, b in zip(rr, rr[1:]):
        gap = b[0] - a[1]
        if gap < 0 or gap > 200000:
            continue
        ov = overlapped(a[1], max(b[0], a[1] + 1), q)
        if ov > 0.5:
            g[2] += 1; g[3] += gap
        else:
            g[0] += 1; g[1] += gap
print(f"gap between consecutive kernels of a queue: alone {g[1] / max(1, g[0]) / 1e3:.2f} us (n={g[0]}), beside another queue's cross attention {g[3] / max(1, g[2]) / 1e3:.2f} us (n={g[2]})")
