"""rocprofv3 counter_collection CSVs (one FETCH_SIZE pass, one WRITE_SIZE pass) -> per-kernel HBM bytes per launch.
FETCH_SIZE is doubled: on gfx950 it reports half of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM /
rocprofv3 section); WRITE_SIZE is exact; both are in KB."""
import csv, json, sys, collections

GRID = {}      # kernel -> set of Grid_Size values seen
# the decode cross attention: sequences per launch = grid / 256 / blocks per sequence (12 heads for the K/V-cache kernel, 2 key halves
# for the kernel that streams the encoder output)
STREAMERS = ("dec_cross_stream_kernel", "dec_xs_stream_kernel")


def load(path, counter):
    per = collections.defaultdict(lambda: [0, 0.0])
    seen = set()
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"]
            for pre in ("void ", "(anonymous namespace)::"):
                name = name.replace(pre, "")
            name = name.split("(")[0].replace(", ", ",")      # "dec_cross_stream_kernel<true,12,false>": the label bench.py uses
            g = r.get("Grid_Size") or r.get("Grid_Size_X")
            if name.startswith(STREAMERS) and g:
                # one entry per launch shape (bench.py's probe decodes a slightly smaller group than the timed region): the shape with
                # the most launches becomes the kernel's entry below, the others stay beside it as "<name> @<sequences>"
                name = f"{name} @{int(g) // 256 // (2 if name.startswith('dec_xs_stream_kernel') else 12)}"
            key = (r["Dispatch_Id"], name)
            a = per[name]
            if key not in seen:
                seen.add(key); a[0] += 1
                if g:
                    GRID.setdefault(name, set()).add(int(g))
            a[1] += float(r["Counter_Value"])
    return per

fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
note = (sys.argv[4] if len(sys.argv) > 4 else
        "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --schedule sequential --steps 1 --warmup 0 --sample-len 4` "
        "(pipeline workload, 32 clips, 192-sequence decode group)") + ("; FETCH_SIZE doubled per MI355X_MICROARCH.md "
        "(gfx950 reports half of wide coalesced reads); KB units")
for name in sorted(set(fetch) | set(write)):
    n = max(fetch.get(name, [0, 0])[0], write.get(name, [0, 0])[0])
    if n == 0:
        continue
    fk = fetch.get(name, [0, 0.0])[1] / n
    wk = write.get(name, [0, 0.0])[1] / n
    out[name] = {"launches": n, "FETCH_SIZE_kb_per_launch": fk, "WRITE_SIZE_kb_per_launch": wk,
                 "hbm_bytes_per_launch": (2.0 * fk + wk) * 1024.0, "note": note}
# aggregates over template instantiations (bench.py labels the GEMM family by its base name)
base = collections.defaultdict(lambda: [0, 0.0, 0.0])
for name, v in out.items():
    if "<" in name:
        b = base[name.split("<")[0]]
        b[0] += v["launches"]; b[1] += v["FETCH_SIZE_kb_per_launch"] * v["launches"]; b[2] += v["WRITE_SIZE_kb_per_launch"] * v["launches"]
# the GEMM family is ONE label in bench.py ("gemm_bf16_nt_kernel"): the two-stage and the phased tile kernels together
if "gemm_bf16_phased_kernel" in base:
    a, b = base["gemm_bf16_nt_kernel"], base["gemm_bf16_phased_kernel"]
    a[0] += b[0]; a[1] += b[1]; a[2] += b[2]
for name, (n, fk, wk) in base.items():
    if name not in out:
        out[name] = {"launches": n, "FETCH_SIZE_kb_per_launch": fk / n, "WRITE_SIZE_kb_per_launch": wk / n,
                     "hbm_bytes_per_launch": (2.0 * fk + wk) / n * 1024.0, "note": note + "; all template instantiations together"}
# Counter collection serialises dispatches, so the decode-lane stream probe finds no concurrent stream and a decode group runs in
# ONE lane: the sequences a cross-attention launch covers are read off its grid.
shapes = collections.defaultdict(list)
for name in list(out):
    if name.startswith(STREAMERS) and " @" in name:
        base_name, seqs = name.split(" @")
        out[name]["sequences_per_launch"] = int(seqs)
        shapes[base_name].append((out[name]["launches"], int(seqs), name))
for base_name, lst in shapes.items():
    lst.sort(reverse=True)
    out[base_name] = dict(out[lst[0][2]], sequences_per_launch_seen=sorted(sq for _, sq, _ in lst))
json.dump(out, open(sys.argv[3], "w"), indent=1)
top = sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:8]
for k, v in top:
    print(f"{k[:60]:60s} launches {v['launches']:6d}  HBM MB/launch {v['hbm_bytes_per_launch'] / 1e6:10.2f}")
