"""rocprofv3 counter_collection CSV of one SQ pass -> per-kernel sums and ratios:
python tools/pmc_sq_to_json.py <counter_collection.csv> <out.json>
Ratios: lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (extra cycles / all LDS-array cycles);
wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES (waves parked in s_waitcnt / barriers); issue_stall_frac = SQ_WAIT_INST_ANY /
SQ_WAVE_CYCLES; active_frac = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES; mfma_busy_per_wave_cycle = SQ_VALU_MFMA_BUSY_CYCLES /
(4 x SQ_WAVE_CYCLES) (the SQ wave counters are in quad-cycles, MI355X_MICROARCH.md)."""
import csv, json, sys, collections

per = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        for pre in ("void ", "(anonymous namespace)::"):
            name = name.replace(pre, "")
        name = name.split("(")[0]
        per[name][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[name].add(r["Dispatch_Id"])
out = {}
for name, c in per.items():
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    e = {"launches": len(launches[name]), **{k: v for k, v in sorted(c.items())}}
    if c.get("SQ_LDS_IDX_ACTIVE", 0.0) > 0:
        e["lds_conflict_frac"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
    if wc > 0:
        e["wait_frac"] = c.get("SQ_WAIT_ANY", 0.0) / wc
        e["issue_stall_frac"] = c.get("SQ_WAIT_INST_ANY", 0.0) / wc
        e["active_frac"] = c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
        e["mfma_busy_per_wave_cycle"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * wc)
    out[name] = e
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0.0))[:14]:
    print(f"{k[:56]:56s} n {v['launches']:6d} ldsconf {v.get('lds_conflict_frac', 0):6.3f} wait {v.get('wait_frac', 0):5.2f} stall {v.get('issue_stall_frac', 0):5.2f} "
          f"active {v.get('active_frac', 0):5.2f} mfma/wavecyc {v.get('mfma_busy_per_wave_cycle', 0):6.3f}")
