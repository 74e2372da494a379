# Round profiles (run on the GPU box from the repo root):  bash tools/collect_profiles.sh r03 [steps]
# 1. rocprofv3 --kernel-trace --stats of the DEFAULT bench schedule (pipelined, decode span 4: 768-sequence groups in 3 x 256-row
#    lanes) -> gpurun_out/<tag>_default_kernel_stats.csv (rocprofv3's own per-symbol summary) and
#    gpurun_out/<tag>_default_kernel_stats_by_grid.csv (the same trace reduced per (symbol, grid size): one symbol runs at several
#    launch shapes in a bench.py process), plus the JSON line of that run.
# 2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) at the 256-sequence launch shape of the cross attention
#    (whisper workload, --batch 256: counter collection serialises dispatches, so the group is decoded in one 384-row lane)
#    -> gpurun_out/<tag>_pmc_traffic.json
# Counters are collected in their own runs (no trace domains beside --pmc), as the MI355X guide prescribes; `python3` comes
# directly after `--`.
tag=${1:-r03}
steps=${2:-4}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/prof_ks /tmp/prof_f /tmp/prof_w
if [ "${3:-pmc}" != "pmconly" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -o ks -- python3 bench.py --steps $steps --warmup 4 --no-cpu-baseline --no-comparisons > gpurun_out/${tag}_bench_under_rocprof.log 2>&1 || exit 1
cp "$(find /tmp/prof_ks -name 'ks_kernel_stats.csv' | head -1)" gpurun_out/${tag}_default_kernel_stats.csv || exit 1
python3 tools/kernel_trace_by_grid.py "$(find /tmp/prof_ks -name 'ks_kernel_trace.csv' | head -1)" gpurun_out/${tag}_default_kernel_stats_by_grid.csv || exit 1
grep '^{"metric"' gpurun_out/${tag}_bench_under_rocprof.log > gpurun_out/${tag}_bench_under_rocprof.json
rm -rf /tmp/prof_ks
fi
if [ "${3:-pmc}" != "nopmc" ]; then
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -o f -- python3 bench.py --workload whisper --batch 256 --steps 1 --warmup 0 --sample-len 4 --no-cpu-baseline > /tmp/pmc_f.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -o w -- python3 bench.py --workload whisper --batch 256 --steps 1 --warmup 0 --sample-len 4 --no-cpu-baseline > /tmp/pmc_w.log 2>&1 || exit 1
python3 tools/pmc_to_json.py "$(find /tmp/prof_f -name 'f_counter_collection.csv' | head -1)" "$(find /tmp/prof_w -name 'w_counter_collection.csv' | head -1)" gpurun_out/${tag}_pmc_traffic.json \
  "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over \`bench.py --workload whisper --batch 256 --steps 1 --warmup 0 --sample-len 4\` (256 windows encoded, one 256-sequence decode lane)"
fi
