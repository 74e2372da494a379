# Round profiles (run on the GPU box from the repo root):  bash tools/collect_profiles.sh r02
# 1. rocprofv3 --kernel-trace --stats of the default bench command -> gpurun_out/<tag>_pipeline_kernel_stats.csv
# 2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of a short run -> gpurun_out/<tag>_pmc_traffic.json
# Counters are collected in their own runs (no trace domains beside --pmc), as the MI355X guide prescribes.
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/prof_ks /tmp/prof_f /tmp/prof_w
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -o ks -- python3 bench.py --schedule sequential --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_bench_under_rocprof.log 2>&1 || exit 1
cp "$(find /tmp/prof_ks -name 'ks_kernel_stats.csv' | head -1)" gpurun_out/${tag}_pipeline_b32_kernel_stats.csv || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -o f -- python3 bench.py --schedule sequential --steps 1 --warmup 0 --sample-len 4 --no-cpu-baseline > /tmp/pmc_f.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -o w -- python3 bench.py --schedule sequential --steps 1 --warmup 0 --sample-len 4 --no-cpu-baseline > /tmp/pmc_w.log 2>&1 || exit 1
python3 tools/pmc_to_json.py "$(find /tmp/prof_f -name 'f_counter_collection.csv' | head -1)" "$(find /tmp/prof_w -name 'w_counter_collection.csv' | head -1)" gpurun_out/${tag}_pmc_traffic.json
# 3. one SQ pass (8 slots): LDS bank conflicts, wave-parked / issue-stall / active fractions, MFMA busy -> gpurun_out/<tag>_pmc_sq.json
rm -rf /tmp/prof_sq
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/prof_sq -o s -- python3 bench.py --schedule sequential --steps 1 --warmup 0 --sample-len 4 --no-cpu-baseline > /tmp/pmc_s.log 2>&1 || exit 1
python3 tools/pmc_sq_to_json.py "$(find /tmp/prof_sq -name 's_counter_collection.csv' | head -1)" gpurun_out/${tag}_pmc_sq.json
