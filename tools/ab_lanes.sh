cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CCX_DEC_LANES=2 CCX_LANE_FAKE=1
for v in BASE=1 CCX_LANE_NOPOLL=1 CCX_LANE_NOSTAGGER=1; do
export $v
echo "== $v"
rm -rf /tmp/pl; rocprofv3 --kernel-trace -d /tmp/pl -o w -- python3 bench.py --workload whisper --batch 192 --sample-len 32 --steps 1 --warmup 0 --no-cpu-baseline > /tmp/pl.log 2>&1
python3 tools/lane_overlap.py /tmp/pl/w_results.db lane_probe_spin | tail -1
cp /tmp/pl/w_results.db gpurun_out/fake_${v%%=*}.db
unset ${v%%=*}
done
