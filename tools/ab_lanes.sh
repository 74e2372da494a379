cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_whisper_gpu.py -x -q 2>&1 | tail -2
run() { timeout -k 10 300 python bench.py --workload whisper --batch 192 --sample-len 64 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['kernel'], r['avg_launch_us'], r['frac'])"; }
echo "== hoisted"; run; run
