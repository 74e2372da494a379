cd $GRAFT_REPO_ROOT
run() { timeout -k 10 300 python bench.py --workload whisper --batch $1 --sample-len 64 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:v for k,v in d['kernel_ms_per_step'].items() if '3>' in k})"; }
for g in 0 1; do for b in 192 8; do echo "== logits gemm=$g batch=$b"; CCX_LOGITS_GEMM=$g run $b; done; done
CCX_LOGITS_GEMM=1 timeout -k 10 600 python -m pytest tests/test_whisper_gpu.py -x -q 2>&1 | tail -2
