cd $GRAFT_REPO_ROOT
for m in 0 1 2; do echo "== CCX_GEMM_NARROW=$m"; CCX_GEMM_NARROW=$m timeout -k 10 500 python tools/gemm_shapes.py 2>&1 | grep "128x64\|256x64\|total"; done
CCX_GEMM_NARROW=1 timeout -k 10 300 python -m pytest tests/test_resnet_gpu.py -x -q 2>&1 | tail -1
CCX_GEMM_NARROW=2 timeout -k 10 300 python -m pytest tests/test_resnet_gpu.py -x -q 2>&1 | tail -1
