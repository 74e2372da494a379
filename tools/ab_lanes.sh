cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_whisper_gpu.py tests/test_kernels_gpu.py -x -q 2>&1 | tail -2
timeout -k 10 500 python tools/gemm_shapes.py 2>&1 | grep -v amdgpu | head -8
