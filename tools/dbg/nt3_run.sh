cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gemm_gpu.py tests/test_linear_fusion_gpu.py tests/test_pct_gpu.py tests/test_pct_train_gpu.py tests/test_pointnet_gpu.py -x -q 2>&1 | tail -3
python tools/bench_pct_step.py 2>&1 | tail -1
