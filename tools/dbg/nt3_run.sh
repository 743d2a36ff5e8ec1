cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gemm_gpu.py tests/test_linear_fusion_gpu.py tests/test_pct_gpu.py tests/test_pct_train_gpu.py -x -q 2>&1 | tail -3
echo "--- three planes"; python tools/bench_gemm.py 163840 2>&1 | grep "M="; python tools/bench_pct_step.py 2>&1 | tail -1
echo "--- fp32 MFMA"; SGA_LIB_PATH=variants/libsga_ntfp32.so python tools/bench_gemm.py 163840 2>&1 | grep "M="; SGA_LIB_PATH=variants/libsga_ntfp32.so python tools/bench_pct_step.py 2>&1 | tail -1
