set -x
python -m pytest tests/test_bf16x6_gpu.py tests/test_loss_gpu.py tests/test_c3_gpu.py tests/test_onepass_gpu.py "tests/test_fp64_chunked_gpu.py::test_headline_loss_gradient_vs_fp64[1024]" tests/test_dist_gpu.py::test_bench_gpus2_creates_its_two_ranks -q -x 2>&1 | tail -15
for i in 1 2; do
SGA_LIB_PATH=variants/libsga_prev.so python tools/bench_sweep.py 1024 128 2>&1 | tail -1
python tools/bench_sweep.py 1024 128 2>&1 | tail -1
done
SGA_LIB_PATH=variants/libsga_prev.so python tools/bench_sweep.py 512 64 2>&1 | tail -1
python tools/bench_sweep.py 512 64 2>&1 | tail -1
