set -x
python -m pytest tests/test_bf16x6_gpu.py tests/test_loss_gpu.py tests/test_onepass_gpu.py tests/test_modules_gpu.py "tests/test_fp64_chunked_gpu.py::test_headline_loss_gradient_vs_fp64[1024]" tests/test_fullsize_gpu.py -q -x 2>&1 | tail -12
for i in 1 2; do
SGA_BF16X6_SUMS_LITE=0 python tools/bench_sweep.py 1024 128 2>&1 | tail -1
python tools/bench_sweep.py 1024 128 2>&1 | tail -1
done
SGA_BF16X6_SUMS_LITE=0 python tools/bench_sweep.py 512 64 2>&1 | tail -1
python tools/bench_sweep.py 512 64 2>&1 | tail -1
