set -x
python -m pytest tests/test_bf16x6_gpu.py tests/test_loss_gpu.py "tests/test_fp64_chunked_gpu.py::test_headline_loss_gradient_vs_fp64[1024]" -q -x 2>&1 | tail -4
python tools/bench_sweep.py 1024 128 2>&1 | tail -1
python tools/bench_sweep.py 512 64 2>&1 | tail -1
