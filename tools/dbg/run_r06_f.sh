set -x
python -m pytest tests/test_bf16x6_gpu.py tests/test_loss_gpu.py tests/test_onepass_gpu.py tests/test_modules_gpu.py "tests/test_fp64_chunked_gpu.py::test_headline_loss_gradient_vs_fp64[1024]" tests/test_fullsize_gpu.py tests/test_c3_gpu.py tests/test_dist_gpu.py -q 2>&1 | tail -12
python bench.py > gpurun_out/r06_f_bench.out 2> gpurun_out/r06_f_bench.err; tail -c 3000 gpurun_out/r06_f_bench.out; cp bench_extras.json gpurun_out/r06_f_bench_extras.json
