set -x
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_bf16x6_gpu.py tests/test_onepass_gpu.py tests/test_loss_gpu.py tests/test_c3_gpu.py -q -x 2>&1 | tail -4
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_l_c3_stats -- python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-hits --no-attr --no-c2 --no-pct --no-exact < /dev/null > gpurun_out/r06_l_c3_bench_under_rocprof.json 2> gpurun_out/r06_l_c3_stats.err
python tools/prof_summary.py gpurun_out/r06_l_c3_stats gpurun_out/r06_l_c3_kernel_stats.csv > /dev/null
rm -rf gpurun_out/r06_l_c3_stats
head -12 gpurun_out/r06_l_c3_kernel_stats.csv | cut -c1-200
