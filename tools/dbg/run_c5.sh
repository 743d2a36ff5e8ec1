cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r05_n_bench_c5.json 2> gpurun_out/r05_n_bench_c5.err
tail -c 300 gpurun_out/r05_n_bench_c5.err
python tools/dbg/show_bench.py gpurun_out/r05_n_bench_c5.json
timeout 600 python -m pytest tests/test_c5_gpu.py tests/test_wide16_gpu.py -q 2>&1 | tail -2
