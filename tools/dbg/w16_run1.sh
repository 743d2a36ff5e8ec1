cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_c5_gpu.py -x -q > gpurun_out/w1_test.log 2>&1; tail -4 gpurun_out/w1_test.log
F="--config c5 --steps 6 --warmup 2 --no-cpu-baseline --no-hits --no-attr --no-c2 --no-pct --no-exact"
timeout 400 python bench.py $F > gpurun_out/w1_new.json 2> gpurun_out/w1_new.err; cp bench_extras.json gpurun_out/w1_new_extras.json
SGA_LIB_PATH=variants/libsga_w16old.so timeout 400 python bench.py $F > gpurun_out/w1_old.json 2> gpurun_out/w1_old.err; cp bench_extras.json gpurun_out/w1_old_extras.json
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/w1_prof -- python bench.py $F > /dev/null 2>&1
python tools/prof_summary.py gpurun_out/w1_prof gpurun_out/w1_new_kernel_stats.csv > /dev/null; rm -rf gpurun_out/w1_prof
head -c 600 gpurun_out/w1_new.json; echo; head -c 600 gpurun_out/w1_old.json; echo; head -14 gpurun_out/w1_new_kernel_stats.csv | cut -c1-150
