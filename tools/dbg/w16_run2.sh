cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
tag=$1
timeout 600 python -m pytest tests/test_c5_gpu.py -x -q > gpurun_out/${tag}_test.log 2>&1; tail -4 gpurun_out/${tag}_test.log
F="--config c5 --steps 6 --warmup 2 --no-cpu-baseline --no-hits --no-attr --no-c2 --no-pct --no-exact"
timeout 400 python bench.py $F > gpurun_out/${tag}_new.json 2> gpurun_out/${tag}_new.err; cp bench_extras.json gpurun_out/${tag}_new_extras.json
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python bench.py $F > /dev/null 2>&1
python tools/prof_summary.py gpurun_out/${tag}_prof gpurun_out/${tag}_new_kernel_stats.csv > /dev/null; rm -rf gpurun_out/${tag}_prof
head -c 400 gpurun_out/${tag}_new.json; echo; grep -E "wide16|anchor_kernel" gpurun_out/${tag}_new_kernel_stats.csv | cut -c1-160
