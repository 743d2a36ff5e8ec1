cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
tag=$1
cp profiles/r06_pmc_traffic.csv gpurun_out/${tag}_pmc_traffic.csv
bash tools/pmc_traffic_c5.sh gpurun_out/${tag}_pmc_traffic.csv > gpurun_out/${tag}_pmc_c5.log 2>&1
cp gpurun_out/${tag}_pmc_traffic.csv profiles/r06_pmc_traffic.csv
timeout 600 python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline --no-attr --no-c2 --no-pct --no-exact > gpurun_out/${tag}_bench_c5.json 2> gpurun_out/${tag}_bench_c5.err
cp bench_extras.json gpurun_out/${tag}_bench_c5_extras.json
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python bench.py --config c5 --steps 6 --warmup 2 --no-cpu-baseline --no-hits --no-attr --no-c2 --no-pct --no-exact > gpurun_out/${tag}_c5_bench_under_rocprof.json 2>/dev/null
python tools/prof_summary.py gpurun_out/${tag}_prof gpurun_out/${tag}_c5_kernel_stats.csv > /dev/null; rm -rf gpurun_out/${tag}_prof
python tools/bench_wide16.py > gpurun_out/${tag}_bench_wide16.txt 2>&1; python tools/bench_wide16.py 4864 11520 3072 >> gpurun_out/${tag}_bench_wide16.txt 2>&1
head -c 1500 gpurun_out/${tag}_bench_c5.json; echo; tail -8 gpurun_out/${tag}_pmc_traffic.csv
