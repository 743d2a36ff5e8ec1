set -x
python -m pytest tests/test_pointnet_gpu.py tests/test_modules_gpu.py tests/test_engine_gpu.py tests/test_edge_cases_gpu.py -q -x 2>&1 | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python tools/bench_pointnet_bwd.py 65536 2>&1 | tail -9
python tools/bench_pointnet_bwd.py 1048576 2>&1 | tail -9
python bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline --no-hits --no-attr --no-pct --no-exact 2>/dev/null | tail -1 | cut -c1-400
