set -x
python -m pytest tests/test_pointnet_gpu.py tests/test_modules_gpu.py tests/test_edge_cases_gpu.py -q -x 2>&1 | tail -4
python tools/bench_pointnet_bwd.py 65536 2>&1 | tail -8
python tools/bench_pointnet_bwd.py 1048576 2>&1 | tail -8
