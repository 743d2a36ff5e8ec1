set -x
python -m pytest tests/test_pointnet_gpu.py "tests/test_fp64_chunked_gpu.py::test_chunked_fp64_equals_oracle" "tests/test_fp64_chunked_gpu.py::test_headline_loss_gradient_vs_fp64[1024]" -q -x --durations=8 2>&1 | tail -25
python -c "
import json; r=json.load(open('gpurun_out/gradient_vs_fp64_1024.json')); print('fp64 phases', r['fp64_evaluation_seconds'])"
python tools/bench_pointnet_bn.py 2>&1 | tail -12
