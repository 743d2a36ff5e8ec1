set -x
python -m pytest tests/test_fp64_chunked_gpu.py -q -x --durations=5 2>&1 | tail -15
python -c "
import json
for n in (1024, 4096):
    r=json.load(open(f'gpurun_out/gradient_vs_fp64_{n}.json')); print(n, 'fp64 phases', r['fp64_evaluation_seconds'], r['meta_embedding_rel_err_vs_fp64_rel_to_own_max'])"
