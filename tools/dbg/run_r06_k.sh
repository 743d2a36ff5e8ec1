set -x
python -m pytest tests/test_bf16x6_gpu.py -q -x 2>&1 | tail -4
python tools/fuzz_parity.py 200 7 big 2>&1 | tail -3
python tools/fuzz_parity.py 100 8 2>&1 | tail -2
