timeout 900 python -m pytest tests/test_gpu_w64.py -m gpu -x -q 2>&1 | tail -6
bash tools/ab_bench.sh 2>&1 | tail -4
