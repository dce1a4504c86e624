timeout 800 python -m pytest tests/test_gpu_w64.py -m gpu -x -q 2>&1 | tail -4
python tools/lab_dbg2.py 1 | tail -5
python tools/bench_one.py 1 24 4096 128 causal; python tools/bench_one.py 4 16 8192 128 causal
bash tools/ab_bench.sh
