timeout 600 python -m pytest tests/test_gpu_w64.py -m gpu -x -q 2>&1 | tail -5
for s in "1 24 4096 128" "1 8 4096 128" "1 16 4096 128" "1 32 4096 128" "1 16 8192 128"; do python tools/bench_one.py $s; done
