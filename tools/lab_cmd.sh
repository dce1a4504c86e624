timeout 900 python -m pytest tests/test_gpu_w64.py tests/test_gpu_quantized.py -m gpu -q 2>&1 | tail -3
python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_mean'], d['int8'])"
for s in "1 16 8192 128" "2 24 4096 128" "4 16 8192 128 causal" "1 24 4096 128 causal"; do python tools/bench_one.py $s; done
