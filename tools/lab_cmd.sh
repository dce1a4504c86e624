timeout 600 python -m pytest tests/test_gpu_w64.py -m gpu -x -q 2>&1 | tail -3
for s in "1 24 4096 128" "1 16 8192 128"; do UMFA_LIBRARY=tools/lab_bin/libMFAFFI_stamps.so python tools/w64_stamps.py $s | sed "s/^/$s: /"; done
python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
