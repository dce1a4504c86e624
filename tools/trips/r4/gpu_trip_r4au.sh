#!/bin/bash
O=gpurun_out/r4au; mkdir -p $O
export TMPDIR=/tmp
for a in "1 8 1 131072 128" "1 4 1 65536 128" "1 2 1 131072 128" "1 4 1 8192 128" "1 2 16 32768 128" "2 2 1 16384 64"; do timeout 60 python tools/bench_decode.py $a 2>/dev/null | tail -1; done | tee $O/decode_kmax64.txt
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_fuzz.py -m gpu -q 2>&1 | tail -2
