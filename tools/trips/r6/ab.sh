#!/bin/bash
# trip ab: decode-shaped launches after the split-KV fold's 16-byte layout + read-ahead; the tests that cover the fold
O=gpurun_out/r6ab; mkdir -p $O
for sh in "8 32 1 8192 128" "1 32 1 8192 128" "1 32 1 32768 128" "32 32 1 2048 128" "8 32 1 8192 64" "4 32 8 8192 128" "16 8 1 4096 128" "1 8 1 131072 128" "64 8 1 1024 128" "1 2 4096 4096 128" "2 8 1024 1024 128"; do
  timeout 120 python3 tools/bench_decode.py $sh 2>&1 | tail -1
done | tee $O/decode.txt
timeout 1200 python3 -m pytest tests/test_gpu_forward.py tests/test_gpu_pv16_range.py tests/test_gpu_configs.py tests/test_gpu_cbal.py tests/test_gpu_streams.py -m gpu -q 2>&1 | tail -5 | tee $O/tests.txt
timeout 600 python3 tools/lab/value_fuzz.py 0 400 run_graph_case 2>&1 | tail -3 | tee $O/fuzz_graph.txt
