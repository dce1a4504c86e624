#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_backward.py tests/test_gpu_sdpa.py tests/test_gpu_compat_surfaces.py -m gpu -q -x 2>&1 | tail -5
for a in "2 16 4096 64 bf16" "2 16 4096 64 bf16 causal" "1 16 8192 64 fp16" "4 32 2048 64 bf16" "1 24 4096 128 bf16"; do python tools/bench_bwd.py $a 2>&1 | grep -v amdgpu.ids; done
