#!/bin/bash
# trip d: the whole GPU suite + the bench line on the build with the additive-mask kernels (MASKA)
O=gpurun_out/r6d; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/gpu_suite.txt
python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
timeout 300 python3 tools/lab/bias_probe.py 2>&1 | grep -v amdgpu | tee $O/bias_probe.jsonl
