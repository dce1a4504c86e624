#!/bin/bash
# trip ba: health check of the restored tree (third session of the round): GPU suite, smoke, the driver's bench command
O=gpurun_out/r6ba; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee $O/gpu_suite.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.txt
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
