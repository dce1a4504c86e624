#!/bin/bash
# trip m: the whole GPU suite, smoke(), and the driver's bench command on the round's build
O=gpurun_out/r6m; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/gpu_suite.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
