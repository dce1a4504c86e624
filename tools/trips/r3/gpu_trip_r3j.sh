#!/bin/bash
# round 3, trip J: rehearsal of the N > 1 bench code path on one device (gloo, all ranks on cuda:0) -- not a measurement
O=gpurun_out/r3j; mkdir -p $O
UMFA_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline --headline-only > $O/bench_rehearsal_n2.json 2>$O/err_n2.txt
tail -c 1500 $O/bench_rehearsal_n2.json; tail -5 $O/err_n2.txt
UMFA_BENCH_ONE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 4 --steps 4 --warmup 2 --no-cpu-baseline --headline-only > $O/bench_rehearsal_n4.json 2>$O/err_n4.txt
tail -c 1200 $O/bench_rehearsal_n4.json; tail -5 $O/err_n4.txt
