#!/bin/bash
# round 5, trip k: fresh routing data at this build (random launch sizes, dispatcher / forced w64 / forced 128-row), for the cost-model fit
O=gpurun_out/r5k; mkdir -p $O
for s in 101 102 103 104 105 106; do timeout 900 python tools/lab/routing_random_probe.py $s 150 > $O/routing_random_bf16_seed$s.jsonl 2>> $O/err.txt; tail -1 $O/routing_random_bf16_seed$s.jsonl; done
for s in 107 108; do timeout 900 python tools/lab/routing_random_probe.py $s 120 fp16 > $O/routing_random_fp16_seed$s.jsonl 2>> $O/err.txt; tail -1 $O/routing_random_fp16_seed$s.jsonl; done
for dt in bf16 fp16; do timeout 900 python tools/lab/small_nqb_probe.py sweep $dt > $O/routing_sweep_$dt.jsonl 2>> $O/err.txt; wc -l $O/routing_sweep_$dt.jsonl; done
