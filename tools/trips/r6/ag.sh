#!/bin/bash
# trip ag: the routing sweep again (tools/lab/routing_random_probe.py: random launch sizes, dispatcher / w64 forced / 128-row forced) -- the 128-row kernel
# changed this round (balanced causal pairs, the split-KV fold's read-ahead and the plan's price of a part), so its cost model is refitted
O=gpurun_out/r6ag; mkdir -p $O
cd tools/lab
for s in 101 102 103 104 105 106 201 202; do timeout 900 python3 routing_random_probe.py $s 150 2>/dev/null | grep shape > ../../$O/routing_random_bf16_seed$s.jsonl; done
for s in 107 108 203; do timeout 900 python3 routing_random_probe.py $s 150 fp16 2>/dev/null | grep shape > ../../$O/routing_random_fp16_seed$s.jsonl; done
wc -l ../../$O/*.jsonl | tail -1
