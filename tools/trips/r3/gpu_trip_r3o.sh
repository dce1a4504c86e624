#!/bin/bash
O=gpurun_out/r3o; mkdir -p $O
timeout 600 python tools/lab/region_probe.py > $O/region_probe.json 2>$O/err.txt; cat $O/region_probe.json; tail -3 $O/err.txt
