#!/bin/bash
# trip bo: the one failure of the last soak (run_bwd_case 401643, 'sink_mid' + causal, dq 0.34): which engine, which rows
O=gpurun_out/r6bo; mkdir -p $O
timeout 300 python3 tools/lab/bwd_sink_mid_probe.py 2>&1 | tail -12 | tee $O/bwd_sink_mid_probe.txt
