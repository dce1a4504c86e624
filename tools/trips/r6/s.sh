#!/bin/bash
# trip s: in-kernel stamps of the balanced causal pairs at config 2 (tools/fwd_lab.hip, -DUMFA_LAB_CBAL)
O=gpurun_out/r6s; mkdir -p $O
for d in 0 1 2; do echo "== cbal delta $d"; timeout 60 tools/lab_bin/fwd_lab_c1 16 1024 30 4 $d; done 2>&1 | tee $O/stamps_cbal.txt
echo "== unpaired"; timeout 60 tools/lab_bin/fwd_lab_c0 16 1024 30 4 2>&1 | tee $O/stamps_plain.txt
