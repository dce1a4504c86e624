#!/bin/bash
# round 4, trip U: dS-store backward with key-major 16-byte stores + LDS-DMA GEMM: probe (parity + A/B), kernel stats; mask tests (few-block routing)
O=gpurun_out/r4u; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/lab/ds_store_probe.py > $O/ds_store_probe.txt 2>&1; tail -12 $O/ds_store_probe.txt | cut -c1-400
cd /tmp && UMFA_BWD_DS_STORE=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/tools/run_bwd.py 1 24 4096 128 20 > $GRAFT_REPO_ROOT/$O/run_bwd.txt 2>&1; cd $GRAFT_REPO_ROOT
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bwd_ds_store_kernel_stats.csv; cut -c1-160 $O/bwd_ds_store_kernel_stats.csv | head -8
rm -rf $O/trace; find $O -name "*.db" -delete
timeout 900 python -m pytest tests/test_gpu_backward.py -m gpu -q > $O/tests_bwd.txt 2>&1; tail -3 $O/tests_bwd.txt | cut -c1-300
