#!/bin/bash
# trip bc: fp32 additive masks, second build (verdict folded into the list kernel, the 128-row route's tile flags from the classification pass, copy written for mixed tiles only)
O=gpurun_out/r6bc; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_w64_f32_mask.py -q -x 2>&1 | tail -25 | tee $O/tests_f32_mask.txt
timeout 1200 python3 -m pytest tests/test_gpu_w64_bias.py tests/test_gpu_w64_masks.py tests/test_gpu_value_fuzz.py tests/test_gpu_forward.py tests/test_gpu_routing.py -q 2>&1 | tail -8 | tee $O/tests_neighbours.txt
timeout 600 python3 tools/bench_mask_f32.py $O/mask_f32_timing.jsonl 2>&1 | tail -40
cd /tmp && export TMPDIR=/tmp
for kind in bias_f32 bias_f32_inexact additive_blockdiag; do
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_$kind -o t -- python3 $GRAFT_REPO_ROOT/tools/run_masked.py 12 $kind > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/$O/prof_$kind -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-160 "$f" | head -9 > $GRAFT_REPO_ROOT/$O/kernel_stats_$kind.csv
  rm -rf $GRAFT_REPO_ROOT/$O/prof_$kind
done
