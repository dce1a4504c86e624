#!/bin/bash
# trip bh: fp32 additive masks, sixth build (the V cast pass rides in the classification launch) -- tests, neighbours, timing, kernel trace, mask fuzz leg
O=gpurun_out/r6bh; mkdir -p $O; R=$PWD
timeout 1500 python3 -m pytest tests/test_gpu_w64_f32_mask.py -q 2>&1 | tail -25 | tee $O/tests_f32_mask.txt
timeout 1200 python3 -m pytest tests/test_gpu_w64_bias.py tests/test_gpu_w64_masks.py tests/test_gpu_value_fuzz.py tests/test_gpu_forward.py tests/test_gpu_routing.py -q 2>&1 | tail -8 | tee $O/tests_neighbours.txt
timeout 600 python3 tools/bench_mask_f32.py $O/mask_f32_timing.jsonl 2>&1 | cut -c1-200 | tail -40
for kind in bias_f32 bias_f32_inexact additive_blockdiag; do
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$kind -- python3 $R/tools/run_masked.py 12 $kind > /dev/null 2>>$R/$O/prof_err.txt )
  find $O/prof_$kind -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$kind.csv \;
  rm -rf $O/prof_$kind
  cut -c1-150 $O/kernel_stats_$kind.csv | head -8
done
(time timeout 1200 python3 tools/lab/value_fuzz.py 60000 2000 run_w64_mask_case) 2>&1 | tail -6 | tee $O/fuzz_w64_mask_leg_2000_seeds.txt
