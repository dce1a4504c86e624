#!/bin/bash
# trip ai: end-of-round records -- the whole GPU suite, smoke, the driver's bench command, the headline under rocprofv3 --kernel-trace --stats,
# kernel trace of a paired / unpaired causal launch, a 2500-seed soak of every fuzz leg
O=gpurun_out/r6ai; mkdir -p $O
R=$PWD
timeout 3000 python3 -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/gpu_suite.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench.err; tail -c 300 $O/bench.err
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -- python3 $R/bench.py --steps 20 --warmup 5 --headline-only --no-graph > $R/$O/bench_under_rocprof.json 2>$R/$O/prof_err.txt )
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/fwd_kernel_stats.csv \;
rm -rf $O/trace
cat > $O/run_causal.py <<'PY'
import sys
sys.path[:0] = ['/root/repo', '/root/repo/universal-metal-flash-attention_amd']
import torch, umfa_torch
q, k, v = (torch.randn(1, 8, 4096, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3))
o = torch.empty(1, 8, 4096, 128, device="cuda", dtype=torch.float32)
with umfa_torch.options(no_w64=1, cbal=int(sys.argv[1])):
    for _ in range(30):
        umfa_torch.attention_forward(q, k, v, causal=True, out=o)
torch.cuda.synchronize()
PY
for c in 1 2; do
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_c$c -- python3 $R/$O/run_causal.py $c > /dev/null 2>>$R/$O/prof_err.txt )
find $O/trace_c$c -name "*kernel_stats.csv" -exec cp {} $O/causal_B1_H8_S4096_D128_cbal${c}_kernel_stats.csv \;
rm -rf $O/trace_c$c
done
(time timeout 2400 python3 tools/lab/value_fuzz.py 20000 2500) 2>&1 | tail -8 | tee $O/soak_2500_seeds_all_legs.txt
