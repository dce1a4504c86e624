#!/bin/bash
# trip bl: end-of-round records at the final build of the round (clean rebuild) -- the whole GPU suite, smoke, the driver's bench command (timed), the headline under
# rocprofv3 --kernel-trace --stats
O=gpurun_out/r6bl; mkdir -p $O
R=$PWD
timeout 3000 python3 -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/gpu_suite.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench.err ) 2>&1 | tail -4 | tee $O/bench_wall_time.txt; tail -c 300 $O/bench.err
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -- python3 $R/bench.py --steps 20 --warmup 5 --headline-only --no-graph > $R/$O/bench_under_rocprof.json 2>$R/$O/prof_err.txt )
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/fwd_kernel_stats.csv \;
rm -rf $O/trace
cut -c1-160 $O/fwd_kernel_stats.csv | head -6
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r6bl/bench_driver_command.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for k, v in d['configs'].items():
    if 'mask' in k or 'cfg2' in k: print(k, v.get('ms'), v.get('ms_128row_alone'), v.get('kernel'))
PY
