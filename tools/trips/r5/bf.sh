#!/bin/bash
# round 5, trip bc: the last build of round 5 -- whole GPU suite, smoke, full bench line, kernel trace of the headline, PMC traffic
O=gpurun_out/r5bf; mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; tail -6 $O/tests.txt | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5bf/bench.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms',d['ms_per_step'],'frac',d['roofline']['frac'],d['roofline'].get('frac_of_2516'),'cold',d['settle']['cold_start_ms_per_step'])
for k,v in d.get('configs',{}).items(): print(k, {a:b for a,b in v.items() if a in ('ms','rel','frac','kernel','frac_of_visible_work','error','rel_vs_quantised_oracle','mask_read_tbps_if_read_once')})
print(d['int8'].get('summary'))
print('cpu', json.dumps(d.get('cpu_baseline'))[:300])
PY
R=$GRAFT_REPO_ROOT
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -- python3 $R/bench.py --steps 20 --warmup 5 --headline-only --no-graph > $R/$O/bench_under_rocprof.json 2>$R/$O/prof_err.txt )
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -4 {} | cut -c1-200'
( cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -- python3 $R/tools/run_fwd.py 10 > /dev/null 2>>$R/$O/prof_err.txt )
( cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -- python3 $R/tools/run_fwd.py 10 > /dev/null 2>>$R/$O/prof_err.txt )
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_traffic.txt 2>&1; cat $O/pmc_traffic.txt | cut -c1-200
