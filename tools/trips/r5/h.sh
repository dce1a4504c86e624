#!/bin/bash
# round 5, trip h: tightened backward bar at FLUX size, new bench entries, the streaming rate of the box (hipMemcpy device-to-device)
O=gpurun_out/r5h; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_backward.py tests/test_gpu_smoke_entry.py -q -x 2>&1 | tail -6 | cut -c1-300
python3 - <<'PY'
import torch, time
for mb in (64, 126, 256):
    n = mb * (1 << 20)
    a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
    a.fill_(1)
    for _ in range(3): b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20
    print(f"d2d copy {mb} MB: {t*1e3:.1f} us, read+write {2*n/t/1e9:.2f} TB/s")
PY
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5h/bench.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms',d['ms_per_step'],'frac',d['roofline']['frac'],d['roofline'].get('frac_of_2516'))
for k,v in d.get('configs',{}).items():
    if 'window' in k or 'mask' in k: print(k, {a:b for a,b in v.items() if a in ('ms','rel','frac','kernel','frac_of_visible_work','error')})
print(d['int8'].get('summary'))
PY
