#!/bin/bash
# trip ad: exact tile count of a window's band -- window tests, fuzz legs with windows, A/B of the +-512 launch
O=gpurun_out/r5ad; mkdir -p $O
python3 -m pytest tests/test_gpu_w64.py tests/test_gpu_forward.py tests/test_gpu_fuzz.py tests/test_gpu_value_fuzz.py tests/test_gpu_routing.py -m gpu -x -q -k "window or fuzz or routing or shape" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
python3 - <<'PY' > $O/ab_window.txt 2>&1
import sys, torch
sys.path[:0]=['.','universal-metal-flash-attention_amd']
import umfa_torch
q,k,v=(torch.randn(1,24,4096,128,device='cuda',dtype=torch.bfloat16) for _ in range(3))
o=torch.empty(1,24,4096,128,device='cuda',dtype=torch.float32)
for w in ((512,512),(500,500),(1024,0),(256,256)):
    fn=lambda: umfa_torch.attention_forward(q,k,v,window=w,out=o)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g=torch.cuda.CUDAGraph(); s=torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); s.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(20): fn()
        g.replay(); s.synchronize()
        ts=[]
        for r in range(7):
            a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            a.record(); g.replay(); g.replay(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b)/40)
    print(w, umfa_torch.last_kernel(), 'ms median', sorted(ts)[3])
PY
cat $O/ab_window.txt | grep -v amdgpu
