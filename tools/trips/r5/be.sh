#!/bin/bash
# trip be: additive masks on the 128-row kernel at FLUX: is it the mask reads' access pattern (a lane per row) or the per-score arithmetic?
python3 - <<'PY' 2>&1 | grep -v amdgpu
import sys, torch
sys.path[:0]=['.','universal-metal-flash-attention_amd']
import umfa_torch
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g=torch.cuda.CUDAGraph(); s=torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); s.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
        g.replay(); s.synchronize(); ts=[]
        for r in range(5):
            a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            a.record(); g.replay(); g.replay(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b)/(2*n))
    return sorted(ts)[2]
B,H,S,D=1,24,4096,128
q,k,v=(torch.randn(B,H,S,D,device='cuda',dtype=torch.bfloat16) for _ in range(3))
o=torch.empty(B,H,S,D,device='cuda',dtype=torch.float32)
i=torch.arange(S,device='cuda')
bias=(-(i[:,None]-i[None,:]).abs().float()/256.0)
cases={'none (128-row kernel)':None,
       'fp16 [1,1,S,S]':bias.half()[None,None].contiguous(),
       'bf16 [1,1,S,S]':bias.bfloat16()[None,None].contiguous(),
       'fp32 [1,1,S,S]':bias[None,None].contiguous(),
       'fp16 [1,1,1,S] (one row for all)':bias[0].half()[None,None,None].contiguous(),
       'fp16 [1,H,S,S]':bias.half()[None,None].expand(1,H,S,S).contiguous(),
       'bool random [1,1,S,S] on the 128-row kernel':(torch.rand(S,S,device='cuda')<0.5)[None,None].contiguous()}
with umfa_torch.options(no_w64=1, no_w64_mask=1):
    for name,m in cases.items():
        t=timeit(lambda: umfa_torch.attention_forward(q,k,v,mask=m,out=o))
        print(f"{name}: {t:.4f} ms  {umfa_torch.last_kernel()}")
PY
