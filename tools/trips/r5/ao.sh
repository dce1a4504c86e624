#!/bin/bash
# trip ao: fa_fwd_i8 (the 128-row int8 kernel) against fa_fwd_w64_i8 at config 4, unmasked; and with key padding
python3 - <<'PY' 2>&1 | grep -v amdgpu
import sys, torch
sys.path[:0]=['.','universal-metal-flash-attention_amd']
import umfa_torch
def timeit(fn, n=6):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    g=torch.cuda.CUDAGraph(); s=torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); s.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
        g.replay(); s.synchronize(); ts=[]
        for r in range(5):
            a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            a.record(); g.replay(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b)/n)
    return sorted(ts)[2]
B,H,S,D=1,16,8192,128
q,k,v=(torch.randn(B,H,S,D,device='cuda',dtype=torch.bfloat16) for _ in range(3))
o=torch.empty(B,H,S,D,device='cuda',dtype=torch.float32); l=torch.empty(B*H*S,device='cuda',dtype=torch.float32)
i=torch.arange(S,device='cuda')
print('w64 unmasked', round(timeit(lambda: umfa_torch.quantized_attention_forward_stream(q,k,v,out=o,lse=l)),4), umfa_torch.last_kernel())
with umfa_torch.options(no_w64=1):
    print('128-row unmasked', round(timeit(lambda: umfa_torch.quantized_attention_forward_stream(q,k,v,out=o,lse=l)),4), umfa_torch.last_kernel())
for frac in (1.0, 0.73, 0.5):
    m=(i<int(S*frac))[None,None,None,:].contiguous()
    print('padding', frac, round(timeit(lambda: umfa_torch.quantized_attention_forward_stream(q,k,v,mask=m,out=o,lse=l)),4), umfa_torch.last_kernel())
m=torch.ones(1,1,S,S,dtype=torch.bool,device='cuda')
print('all-true [1,1,S,S]', round(timeit(lambda: umfa_torch.quantized_attention_forward_stream(q,k,v,mask=m,out=o,lse=l)),4), umfa_torch.last_kernel())
PY
python3 -m pytest tests/test_gpu_quantized.py tests/test_gpu_value_fuzz.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -2
