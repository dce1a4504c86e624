#!/bin/bash
# trip af: causal + bool mask on the mask kernel -- tests, and the call against the 128-row kernel (FLUX size, causal + key padding / documents)
O=gpurun_out/r5af; mkdir -p $O
python3 -m pytest tests/test_gpu_w64_masks.py tests/test_gpu_forward.py tests/test_gpu_value_fuzz.py tests/test_gpu_fuzz.py tests/test_gpu_sdpa.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
python3 - <<'PY' 2>&1 | grep -v amdgpu | tee $O/causal_mask_timing.txt
import sys, torch
sys.path[:0]=['.','universal-metal-flash-attention_amd']
import umfa_torch
def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g=torch.cuda.CUDAGraph(); s=torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); s.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(20): fn()
        g.replay(); s.synchronize(); ts=[]
        for r in range(5):
            a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            a.record(); g.replay(); g.replay(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b)/40)
    return sorted(ts)[2]
for (B,H,S) in ((1,24,4096),(4,16,4096),(2,16,8192)):
    q,k,v=(torch.randn(B,H,S,128,device='cuda',dtype=torch.bfloat16) for _ in range(3))
    o=torch.empty(B,H,S,128,device='cuda',dtype=torch.float32)
    i=torch.arange(S,device='cuda')
    masks={'padding 3/4':(i<3*S//4)[None,None,None,:].expand(B,1,1,S).contiguous(), 'documents of S/4':((i[:,None]//(S//4))==(i[None,:]//(S//4)))[None,None].contiguous()}
    for name,m in masks.items():
        t1=timeit(lambda: umfa_torch.attention_forward(q,k,v,mask=m,causal=True,out=o)); k1=umfa_torch.last_kernel()
        with umfa_torch.options(no_w64_mask=1):
            t0=timeit(lambda: umfa_torch.attention_forward(q,k,v,mask=m,causal=True,out=o)); k0=umfa_torch.last_kernel()
        print(f"B{B} H{H} S{S} causal + {name}: {k1} {t1:.4f} ms | {k0} {t0:.4f} ms | ratio {t0/t1:.2f}")
PY
