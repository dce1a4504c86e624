#!/bin/bash
# trip ae: the backward above head_dim 256 (fa_bwd_wide) -- its tests, the backward / SDPA / library suites, timing of one call
O=gpurun_out/r5ae; mkdir -p $O
python3 -m pytest tests/test_gpu_wide_heads.py tests/test_gpu_backward.py tests/test_gpu_sdpa.py tests/test_gpu_library.py tests/test_gpu_compat_surfaces.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
python3 - <<'PY' 2>&1 | grep -v amdgpu
import sys, time, torch
sys.path[:0]=['.','universal-metal-flash-attention_amd']
import umfa_torch
for D in (320, 512, 1024):
    q,k,v,do=(torch.randn(1,8,2048,D,device='cuda',dtype=torch.bfloat16) for _ in range(4))
    o,lse=umfa_torch.attention_forward(q,k,v,out_dtype=torch.float32,return_lse=True)
    for _ in range(2): umfa_torch.attention_backward(do,q,k,v,o,lse,scale=D**-0.5)
    torch.cuda.synchronize(); t=time.time()
    for _ in range(3): umfa_torch.attention_backward(do,q,k,v,o,lse,scale=D**-0.5)
    torch.cuda.synchronize(); print('D',D,umfa_torch.last_kernel(),'backward ms',(time.time()-t)/3*1e3)
PY
