#!/bin/bash
# round 5, trip i: the int8 FLUX call, wave-per-block quantiser against the workgroup form, eager and graph-replayed, in one process
O=gpurun_out/r5i; mkdir -p $O
cp universal-metal-flash-attention_amd/lib/libMFAFFI.so /tmp/libMFAFFI_wg.so
for g in "" "--graph"; do
python tools/ab_inproc.py $g --rounds 12 --inner 20 --quant 2 --shape 1,24,4096,128 r4=tools/lab_bin/libMFAFFI_r4.so wave=intree wg=/tmp/libMFAFFI_wg.so:quant_block_wg=1 | tee -a $O/ab_int8.jsonl | cut -c1-600
done
python tools/ab_inproc.py --graph --rounds 12 --inner 20 --quant 3 --shape 1,24,4096,128 r4=tools/lab_bin/libMFAFFI_r4.so wave=intree wg=/tmp/libMFAFFI_wg.so:quant_block_wg=1 | tee -a $O/ab_int8.jsonl | cut -c1-600
python - <<'PY'
import sys
sys.path[:0]=['.','universal-metal-flash-attention_amd']
import torch, umfa_torch
q,k,v=(torch.randn(1,24,4096,128,device='cuda',dtype=torch.bfloat16) for _ in range(3))
out=torch.empty(1,24,4096,128,device='cuda',dtype=torch.float32); lse=torch.empty(1,24,4096,device='cuda',dtype=torch.float32)
def ev(fn,n=20):
    ts=[]
    for _ in range(n):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); return ts[len(ts)//2], ts[0]
for mode in ("blockwise","blockwise_fp8pv"):
    for opt in (0,1):
        with umfa_torch.options(quant_block_wg=opt):
            f=lambda: umfa_torch.quantized_attention_forward_stream(q,k,v,quant_mode=mode,out=out,lse=lse)
            for _ in range(5): f()
            print(mode,'quant_block_wg',opt,'median/min ms per call (events around ONE call)',ev(f))
PY
