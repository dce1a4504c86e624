import sys, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'universal-metal-flash-attention_amd')
import umfa_torch
torch.manual_seed(0)
for dt in (torch.float16, torch.bfloat16):
    q,k,v = (torch.randn(1,2,128,128,device='cuda',dtype=dt) for _ in range(3))
    outs=[umfa_torch.attention_forward(q,k,v,out_dtype=torch.float32) for _ in range(4)]
    print(dt, 'fp32 deterministic:', all(torch.equal(outs[0],o) for o in outs))
    o16=[umfa_torch.attention_forward(q,k,v) for _ in range(4)]
    print(dt, '16 deterministic:', all(torch.equal(o16[0],o) for o in o16))
    ref=outs[0].to(dt)
    ne=(o16[0]!=ref)
    print(dt,'mismatch count',int(ne.sum()),'of',ne.numel(), 'max diff', float((o16[0].float()-outs[0]).abs().max()), float((ref.float()-outs[0]).abs().max()))
    idx=ne.nonzero()[:5]
    for i in idx:
        i=tuple(i.tolist()); print(i, float(outs[0][i]), float(o16[0][i]), float(ref[i]))
