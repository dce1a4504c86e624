import sys, time, torch
sys.path[:0]=['.','universal-metal-flash-attention_amd']
import umfa_torch
q,k,v=(torch.randn(1,2,128,64,device='cuda',dtype=torch.bfloat16) for _ in range(3))
o=torch.empty(1,2,128,64,device='cuda',dtype=torch.bfloat16)
for fn,name in ((lambda: umfa_torch.attention_forward(q,k,v,out=o),'ops.attention_forward(out=)'),(lambda: umfa_torch.attention_forward(q,k,v),'ops.attention_forward'),(lambda: umfa_torch.scaled_dot_product_attention(q,k,v),'umfa sdpa'),(lambda: torch.nn.functional.scaled_dot_product_attention(q,k,v),'torch sdpa')):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t=time.time()
    for _ in range(2000): fn()
    torch.cuda.synchronize(); print(name, round((time.time()-t)/2000*1e6,1),'us per call')
