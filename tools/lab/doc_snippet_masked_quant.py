import sys, ctypes, torch
sys.path[:0]=['.','universal-metal-flash-attention_amd']
import umfa_torch
from umfa_torch import ops
B,H,S,D=1,2,512,128
Sq=Skv=S
q,k,v=(torch.randn(B,H,S,D,device='cuda',dtype=torch.bfloat16) for _ in range(3))
doc_id=torch.arange(S,device='cuda')//128
m = (doc_id[:, None] == doc_id[None, :])[None, None]
o = umfa_torch.quantized_attention_forward_stream(q, k, v, mask=m)
out=torch.empty(B,H,S,D,device='cuda',dtype=torch.float32)
i64 = lambda t: (ctypes.c_int64 * len(t))(*t)
rc=ops._lib.umfa_quantized_forward_masked_stream(
    ops.context(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), None,
    m.data_ptr(), i64(m.shape), i64(m.stride()), m.dim(), 1, 0,
    B, Sq, Skv, H, D, D ** -0.5, False, 3, 2, 1)
torch.cuda.synchronize()
print(rc, torch.equal(o,out), umfa_torch.last_kernel())
