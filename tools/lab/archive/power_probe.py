"""Is the forward power-limited?  Same instruction stream, operands of different switching activity:
N(0,1) data vs all-zero vs constant operands.  HIP events, interleaved rounds."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "universal-metal-flash-attention_amd"))
import torch
import umfa_torch
from umfa_torch import ops

def med(f, n=40):
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]

for (B, H, S) in [(1, 24, 4096), (1, 16, 8192)]:
    D = 128
    sets = {}
    torch.manual_seed(0)
    sets["randn"] = [torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3)]
    sets["zeros"] = [torch.zeros(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3)]
    sets["ones"] = [torch.ones(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3)]
    sets["randn*0.01"] = [t * 0.01 for t in sets["randn"]]
    sets["q,k zeros v randn"] = [sets["zeros"][0], sets["zeros"][1], sets["randn"][2]]
    sets["q,k randn v zeros"] = [sets["randn"][0], sets["randn"][1], sets["zeros"][2]]
    out = torch.empty(B, H, S, D, device="cuda", dtype=torch.bfloat16)
    fns = {k: (lambda t=t: ops.attention_forward(t[0], t[1], t[2], out=out)) for k, t in sets.items()}
    for f in fns.values():
        for _ in range(5): f()
    res = {k: [] for k in fns}
    for _ in range(3):
        for k, f in fns.items():
            res[k].append(med(f))
    print(f"B{B} H{H} S{S}: " + "  ".join(f"{k}: {min(v):.4f}" for k, v in res.items()), umfa_torch.last_kernel(), flush=True)
