"""Forward time of the strong-scaling shards of the FLUX problem: B1 H{24,12,6,3} S4096 D128 bf16 (N = 1, 2, 4, 8 ranks),
graph of 50 launches; both kernels."""
import sys, os
ROOT = os.path.join(os.path.dirname(__file__), "..", "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "universal-metal-flash-attention_amd")]
import torch
import umfa_torch
def graph_ms(fn, n=50):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): fn()
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(n): fn()
        g.replay(); side.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); b.synchronize()
    return a.elapsed_time(b) / n
base = None
for H in (24, 12, 6, 3):
    q, k, v = (torch.randn(1, H, 4096, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    out = torch.empty_like(q)
    t = graph_ms(lambda: umfa_torch.attention_forward(q, k, v, out=out))
    name = umfa_torch.last_kernel()
    base = base or t
    print(f"H{H}: {t*1e3:.1f} us  {name}  strong-scaling efficiency vs H24/N: {base / (24 // H) / t:.2f}", flush=True)
