"""Per-launch durations and gaps of 20 back-to-back FLUX forwards right after a synchronize (what the driver's --steps 20 sees)."""
import sys, os, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "universal-metal-flash-attention_amd")]
import torch, umfa_torch
q, k, v = (torch.randn(1, 24, 4096, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3))
out = torch.empty_like(q)
fn = lambda: umfa_torch.attention_forward(q, k, v, out=out)
for _ in range(10): fn()
for rep in range(3):
    torch.cuda.synchronize()
    time.sleep(0.0 if rep < 2 else 0.05)
    n = 20
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(n):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    d = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n)]
    print(f"rep {rep}: wall {wall*1e3:.3f} ms, sum of event intervals {sum(d)/1e3:.3f} ms; per launch us:", [round(x) for x in d])
# graph of 20
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): fn()
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(20): fn()
    g.replay(); side.synchronize()
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); a.record(); g.replay(); b.record(); b.synchronize(); wall = time.perf_counter() - t0
        print(f"graph rep {rep}: wall {wall*1e3:.3f} ms, events {a.elapsed_time(b):.3f} ms")
    # back-to-back replays without sync in between
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); [g.replay() for _ in range(5)]; b.record(); b.synchronize()
    print(f"5 replays back to back: {a.elapsed_time(b)/5:.3f} ms per replay")
