import sys, os, subprocess
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
code = r'''
import sys, os
sys.path[:0] = [%r, os.path.join(%r, "universal-metal-flash-attention_amd")]
import torch, umfa_torch
H = int(sys.argv[1])
q, k, v = (torch.randn(1, H, 4096, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3))
out = torch.empty_like(q)
fn = lambda: umfa_torch.attention_forward(q, k, v, out=out)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): fn()
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(50): fn()
    g.replay(); side.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); b.synchronize()
print("H%%d grid %%s: %%.1f us %%s" %% (H, os.environ.get("UMFA_W64_GRID", "default"), a.elapsed_time(b) / 50 * 1e3, umfa_torch.last_kernel()))
''' % (ROOT, ROOT)
for H, grids in ((3, ["", "240", "192", "144", "96", "48"]), (6, ["", "192", "96"]), (12, ["", "192"])):
    for g in grids:
        env = dict(os.environ)
        if g: env["UMFA_W64_GRID"] = g
        env["UMFA_FORCE_W64"] = "1"
        r = subprocess.run([sys.executable, "-c", code, str(H)], env=env, capture_output=True, text=True)
        print(r.stdout.strip() or r.stderr[-300:], flush=True)
