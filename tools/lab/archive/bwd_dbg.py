import sys, os
ROOT = os.path.join(os.path.dirname(__file__), "..", "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "universal-metal-flash-attention_amd")]
import numpy as np
import umfa
from oracle import oracle as orc
ctx = umfa.MFAContext() if hasattr(umfa, "MFAContext") else umfa.create_context()
for shape, causal in [((1, 2, 128, 128), True), ((1, 2, 256, 128), True), ((1, 1, 1024, 128), True), ((1, 2, 130, 128), True), ((1, 2, 128, 128), False)]:
    rng = np.random.default_rng(9)
    f = [rng.standard_normal(shape).astype(np.float32) for _ in range(4)]
    q, k, v, do = (orc.f32_to_bf16_bits(a).reshape(shape) for a in f)
    o, lse = orc.sdpa_forward(q, k, v, causal=causal, return_lse=True)
    rdq, rdk, rdv, rd = orc.sdpa_backward(do, q, k, v, o, lse, causal=causal)
    dq, dk, dv, dvec = umfa.attention_backward(ctx, do, q, k, v, o, lse.ravel(), causal=causal, input_precision="bf16")
    for got, ref, name in [(dq, rdq, "dq"), (dk, rdk, "dk"), (dv, rdv, "dv")]:
        err = np.abs(got - ref)
        rel = err.max() / np.abs(ref).max()
        bad = np.argwhere(err > 0.02 * np.abs(ref).max())
        rows = sorted(set(bad[:, 2].tolist())) if len(bad) else []
        cols = sorted(set(bad[:, 3].tolist())) if len(bad) else []
        print(shape, causal, name, f"rel {rel:.4f} finite {np.isfinite(got).all()} bad rows {rows[:20]} ({len(rows)}) bad cols {cols[:8]}..{cols[-3:] if cols else ''} ({len(cols)})", flush=True)
