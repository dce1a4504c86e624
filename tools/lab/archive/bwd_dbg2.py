import sys, os
ROOT = os.path.join(os.path.dirname(__file__), "..", "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "universal-metal-flash-attention_amd")]
import numpy as np
import umfa
from oracle import oracle as orc
ctx = umfa.MFAContext()
shape, causal = (1, 1, 128, 128), True
rng = np.random.default_rng(9)
f = [rng.standard_normal(shape).astype(np.float32) for _ in range(4)]
q, k, v, do = (orc.f32_to_bf16_bits(a).reshape(shape) for a in f)
o, lse = orc.sdpa_forward(q, k, v, causal=causal, return_lse=True)
rdq, rdk, rdv, rd = orc.sdpa_backward(do, q, k, v, o, lse, causal=causal)
dq, dk, dv, dvec = umfa.attention_backward(ctx, do, q, k, v, o, lse.ravel(), causal=causal, input_precision="bf16")
# non-causal reference of the same inputs: does dq look like the UNMASKED gradient?
o2, lse2 = orc.sdpa_forward(q, k, v, causal=False, return_lse=True)
for r in (0, 1, 31, 32, 63, 64, 95, 96, 127):
    e = np.abs(dq[0, 0, r] - rdq[0, 0, r]).max() / max(np.abs(rdq[0, 0, r]).max(), 1e-9)
    print("row", r, "rel", round(float(e), 4), "got", dq[0, 0, r, :3], "ref", rdq[0, 0, r, :3])
