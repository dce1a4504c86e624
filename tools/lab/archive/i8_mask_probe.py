#!/usr/bin/env python3
"""Lab: the runtime-quantised forward with the reference ABI's dense fp32 additive mask: time per call and the rate at which the mask comes in.
    python tools/lab/i8_mask_probe.py [H S]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd"), str(ROOT / "tools" / "lab")]
import torch  # noqa: E402
import umfa_torch  # noqa: E402
from split_probe import graph_us  # noqa: E402
H, S = (int(x) for x in sys.argv[1:3]) if len(sys.argv) >= 3 else (16, 8192)
D = 128
torch.manual_seed(0)
q, k, v = (torch.randn(1, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
i = torch.arange(S, device="cuda")
mb = ((i[:, None] // (S // 4)) == (i[None, :] // (S // 4)))
m32 = torch.zeros(1, H, S, S, device="cuda", dtype=torch.float32).masked_fill_(~mb[None, None], float("-inf"))
out = torch.empty(1, H, S, D, device="cuda", dtype=torch.float32)
lse = torch.empty(H * S, device="cuda", dtype=torch.float32)
for name, mask in (("unmasked", None), ("block-diagonal fp32 [1,H,S,S]", m32), ("all-zero fp32 [1,H,S,S]", torch.zeros_like(m32))):
    us = graph_us(lambda: umfa_torch.quantized_attention_forward_stream(q, k, v, mask=mask, out=out, lse=lse), n=6)
    print(f"H{H} S{S} {name}: {us:.1f} us  {umfa_torch.last_kernel()}" + (f"  mask {mask.numel() * 4 / us / 1e6:.2f} TB/s" if mask is not None else ""))
with umfa_torch.options(no_mask_flags=1):
    us = graph_us(lambda: umfa_torch.quantized_attention_forward_stream(q, k, v, mask=m32, out=out, lse=lse), n=4)
    print(f"H{H} S{S} block-diagonal, WITHOUT the tile-flag pre-pass (per-score mask reads, 16 bytes per load): {us:.1f} us  mask {m32.numel() * 4 / us / 1e6:.2f} TB/s")
o0 = umfa_torch.quantized_attention_forward_stream(q, k, v).clone()
o1 = umfa_torch.quantized_attention_forward_stream(q, k, v, mask=torch.zeros_like(m32))
print("zero mask vs no mask: max diff", float((o0 - o1).abs().max()), "of", float(o0.abs().max()))
