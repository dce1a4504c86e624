#!/usr/bin/env python3
"""few 256-row blocks with long key ranges (the strong-scaling shards of FLUX: 1 ... 6 heads): one-workgroup-per-CU kernel (forced) against the
128-row kernel with its split-KV plan, bf16 default options"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd"), str(ROOT / "tools" / "lab")]
import torch  # noqa: E402

import umfa_torch  # noqa: E402
from split_probe import graph_us  # noqa: E402

CAUSAL = len(sys.argv) > 1 and sys.argv[1] == "causal"
CASES_C = [(1, 32, 2048, 128), (1, 16, 4096, 128), (2, 16, 2048, 128), (1, 40, 2048, 128), (1, 24, 4096, 128), (4, 8, 2048, 128), (1, 8, 8192, 128), (1, 20, 4096, 128), (1, 12, 4096, 128),
           (1, 32, 2048, 64), (1, 16, 4096, 64), (1, 24, 4096, 64), (2, 24, 2048, 64), (1, 8, 8192, 64)]
for (B, H, S, D) in CASES_C if CAUSAL else [(1, 1, 4096, 128), (1, 2, 4096, 128), (1, 3, 4096, 128), (1, 4, 4096, 128), (1, 6, 4096, 128), (1, 2, 8192, 128), (1, 4, 2048, 128), (1, 8, 2048, 128), (1, 12, 2048, 128),
                     (1, 1, 16384, 128), (1, 4, 4096, 64), (1, 8, 4096, 64)]:
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    o = torch.empty(B, H, S, D, device="cuda", dtype=torch.float32)
    row = {"shape": f"B{B} H{H} S{S} D{D}", "steps_per_cu": H * B * (S // 256) * (S // 64) / 256}
    for name, opts in (("default", {}), ("w64", {"force_w64": 1}), ("r128", {"no_w64": 1})):
        with umfa_torch.options(**opts):
            row[name + "_us"] = graph_us(lambda: umfa_torch.attention_forward(q, k, v, causal=CAUSAL, out=o))
            row[name + "_kernel"] = umfa_torch.last_kernel()
    print(json.dumps(row), flush=True)
