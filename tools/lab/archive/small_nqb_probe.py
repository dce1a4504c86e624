#!/usr/bin/env python3
"""bf16 forward with few 256-row q-blocks per head (the fp16 image of V is then re-read only once or twice): the dispatcher's choice against the
128-row kernel (option no_w64: V converted in the kernel), graph-replayed us"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch  # noqa: E402

import umfa_torch  # noqa: E402
sys.path.insert(0, str(ROOT / "tools" / "lab"))
from split_probe import graph_us  # noqa: E402

DT = torch.float16 if "fp16" in sys.argv else torch.bfloat16
CASES2 = [(2, 32, 1024, 4096, 128, False), (4, 16, 1024, 1024, 128, False), (1, 64, 1024, 8192, 128, False), (2, 32, 1536, 1536, 128, False), (1, 32, 2048, 2048, 128, False),
          (8, 16, 1024, 1024, 128, True), (4, 32, 1280, 1280, 128, True), (8, 16, 1024, 1024, 64, False), (8, 16, 1024, 1024, 64, True), (4, 64, 1024, 1024, 64, False),
          (1, 24, 1024, 4096, 128, False), (2, 24, 1100, 1100, 128, False)]
CASES3 = [(B, H, S, S, D, c) for D in (128, 64) for c in (False, True)
          for (B, H, S) in [(1, 8, 2048), (1, 16, 2048), (1, 32, 2048), (2, 16, 2048), (1, 8, 4096), (1, 16, 4096), (1, 40, 2048), (4, 8, 2048), (1, 12, 3072), (1, 24, 2304), (2, 24, 2560)]]
CASES3 += [(1, 24, 4096, 512, 128, False), (1, 24, 4096, 77, 128, False), (2, 16, 2048, 8192, 128, False), (1, 24, 4096, 4352, 128, False), (2, 16, 2048, 8192, 64, False)]
for (B, H, Sq, Skv, D, causal) in CASES3 if len(sys.argv) > 1 and sys.argv[1] == "sweep" else CASES2 if len(sys.argv) > 1 and sys.argv[1] == "gate" else [(8, 32, 256, 8192, 128, False), (8, 32, 512, 4096, 128, False), (16, 16, 512, 512, 128, False), (4, 32, 768, 2048, 128, False),
                                   (2, 32, 1024, 1024, 128, False), (16, 16, 256, 256, 128, False), (8, 32, 256, 8192, 64, False), (16, 16, 512, 512, 64, False),
                                   (8, 16, 512, 512, 128, True), (4, 32, 768, 768, 128, True)]:
    torch.manual_seed(0)
    q = torch.randn(B, H, Sq, D, device="cuda", dtype=DT)
    k = torch.randn(B, H, Skv, D, device="cuda", dtype=DT)
    v = torch.randn(B, H, Skv, D, device="cuda", dtype=DT)
    o = torch.empty(B, H, Sq, D, device="cuda", dtype=torch.float32)
    row = {"shape": f"B{B} H{H} Sq{Sq} Skv{Skv} D{D} {'causal' if causal else 'full'}"}
    for name, opts in (("default", {}), ("r128", {"no_w64": 1})):
        with umfa_torch.options(**opts):
            row[name + "_us"] = graph_us(lambda: umfa_torch.attention_forward(q, k, v, causal=causal, out=o))
            row[name + "_kernel"] = umfa_torch.last_kernel()
    row["r128_over_default"] = round(row["r128_us"] / row["default_us"], 3)
    print(json.dumps(row), flush=True)
