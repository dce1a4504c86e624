#!/usr/bin/env python3
"""backward (head_dim 128 / 64, bf16): the launcher's kernel choices against the alternatives it has options for -- dQ kernel (bwd_dq = 1: two
workgroups per CU, 2: one per CU with the pinned pipeline), persistent dK / dV grid (bwd_persist); graph-replayed us"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd"), str(ROOT / "tools" / "lab")]
import torch  # noqa: E402

import umfa_torch  # noqa: E402
from split_probe import graph_us  # noqa: E402

for (B, H, S, D, causal) in [(1, 24, 4096, 128, False), (16, 16, 512, 128, False), (8, 16, 1024, 128, False), (2, 16, 2048, 128, False), (1, 8, 4096, 128, False), (4, 16, 1024, 128, True),
                             (1, 24, 4096, 128, True), (2, 16, 2048, 128, True), (1, 4, 8192, 128, False), (8, 8, 512, 128, True), (1, 2, 4096, 128, False), (32, 8, 256, 128, False)]:
    torch.manual_seed(0)
    q, k, v, do = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(4))
    o, lse = umfa_torch.attention_forward(q, k, v, causal=causal, return_lse=True)
    row = {"shape": f"B{B} H{H} S{S} D{D} {'causal' if causal else 'full'}"}
    for name, opts in (("default", {}), ("dq1", {"bwd_dq": 1}), ("dq2", {"bwd_dq": 2}), ("persist", {"bwd_persist": 1})):
        with umfa_torch.options(**opts):
            row[name + "_us"] = graph_us(lambda: umfa_torch.attention_backward(do, q, k, v, o, lse, scale=D ** -0.5, causal=causal), n=20)
    best = min(row[k_] for k_ in row if k_.endswith("_us"))
    row["default_over_best"] = round(row["default_us"] / best, 3)
    print(json.dumps(row), flush=True)
