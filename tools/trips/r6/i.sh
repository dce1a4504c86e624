#!/bin/bash
# trip i: bool mask tensors on the one-wave-per-SIMD int8 kernel (MASKT instantiation of fa_fwd_w64_i8): parity + config 4 with the block-diagonal mask
O=gpurun_out/r6i; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_quantized.py -x -q 2>&1 | tail -6 | tee $O/tests.txt
python3 - <<'PY' 2>&1 | grep -v amdgpu | tee $O/cfg4_mask_timing.txt
import sys, json
sys.path[:0] = [".", "universal-metal-flash-attention_amd"]
import torch, umfa_torch
exec(open("tools/lab/bias_probe.py").read().split("shapes = ")[0])  # graph_ms
B, H, S, D = 1, 16, 8192, 128
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
out = torch.empty(B, H, S, D, device="cuda", dtype=torch.float32); lse = torch.empty(B * H * S, device="cuda", dtype=torch.float32)
i = torch.arange(S, device="cuda")
masks = {"blockdiag 4 x 2048 [1,1,S,S]": ((i[:, None] // 2048) == (i[None, :] // 2048))[None, None].contiguous(),
         "padding 3/4 [1,1,1,S]": (i < 6144)[None, None, None, :].contiguous(),
         "all true [1,1,S,S]": torch.ones(1, 1, S, S, dtype=torch.bool, device="cuda")}
t0 = graph_ms(lambda: umfa_torch.quantized_attention_forward_stream(q, k, v, out=out, lse=lse))
for name, m in masks.items():
    row = {"mask": name, "unmasked_ms": round(t0, 4)}
    for side, opts in (("w64_i8_mask", {}), ("fa_fwd_i8", {"no_w64_mask": 1})):
        with umfa_torch.options(**opts):
            row[side + "_ms"] = round(graph_ms(lambda: umfa_torch.quantized_attention_forward_stream(q, k, v, mask=m, out=out, lse=lse)), 4)
            row[side + "_kernel"] = umfa_torch.last_kernel()
    print(json.dumps(row), flush=True)
PY
