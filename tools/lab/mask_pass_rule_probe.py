#!/usr/bin/env python3
"""The size rule of the mask pre-passes (fa_aux.hip mask_flags_worthwhile: a float mask is read by a pre-pass -- tile flags for the 128-row kernel, classes / lists for the
bias kernels -- only when its bytes stay within 2 x the call's Q + K + V + O bytes) was set for dense per-head biases.  What does it cost masks WITHOUT a head dimension
(one per batch element: padding / document masks, usually sparse)?  For [B,1,S,S] masks at 1.6 ... 6.4 x the tensors: the rule at 2 against the rule lifted (lab option
mask_pass_ratio), fp16 and fp32 masks, dense bias and documents, on the default route and on the 128-row kernel.  Graph-replayed; JSON lines.
python tools/lab/mask_pass_rule_probe.py [out.jsonl]"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd"), str(ROOT / "tools")]
import torch
import umfa_torch
from bench_mask_f32 import graph_us

NEG = float("-inf")


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
    for (B, H, S, D) in [(2, 8, 4096, 128), (4, 4, 4096, 128), (8, 2, 4096, 128), (4, 4, 4096, 64), (8, 4, 2048, 128), (16, 2, 2048, 128)]:
        torch.manual_seed(1)
        q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
        o = torch.empty(B, H, S, D, device="cuda", dtype=torch.float32)
        i = torch.arange(S, device="cuda")
        d = (i[:, None] - i[None, :]).abs().float()
        lens = torch.tensor([S - 97 - (S // (2 * B)) * b for b in range(B)], device="cuda")
        base = {"dense bias": (-d / 256.0).to(torch.float16)[None, None].expand(B, 1, S, S).contiguous(),
                "documents 0 / -inf": torch.where((i[:, None] // (S // 4)) == (i[None, :] // (S // 4)), 0.0, NEG).to(torch.float16)[None, None].expand(B, 1, S, S).contiguous(),
                "key padding 0 / -inf, a length per batch element": torch.where(i[None, None, None, :] < lens[:, None, None, None], 0.0, NEG).to(torch.float16).expand(B, 1, S, S).contiguous()}
        qkvo = B * H * D * (S * 6 + 2 * S * 2)
        for name, m16 in base.items():
            for mdt in (torch.float16, torch.float32):
                m = m16.to(mdt)
                rec = {"shape": f"B{B} H{H} S{S} D{D}", "mask": f"{name} [B,1,S,S] {str(mdt).split('.')[1]}", "mask_bytes_over_tensor_bytes": round(m.numel() * m.element_size() / qkvo, 2)}
                for route, ropts in (("default_route", {}), ("row128", {"no_w64_bias": 1})):
                    for rule in (2, 8):
                        with umfa_torch.options(mask_pass_ratio=rule, f32_mask_ratio=rule, **ropts):
                            rec[f"{route}_rule{rule}_us"] = round(graph_us(lambda: umfa_torch.attention_forward(q, k, v, mask=m, out=o)), 1)
                            rec[f"{route}_rule{rule}_kernel"] = umfa_torch.last_kernel().split(" (")[0]
                print(json.dumps(rec), flush=True)
                if out:
                    out.write(json.dumps(rec) + "\n")
                del m
        del base


if __name__ == "__main__":
    main()
