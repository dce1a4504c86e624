#!/usr/bin/env python3
"""Where does the guarded pair stop paying for an fp32 additive mask as the mask grows against the call's tensors?  The route's size rule (fa_fwd16_w64.hip
fwd_w64_supported: mask bytes <= f32_mask_ratio x (Q + K + V + O bytes), default 2) is lifted with the lab option and the pair timed against the 128-row kernel
alone, for masks fp16 holds (dense bias; 0 / -inf documents) and one it does not.  Graph-replayed, one process; JSON lines.
python tools/lab/f32_mask_ratio_probe.py [out.jsonl]"""
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
    # (B, H, S, D, mask batch, mask heads): blocks >= 256 CUs in every case
    cases = [(4, 4, 4096, 128, 4, 1), (8, 2, 4096, 128, 8, 1), (2, 8, 4096, 128, 2, 1), (4, 4, 4096, 64, 4, 1), (1, 24, 4096, 128, 1, 24), (1, 16, 4096, 128, 1, 16),
             (2, 8, 4096, 128, 2, 8), (1, 16, 8192, 128, 1, 16), (1, 24, 4096, 128, 1, 1)]
    for (B, H, S, D, Bm, Hm) in cases:
        torch.manual_seed(1)
        q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
        o = torch.empty(B, H, S, D, device="cuda", dtype=torch.float32)
        i = torch.arange(S, device="cuda")
        d = (i[:, None] - i[None, :]).abs().float()
        sl = (1.0 + torch.arange(Bm * Hm, device="cuda").float()).view(Bm, Hm, 1, 1)
        masks = {"dense bias fp16 holds": ((-d / 256.0).to(torch.float16).float()[None, None] * 1.0).expand(Bm, Hm, S, S).contiguous(),
                 "dense bias fp16 does not hold": (-d[None, None] / (256.0 * sl)).contiguous() if Bm * Hm > 1 else (-d / 255.0)[None, None].contiguous(),
                 "documents 0 / -inf": torch.where((i[:, None] // (S // 4)) == (i[None, :] // (S // 4)), 0.0, NEG)[None, None].expand(Bm, Hm, S, S).contiguous()}
        qkvo = B * H * D * (S * 6 + 2 * S * 2)
        for name, m in masks.items():
            with umfa_torch.options(f32_mask_ratio=1000):
                t_pair = graph_us(lambda: umfa_torch.attention_forward(q, k, v, mask=m, out=o))
                kern = umfa_torch.last_kernel()
            with umfa_torch.options(no_w64_f32_mask=1):
                t_128 = graph_us(lambda: umfa_torch.attention_forward(q, k, v, mask=m, out=o))
            rec = {"shape": f"B{B} H{H} S{S} D{D}", "mask_shape": [Bm, Hm, S, S], "mask": name, "mask_bytes_over_tensor_bytes": round(m.numel() * 4 / qkvo, 2),
                   "guarded_pair_us": round(t_pair, 1), "row128_alone_us": round(t_128, 1), "pair_taken": " | " in kern}
            print(json.dumps(rec), flush=True)
            if out:
                out.write(json.dumps(rec) + "\n")
            del m
        del masks


if __name__ == "__main__":
    main()
