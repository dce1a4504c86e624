#!/usr/bin/env python3
"""fp32 additive masks: the guarded pair (classification pass + fp16 copy + bias kernel | 128-row kernel, the device picks) against the 128-row kernel
alone (option no_w64_f32_mask = 1) and against the same values as an fp16 tensor.  Graph-replayed, one process; JSON lines.
python tools/bench_mask_f32.py [out.jsonl]"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch
import umfa_torch

NEG = float("-inf")


def graph_us(fn, n=20, reps=3):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
        g.replay()
        torch.cuda.synchronize()
        best = 1e30
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s)
            for _ in range(reps):
                g.replay()
            b.record(s)
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) * 1e3 / (n * reps))
    return best


def cases(B, H, S, D, dt):
    i = torch.arange(S, device="cuda")[:, None]
    j = torch.arange(S, device="cuda")[None, :]
    d = (i - j).abs().float()
    yield "rel-pos bias -|i-j|/256 held by fp16 [1,1,S,S]", (-d / 256.0).to(torch.float16).float()[None, None].contiguous()
    yield "rel-pos bias -|i-j|/256 in fp32 (not held) [1,1,S,S]", (-d / 256.0)[None, None].contiguous()
    yield "block-diagonal 0 / -inf, 4 documents [1,1,S,S]", torch.where((i // (S // 4)) == (j // (S // 4)), 0.0, NEG)[None, None].contiguous()
    yield "window +-512 as 0 / -inf [1,1,S,S]", torch.where((i - j).abs() <= 512, 0.0, NEG)[None, None].contiguous()
    yield "key padding 0 / -inf [B,1,1,S]", torch.where(j < (S * 3) // 4, 0.0, NEG)[None, None].expand(B, 1, 1, S).contiguous()
    yield "all zero [1,1,S,S]", torch.zeros(1, 1, S, S, device="cuda")


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
    for (B, H, S, D, dt) in [(1, 24, 4096, 128, torch.bfloat16), (1, 24, 4096, 128, torch.float16), (2, 16, 4096, 64, torch.bfloat16), (4, 16, 2048, 128, torch.bfloat16),
                             (1, 16, 8192, 128, torch.bfloat16)]:
        torch.manual_seed(1)
        q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=dt) for _ in range(3))
        o = torch.empty(B, H, S, D, device="cuda", dtype=torch.float32)
        for name, m in cases(B, H, S, D, dt):
            t_pair = graph_us(lambda: umfa_torch.attention_forward(q, k, v, mask=m, out=o))
            kern = umfa_torch.last_kernel()
            with umfa_torch.options(no_w64_f32_mask=1):
                t_128 = graph_us(lambda: umfa_torch.attention_forward(q, k, v, mask=m, out=o))
                k128 = umfa_torch.last_kernel()
            m16 = m.to(torch.float16)
            t_16 = graph_us(lambda: umfa_torch.attention_forward(q, k, v, mask=m16, out=o))
            rec = {"shape": f"B{B} H{H} S{S} D{D} {str(dt).split('.')[1]}", "mask": name, "guarded_pair_us": round(t_pair, 1), "row128_alone_us": round(t_128, 1),
                   "same_values_as_fp16_tensor_us": round(t_16, 1), "kernels": kern, "kernel_128": k128}
            print(json.dumps(rec), flush=True)
            if out:
                out.write(json.dumps(rec) + "\n")
        t0 = graph_us(lambda: umfa_torch.attention_forward(q, k, v, out=o))
        rec = {"shape": f"B{B} H{H} S{S} D{D} {str(dt).split('.')[1]}", "mask": "none", "us": round(t0, 1), "kernels": umfa_torch.last_kernel()}
        print(json.dumps(rec), flush=True)
        if out:
            out.write(json.dumps(rec) + "\n")


if __name__ == "__main__":
    main()
