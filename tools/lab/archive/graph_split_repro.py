#!/usr/bin/env python3
"""is the 'zero the outputs, replay, find zeros' effect in the library, or between an eager write and ANY graph replay?"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch
import umfa_torch

g = torch.Generator(device="cuda").manual_seed(0)
q, k, v, do = (torch.randn(1, 2, 512, 128, device="cuda", dtype=torch.bfloat16, generator=g) for _ in range(4))
a = torch.randn(1, 2, 512, 128, device="cuda", generator=g)


def test(name, fn, how, nrep=6, replay_stream=None):
    eager = [t.clone() for t in fn()]
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn()
        side.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            cap = fn()
    torch.cuda.synchronize()
    res = []
    for rep in range(nrep):
        for t in cap:
            if how == "zero":
                t.zero_()
            elif how == "fill7":
                t.fill_(7.0)
            elif how == "memset":
                torch.cuda.current_stream().synchronize()
                t.copy_(torch.zeros_like(t))
        if how.endswith("sync"):
            for t in cap:
                t.zero_()
            torch.cuda.synchronize()
        if replay_stream is not None:
            replay_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(replay_stream):
                gr.replay()
        else:
            gr.replay()
        torch.cuda.synchronize()
        res.append("".join("T" if torch.equal(x, y) else "F" for x, y in zip(eager, cap)))
    print(name, how, "replay on", "side stream" if replay_stream is not None else "current stream", res, flush=True)


def pure_torch():
    return (a * 2.0 + 1.0, torch.matmul(a, a.transpose(-1, -2)))


def lib_fwd_torch():
    o, lse = umfa_torch.attention_forward(q, k, v, out_dtype=torch.float32, return_lse=True)
    return o, lse, torch.matmul(q.float(), k.float().transpose(-1, -2))


test("pure_torch", pure_torch, "zero")
test("pure_torch", pure_torch, "fill7")
test("lib_fwd_torch", lib_fwd_torch, "zero")
test("lib_fwd_torch", lib_fwd_torch, "fill7")
test("lib_fwd_torch", lib_fwd_torch, "zero_then_sync")
test("lib_fwd_torch", lib_fwd_torch, "zero", replay_stream=torch.cuda.Stream())
with umfa_torch.options(no_split=1):
    test("lib_fwd_torch no_split", lib_fwd_torch, "zero")
with umfa_torch.options(force_w64=1):
    test("lib_fwd_torch w64", lib_fwd_torch, "zero")
