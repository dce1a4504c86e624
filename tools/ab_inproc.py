#!/usr/bin/env python3
"""In-process A/B of several builds of libMFAFFI.so (cdna_hip_programming.md §5.4 rule 24: interleaved rounds in ONE
process on ONE device; separate invocations add cross-process and cross-box variance).

  python tools/ab_inproc.py [--shape B,H,S,D] [--causal] [--dtype bf16|fp16] [--out same|fp32] [--rounds 12]
                            [--inner 20] [--parity] name=path [name=path ...]

Every library gets its own context (each .so carries its own process-wide singleton); a round launches `inner`
forwards of each library in turn between two events on the current stream.  Reports median / min per launch per
library and, with --parity, rel-err against the CPU oracle on a row subset (oracle.parity).
"""
import argparse
import ctypes
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import torch  # noqa: E402

from umfa import _ffi  # noqa: E402


def i64(v):
    return (ctypes.c_int64 * len(v))(*[int(x) for x in v])


class Lib:
    def __init__(self, name, path):
        # path[:option=value,...] -- the options go to THAT library's switch table (umfa_set_option; a library loaded from a copy of the
        # file has a table of its own)
        path, _, opts = path.partition(":")
        self.name = name
        self.lib = _ffi._lib if path == "intree" else _ffi._load_library(str(Path(path).resolve()))
        self.ctx = _ffi.mfa_context_t()
        _ffi._check_error(self.lib.mfa_create_context(ctypes.byref(self.ctx)))
        for kv in filter(None, opts.split(",")):
            k_, v_ = kv.split("=")
            v_ = v_.replace(";", ",")  # (a list value: its commas are written as semicolons here)
            self.lib.umfa_set_option.restype = ctypes.c_int
            rc = self.lib.umfa_set_option(self.ctx, k_.encode(), v_.encode())
            assert rc == 0, (name, kv, rc)

    mask = None  # class-wide: a bool mask tensor for every forward (--mask)

    def forward(self, q, k, v, out, causal, lse=None):
        B, H, Sq, D = q.shape
        prec = {torch.float16: 0, torch.bfloat16: 1, torch.float32: 2}
        m = Lib.mask
        margs = (None, None, None, 0, 0, 0) if m is None else (ctypes.c_void_p(m.data_ptr()), i64(m.shape), i64(m.stride()), m.dim(), 1 if m.dtype == torch.bool else 2,
                                                          {torch.bool: 0, torch.float16: 1, torch.bfloat16: 2, torch.float32: 3}[m.dtype])
        rc = self.lib.umfa_attention_forward_stream(
            self.ctx, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream),
            ctypes.c_void_p(q.data_ptr()), i64(q.stride()), ctypes.c_void_p(k.data_ptr()), i64(k.stride()),
            ctypes.c_void_p(v.data_ptr()), i64(v.stride()), ctypes.c_void_p(out.data_ptr()), prec[out.dtype],
            ctypes.c_void_p(lse.data_ptr()) if lse is not None else None, *margs,
            B, Sq, k.shape[2], H, D, float(D) ** -0.5, bool(causal), prec[q.dtype], prec[q.dtype])
        assert rc == 0, (self.name, rc)

    def qforward(self, q, k, v, out, causal, mode):
        B, H, Sq, D = q.shape
        prec = {torch.float16: 0, torch.bfloat16: 1, torch.float32: 2}
        rc = self.lib.umfa_quantized_forward_stream(
            self.ctx, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.c_void_p(q.data_ptr()),
            ctypes.c_void_p(k.data_ptr()), ctypes.c_void_p(v.data_ptr()), ctypes.c_void_p(out.data_ptr()), None, None,
            B, Sq, k.shape[2], H, D, float(D) ** -0.5, bool(causal), 3, mode, prec[q.dtype])
        assert rc == 0, (self.name, rc)

    def kernel(self):
        return self.lib.umfa_last_kernel_name(self.ctx).decode()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="1,24,4096,128")
    ap.add_argument("--causal", action="store_true")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--out", default="same")
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--inner", type=int, default=20)
    ap.add_argument("--parity", action="store_true")
    ap.add_argument("--mask", default="", help="blockdiag | padding | window_tensor | random | alltrue_keys | alltrue_2d: a bool mask tensor on every forward; bias | bias_per_head: an additive fp16 one; bias_bf16 | bias_f32 | blockdiag_bf16: additive bf16 / fp32 ones")
    ap.add_argument("--graph", action="store_true", help="time hipGraph replays of `inner` launches (short kernels: the Python launch path is not what is measured)")
    ap.add_argument("--quant", type=int, default=0, help="2 / 3: time umfa_quantized_forward_stream with that quant_mode (fp32 O)")
    ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    B, H, S, D = (int(x) for x in a.shape.split(","))
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[a.dtype]
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.float32).to(dt) for _ in range(3))
    out = torch.empty(B, H, S, D, device="cuda", dtype=dt if a.out == "same" else torch.float32)
    if a.mask:
        i_ = torch.arange(S, device="cuda")
        Lib.mask = {"blockdiag": lambda: ((i_[:, None] // 1024) == (i_[None, :] // 1024))[None, None].contiguous(),
                    "padding": lambda: (i_ < (3 * S) // 4)[None, None, None, :].contiguous(),
                    "window_tensor": lambda: ((i_[:, None] - i_[None, :]).abs() <= 512)[None, None].contiguous(),
                    "alltrue_keys": lambda: torch.ones(1, 1, 1, S, dtype=torch.bool, device="cuda"),
                    "alltrue_2d": lambda: torch.ones(1, 1, S, S, dtype=torch.bool, device="cuda"),
                    "bias": lambda: (-(i_[:, None] - i_[None, :]).abs().to(torch.float16) / 256.0)[None, None].contiguous(),  # additive fp16, every tile mixed
                    "bias_bf16": lambda: (-(i_[:, None] - i_[None, :]).abs().to(torch.float16) / 256.0).to(torch.bfloat16)[None, None].contiguous(),  # the same as a bf16 tensor (a bf16 model's mask)
                    "bias_f32": lambda: (-(i_[:, None] - i_[None, :]).abs().to(torch.float16) / 256.0).float()[None, None].contiguous(),  # ... widened to fp32 (fp16 holds it)
                    "blockdiag_bf16": lambda: torch.where((i_[:, None] // 1024) == (i_[None, :] // 1024), 0.0, float("-inf")).to(torch.bfloat16)[None, None].contiguous(),
                    "bias_per_head": lambda: (-(i_[:, None] - i_[None, :]).abs().float()[None] / (64.0 * (1 + torch.arange(H, device="cuda")[:, None, None]))).to(torch.float16)[None].contiguous(),
                    "random": lambda: torch.rand(1, H, S, S, device="cuda") > 0.5}[a.mask]()
    libs = [Lib(*s.split("=", 1)) for s in a.libs]
    if a.quant:
        out = torch.empty(B, H, S, D, device="cuda", dtype=torch.float32)
        for L in libs:
            L.forward = (lambda q_, k_, v_, o_, c_, L=L: L.qforward(q_, k_, v_, o_, c_, a.quant))
    for L in libs:
        for _ in range(5):
            L.forward(q, k, v, out, a.causal)
    torch.cuda.synchronize()
    times = {L.name: [] for L in libs}
    graphs = {}
    if a.graph:
        side = torch.cuda.Stream()
        for L in libs:
            with torch.cuda.stream(side):
                for _ in range(3):
                    L.forward(q, k, v, out, a.causal)
                side.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    for _ in range(a.inner):
                        L.forward(q, k, v, out, a.causal)
            graphs[L.name] = g
        torch.cuda.synchronize()
        for g in graphs.values():
            g.replay()
        torch.cuda.synchronize()
    for r in range(a.rounds):
        order = libs if r % 2 == 0 else libs[::-1]
        for L in order:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if a.graph:
                graphs[L.name].replay()
            else:
                for _ in range(a.inner):
                    L.forward(q, k, v, out, a.causal)
            e1.record()
            torch.cuda.synchronize()
            times[L.name].append(e0.elapsed_time(e1) / a.inner)
    flops = 4.0 * B * H * S * S * D * (0.5 if a.causal else 1.0)
    res = {"shape": a.shape, "causal": a.causal, "dtype": a.dtype, "out": a.out}
    for L in libs:
        t = sorted(times[L.name])
        med = t[len(t) // 2]
        res[L.name] = {"kernel": L.kernel(), "ms_median": round(med, 5), "ms_min": round(t[0], 5),
                       "tflops_median": round(flops / med / 1e9, 1)}
        if a.parity:
            from oracle import parity
            L.forward(q, k, v, out, a.causal)
            torch.cuda.synchronize()
            res[L.name]["parity"] = parity.forward_rel_err(q, k, v, out, causal=a.causal, floor_kind=a.dtype)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
