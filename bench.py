#!/usr/bin/env python3
"""bench.py -- SDPA forward throughput on MI355X (BASELINE.json metric).

Workload (config.workload): the FLUX shape B=1 H=24 S=4096 D=128, bf16, forward, per GPU.
A "step" is one forward pass of that shape through the in-stream C-ABI entry
(umfa_attention_forward_stream -> fa_fwd16<bf16,128>), inputs resident in HBM.
With N GPUs each rank runs its own batch element (batch x head pairs shard with no data-path
exchange) -> weak scaling; the all-gather of O over RCCL that a caller wanting the
full output everywhere would add is timed separately and reported as `with_allgather`.

One JSON line on rank 0:
  value      whole-job TFLOP/s (4*B*H*Sq*Skv*D FLOPs per forward, all ranks) from the barrier-bracketed
             wall time of exactly K steps (max over ranks)
  roofline   bf16 MFMA bound: algorithmic FLOPs per launch / mean kernel duration from HIP events on
             the launch stream, vs the 2.5 PFLOP/s dense peak (MI355X_MICROARCH.md)
  cpu_baseline  the CPU oracle (oracle/sdpa_ref.c, OpenMP) on a bounded sample of the same workload, plus
             torch_cpu_sdpa: PyTorch's own CPU SDPA on the full FLUX shape (fp32 and bf16)
  int8       runtime-quantised (block-wise int8) forward of the same shape vs bf16 (when built)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for p in (str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_BF16_TFLOPS = 2500.0  # dense, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
B, H, S, D = 1, 24, 4096, 128
FLOPS_PER_STEP = 4.0 * B * H * S * S * D  # 206.16 GFLOP


def cpu_baseline(cores: int):
    """Oracle timed on this host: `cores` heads of the FLUX shape in parallel (one OpenMP task per head)."""
    import numpy as np

    from oracle import oracle
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    heads = max(1, cores)
    s_len = 2048  # bounded sample: S=2048 keeps the run at ~10-20 s of CPU work
    rng = np.random.default_rng(0)
    mk = lambda: oracle.f32_to_bf16_bits(rng.standard_normal((1, heads, s_len, D)).astype(np.float32))  # noqa: E731
    q, k, v = mk(), mk(), mk()
    oracle.lib()
    t0 = time.perf_counter()
    oracle.sdpa_forward(q, k, v)
    dt = time.perf_counter() - t0
    flops = 4.0 * heads * s_len * s_len * D
    res = {"value": round(flops / dt / 1e12, 5), "unit": "TFLOP/s", "cores": cores, "kind": "port",
           "sample": f"B=1 H={heads} S={s_len} D={D} bf16 forward, oracle/sdpa_ref.c (fp64 accumulate, OpenMP), {dt:.1f} s"}
    try:  # the north-star's other CPU number: PyTorch's own CPU SDPA (the reference's config-1 path) on the FLUX shape
        res["torch_cpu_sdpa"] = torch_cpu_sdpa()
    except Exception as exc:
        res["torch_cpu_sdpa"] = {"error": repr(exc)}
    return res


def torch_cpu_sdpa():
    """torch.nn.functional.scaled_dot_product_attention on the host cores, full FLUX shape, fp32 and bf16 (best of <= 3
    runs each, at most ~20 s in all); reported beside the oracle port, never a target."""
    import torch
    import torch.nn.functional as F
    out = {"cores": torch.get_num_threads(), "unit": "TFLOP/s", "sample": f"B={B} H={H} S={S} D={D}, best of <= 3"}
    budget = time.perf_counter() + 20.0
    for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        g = torch.Generator().manual_seed(0)
        q, k, v = (torch.randn(B, H, S, D, generator=g).to(dt) for _ in range(3))
        best = None
        for _ in range(3):
            if time.perf_counter() > budget and best is not None:
                break
            t0 = time.perf_counter()
            F.scaled_dot_product_attention(q, k, v)
            el = time.perf_counter() - t0
            best = el if best is None else min(best, el)
            if time.perf_counter() > budget:
                break
        out[name] = round(FLOPS_PER_STEP / best / 1e12, 4)
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--causal", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch the K timed steps eagerly instead of as one hipGraph")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import umfa_torch

    torch.manual_seed(rank)
    q, k, v = (torch.randn(B, H, S, D, device=dev, dtype=torch.float32).to(torch.bfloat16) for _ in range(3))
    out = torch.empty(B, H, S, D, device=dev, dtype=torch.bfloat16)

    def step():
        umfa_torch.attention_forward(q, k, v, causal=args.causal, out=out)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # The K timed steps are captured once into a hipGraph (launch-bound loop: ~20 us of host gap per eager
    # launch vs 226 us of kernel); the graph holds exactly K launches of the forward and nothing else.
    graph = None
    if not args.no_graph:
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                for _ in range(args.steps):
                    step()
        torch.cuda.current_stream(dev).wait_stream(side)
        graph.replay()  # untimed: the first replay pays the graph upload (~1 ms); K more warm-up steps
    barrier()
    t0 = time.perf_counter()
    if graph is not None:
        graph.replay()
    else:
        for _ in range(args.steps):
            step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kernel_name = umfa_torch.last_kernel()

    # per-launch kernel duration from HIP events on the launch stream (torch's current stream)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for a, b in evs:
        a.record()
        step()
        b.record()
    torch.cuda.synchronize()
    durs = sorted(a.elapsed_time(b) for a, b in evs)
    mean_ms = sum(durs) / len(durs)
    flops = FLOPS_PER_STEP * (0.5 if args.causal else 1.0)
    achieved = flops / (mean_ms * 1e-3) / 1e12

    # optional: the O all-gather a caller would add to hold the full output on every rank
    gather = None
    if world > 1:
        full = torch.empty(world * B, H, S, D, device=dev, dtype=torch.bfloat16)
        for _ in range(3):
            step()
            dist.all_gather_into_tensor(full, out)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
            dist.all_gather_into_tensor(full, out)
        barrier()
        dtg = time.perf_counter() - t1
        t = torch.tensor([dtg], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dtg = float(t.item())
        gather = {"value": round(flops * world * args.steps / dtg / 1e12, 2), "unit": "TFLOP/s",
                  "ms_per_step": round(dtg / args.steps * 1e3, 4), "bytes_per_rank": out.numel() * 2}

    int8 = None
    if world == 1:
        try:
            int8 = umfa_torch.bench_int8(min(args.steps, 20), 3)
        except Exception as exc:  # reported, never silently replaced by another path
            int8 = {"error": repr(exc)}

    if rank == 0:
        traffic = None
        tfile = ROOT / "profiles" / "traffic_latest.json"
        if tfile.exists():
            try:
                traffic = json.loads(tfile.read_text()).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "SDPA fwd TFLOPS (bf16) + int8 speedup, B=1 H=24 S=4096 D=128, 1/2/4/8 GPU",
            "value": round(flops * world * args.steps / dt / 1e12, 2),
            "unit": "TFLOP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic N(0,1) Q/K/V, torch.manual_seed(rank)",
            "config": {"workload": f"FLUX-shape SDPA forward B={B} H={H} S={S} D={D} bf16{' causal' if args.causal else ''}, "
                                   "one batch element per GPU, bf16 O (fused cast-back epilogue)",
                       "kernel": kernel_name, "entry": "umfa_attention_forward_stream (in-stream C ABI)",
                       "launch": "eager" if args.no_graph else
                                 f"one hipGraph of {args.steps} launches: replayed once untimed (extra warm-up), once timed",
                       "parallelism": f"batch-x-head shards, {world} rank(s), no data-path collective"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                         "kernel_ms_mean": round(mean_ms, 5), "kernel_ms_min": round(durs[0], 5),
                         "flops_per_launch": flops},
        }
        if gather:
            line["with_allgather"] = gather
        if int8:
            line["int8"] = int8
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(os.cpu_count() or 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
