"""PCIe-inclusive cost of the synchronous entry (mfa_attention_forward on host-wrapping buffers): the chunked form (option
sync_chunks: head chunks on side streams, pinned ranges; 0 = by size, the default) against the one-upload form (sync_chunks = 1),
buffers wrapped once (bench.py's host_boundary leg), interleaved rounds; and the whole Python wrapper (umfa.flash_attention_forward:
wraps -- and, chunked, pins -- four arrays per call).  Output: one JSON line per shape."""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import bench  # noqa: E402
import umfa  # noqa: E402
import umfa_torch  # noqa: E402

SETTINGS = (1, 0, 4, 6, 8)
for shape in ((1, 24, 4096, 128), (1, 32, 8192, 128), (2, 16, 2048, 64), (8, 8, 1024, 128), (4, 16, 1024, 64)):
    ms = {c: [] for c in SETTINGS}
    for _ in range(3):  # interleaved rounds: the box's drift lands on every setting
        for c in SETTINGS:
            with umfa_torch.options(sync_chunks=c):
                ms[c].append(bench.bench_host_boundary(*shape, calls=5)["ms_per_call"])
    rng = np.random.default_rng(0)
    q, k, v = ((rng.standard_normal(shape, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16) for _ in range(3))
    wrap = {}
    with umfa.MFAContext() as ctx:
        for c in (1, 0, 1, 0):
            with umfa_torch.options(sync_chunks=c):
                ts = []
                for _ in range(5):
                    t0 = time.perf_counter()
                    umfa.flash_attention_forward(ctx, q, k, v, input_precision="bf16", intermediate_precision="bf16", layout="bhsd")
                    ts.append(time.perf_counter() - t0)
                wrap.setdefault(c, []).append(round(sorted(ts)[2] * 1e3, 3))
    # ... and mfa_attention_backward on host arrays (wrapper time: ten arrays wrapped per call)
    bw = {}
    if shape[2] <= 4096:
        do = q.copy()
        with umfa.MFAContext() as ctx:
            o, lse = umfa.flash_attention_forward(ctx, q, k, v, input_precision="bf16", intermediate_precision="bf16", layout="bhsd", return_lse=True)
            for c in (1, 0, 1, 0):
                with umfa_torch.options(sync_chunks=c):
                    ts = []
                    for _ in range(5):
                        t0 = time.perf_counter()
                        umfa.attention_backward(ctx, do, q, k, v, o, lse, input_precision="bf16", layout="bhsd")
                        ts.append(time.perf_counter() - t0)
                    bw.setdefault(c, []).append(round(sorted(ts)[2] * 1e3, 3))
    moved = 5 * q.nbytes
    print(json.dumps({"shape": shape, "MB_over_the_link": round(moved / 1e6, 1),
                      "ms_per_call_by_sync_chunks": {str(c): sorted(v)[1] for c, v in ms.items()}, "all": {str(c): v for c, v in ms.items()},
                      "python_wrapper_ms": {str(c): v for c, v in wrap.items()},
                      "python_wrapper_backward_ms": {str(c): v for c, v in bw.items()}}), flush=True)
