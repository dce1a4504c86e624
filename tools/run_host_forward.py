#!/usr/bin/env python3
"""N calls of mfa_attention_forward on host-wrapping buffers at the FLUX shape (default options: head chunks on side streams), for rocprofv3."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "universal-metal-flash-attention_amd")]
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
t0 = time.perf_counter()
r = bench.bench_host_boundary(1, 24, 4096, 128, calls=n)
print({k: r[k] for k in ("ms_per_call", "tflops_pcie_inclusive", "host_link_gbps", "kernel", "calls")}, "wall s", round(time.perf_counter() - t0, 2))
