"""bench.py's N > 1 code path on ONE device (UMFA_BENCH_ONE_DEVICE=1: both ranks on cuda:0, process group gloo -- a code-path
rehearsal, never a measurement): the launcher spawns two ranks, the FLUX problem's heads are dealt over them, every rank's kernel
writes its heads into the final tensor, the in-place all-gathers complete it, and rank 0 checks the gathered tensor bit for bit
against local launches (`config.gathered_equals_local`).  No multi-GPU box exists in this pool; this is what can be run.

The file sorts first on purpose: the child processes must be started from a parent that has not initialised the GPU (an exec from a
GPU-initialised process is refused on this pool), so the test skips itself when an earlier test already did."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = [pytest.mark.gpu, pytest.mark.no_gpu_init]


def test_bench_gpus_2_rehearsal_gathers_the_full_output():
    torch = pytest.importorskip("torch")
    if torch.cuda.is_initialized():
        pytest.skip("the GPU is already initialised in this process: children cannot be exec'ed from here (run this file first / alone)")
    if torch.cuda.device_count() < 1:
        pytest.fail("GPU tests need a device")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["UMFA_BENCH_ONE_DEVICE"] = "1"
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--headline-only"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 only
    d = lines[0]
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and "rehearsal" in d
    assert d["config"]["gathered_equals_local"] is True
    assert d["config"]["ranks"] == 2 and "12 heads per rank" in d["config"]["workload"]
    assert d["value"] > 0 and d["roofline"]["flops_per_launch"] == pytest.approx(206158430208.0 / 2)
