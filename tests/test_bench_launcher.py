"""bench.py --gpus N starts N ranks itself (`--gpus` was parsed and ignored).  On the CPU the same launcher
is driven with --launcher-selftest: the ranks rendezvous over gloo, cover the 24 FLUX heads between them, and rank 0
prints one JSON line; a failing rank makes the run exit non-zero."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("n", [2, 4])
def test_gpus_flag_spawns_n_ranks(n):
    r = _run(["--gpus", str(n), "--launcher-selftest"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 only
    d = lines[0]
    assert d["n_gpus"] == n and d["sum_of_rank_ids_plus_one"] == n * (n + 1) / 2  # every rank joined the all-reduce
    assert d["heads_covered"] == 24 and d["local_rank_env"] == "0"


def test_runs_as_a_rank_under_an_external_launcher():
    """WORLD_SIZE already set (torch.distributed.run): bench.py must not spawn again"""
    r = _run(["--gpus", "1", "--launcher-selftest"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_a_failing_rank_fails_the_run():
    """no GPU here: every rank of the real bench exits non-zero ("needs a GPU"), and so does the launcher"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("this check needs a box without a GPU")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "needs a GPU" in r.stderr
