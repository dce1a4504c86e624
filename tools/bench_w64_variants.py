#!/usr/bin/env python3
"""Time one shape with every variant library in tools/lab_bin (each in a child process): timing-only ablations."""
import os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
shape = sys.argv[1] if len(sys.argv) > 1 else "1,16,8192,128"
libs = [None] + sorted((ROOT / "tools" / "lab_bin").glob("libMFAFFI_*.so"))
for lib in libs:
    env = dict(os.environ)
    if lib:
        env["UMFA_LIBRARY"] = str(lib)
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "bench_one.py")] + shape.split(","), env=env, capture_output=True, text=True)
    print(f"{lib.name if lib else 'base':32s} {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]}")
