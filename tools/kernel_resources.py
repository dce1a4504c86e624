#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS figures of one .hip source, from the device assembly (no GPU needed).

    python tools/kernel_resources.py fa_fwd_16_pv.hip [--filter substr] [--extra "-DFOO -fno-slp-vectorize"] [--src-dir DIR]

Prints one line per kernel: vgpr, agpr, sgpr, scratch bytes, spill count, LDS bytes, occupancy hint (waves / SIMD), MFMA count."""
import argparse
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "universal-metal-flash-attention_amd" / "csrc"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source")
    ap.add_argument("--filter", default="")
    ap.add_argument("--extra", default="")
    ap.add_argument("--src-dir", default=str(CSRC))
    ap.add_argument("--keep", default="", help="write the assembly here")
    ap.add_argument("--reuse", action="store_true", help="with --keep: parse the file if it exists instead of compiling")
    a = ap.parse_args()
    src_dir = Path(a.src_dir)
    extra = a.extra.split()
    mk = (CSRC / "Makefile").read_text()
    m = re.search(r"^(?:build/\S+\.o )*build/%s(?: build/\S+\.o)*: EXTRA \+= (.*)" % re.escape(Path(a.source).stem + ".o"), mk, re.M)  # (a rule may name several objects)
    if m:
        extra += m.group(1).split()
    with tempfile.TemporaryDirectory() as td:
        out = Path(a.keep) if a.keep else Path(td) / "k.s"
        if not (a.keep and a.reuse and out.exists()):
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "--cuda-device-only", "-S", *extra,
                                   str(src_dir / a.source), "-o", str(out)], cwd=src_dir)
        text = out.read_text()
    meta_all = text[text.index("amdhsa.kernels:"):]
    for entry in re.split(r"\n  - ", meta_all)[1:]:
        nm = re.search(r"\.name:\s+(\S+)", entry)
        if not nm:
            continue
        name = nm.group(1)
        if a.filter and a.filter not in name:
            continue

        def get(key, entry=entry):
            g = re.search(key + r":\s+(\d+)", entry)
            return int(g.group(1)) if g else -1
        k0 = text.index("\n" + name + ":")
        body = text[k0:text.index("s_endpgm", k0)]
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        print(f"vgpr {get(r'.vgpr_count'):4d} agpr {get(r'.agpr_count'):4d} sgpr {get(r'.sgpr_count'):4d} scratch {get(r'.private_segment_fixed_size'):5d} "
              f"spill {get(r'.vgpr_spill_count'):4d} lds {get(r'.group_segment_fixed_size'):6d} mfma {body.count('v_mfma'):4d} insts {len(body.splitlines()):6d}  {dem[:170]}")


if __name__ == "__main__":
    sys.exit(main())
