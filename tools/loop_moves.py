#!/usr/bin/env python3
"""Per kernel of a gfx950 assembly file: scratch instructions and compiler-generated accumulator-register moves by loop depth
(depth 2 = the tile loop of the one-wave-per-SIMD kernels).  A change that is nowhere near the loop can still move loop-carried
values into AGPRs (the register allocation of a 512-register kernel is global: profiles/r4/lab_notes.md §1).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -w --cuda-device-only -S fa_fwd16_w64.hip -o w64.s
    python tools/loop_moves.py w64.s [kernel-substring]"""
import re
import sys

text = open(sys.argv[1]).read()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
kern, cur = {}, None
for line in text.split("\n"):
    m = re.match(r"^(_ZN4umfa\w+):", line)
    if m:
        cur = []
        kern[m.group(1)] = cur
    elif cur is not None:
        cur.append(line)
        if "s_endpgm" in line:
            cur = None
for name, lines in kern.items():
    if sub not in name:
        continue
    depth, in_asm, scratch, moves = 0, False, {}, {}
    for ln in lines:
        if re.match(r"^\.LBB\d+_\d+:", ln):
            m = re.search(r"Depth=(\d+)", ln)
            depth = int(m.group(1)) if m else 0
        if "ASMSTART" in ln:
            in_asm = True
        elif "ASMEND" in ln:
            in_asm = False
        elif not in_asm:
            if "scratch_" in ln:
                scratch[depth] = scratch.get(depth, 0) + 1
            if "v_accvgpr" in ln:
                moves[depth] = moves.get(depth, 0) + 1
    print(name, "scratch by depth", dict(sorted(scratch.items())), "compiler accvgpr moves by depth", dict(sorted(moves.items())))
