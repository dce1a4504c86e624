#!/usr/bin/env python3
"""List, for one kernel of an assembly file, the waits / LDS reads / VMEM ops of the tile loop with the running MFMA count
(which gap they sit in): python tools/trace_waits.py file.s kernel-substring [max lines]"""
import re, sys
text = open(sys.argv[1]).read().split("\n")
sub = sys.argv[2]
out, cur = [], None
for l in text:
    m = re.match(r"^(_ZN4umfa\w+):", l)
    if m: cur = m.group(1) if sub in m.group(1) else None
    elif cur:
        out.append(l)
        if "s_endpgm" in l: break
depth, inasm, nm, rows = 0, False, 0, []
for l in out:
    if re.match(r"^\.LBB\d+_\d+:", l):
        m = re.search(r"Depth=(\d+)", l); depth = int(m.group(1)) if m else 0
        rows.append(f"--- {l.split(':')[0]} depth {depth}")
        continue
    s = l.strip()
    if "ASMSTART" in s: inasm = True; continue
    if "ASMEND" in s: inasm = False; continue
    if not s or s.startswith(";") or depth < 2: continue
    t = s.split()[0]
    if "mfma" in t: nm += 1; continue
    if t.startswith(("s_waitcnt", "ds_read", "buffer_load", "global_load", "scratch", "s_load", "s_barrier", "s_cbranch", "s_branch", "v_readlane", "v_readfirstlane")):
        rows.append(f"mfma#{nm:4d} {'asm' if inasm else 'cc '} {s[:80]}")
n = int(sys.argv[3]) if len(sys.argv) > 3 else 200
print("\n".join(rows[:n]))
