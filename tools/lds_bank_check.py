#!/usr/bin/env python3
"""Bank-conflict model of the LDS images in csrc/fa_fwd_16.hip (rules: cdna_hip_programming.md §2,
MI355X_MICROARCH.md §LDS).  Prints the worst N-way conflict per access kind and DP."""
import itertools

B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
               [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS += [[l + 32 for l in g] for g in B128_GROUPS]


def k_off(DP, row, ch):
    sw = (ch ^ (row & 15)) if DP >= 128 else (ch ^ ((row >> 1) & 7)) if DP == 64 else (ch ^ ((row >> 2) & 3))
    return row * 2 * DP + 16 * sw


def v_off(DP, row, ch):
    sw = (ch ^ ((row & 3) << 2)) if DP >= 128 else (ch ^ (((row >> 1) & 1) << 2)) if DP == 64 else ch
    return row * 2 * DP + 16 * sw


def ways(addrs, width, nbanks):
    """max number of distinct dword rows touching one bank"""
    per_bank = {}
    for a in addrs:
        for d in range(width // 4):
            dw = a // 4 + d
            per_bank.setdefault(dw % nbanks, set()).add(dw)
    return max(len(s) for s in per_bank.values())


for DP in (32, 64, 128, 256):
    worst_k = 0
    for kb, ks in itertools.product(range(2), range(DP // 16)):
        for g in B128_GROUPS:
            addrs = [k_off(DP, 32 * kb + (l & 31), 2 * ks + (l >> 5)) for l in g]
            worst_k = max(worst_k, ways(addrs, 16, 64))
    worst_v = 0
    for i, st, plus8 in itertools.product(range(DP // 32), range(4), (0, 8)):
        for half in (range(0, 32), range(32, 64)):
            addrs = []
            for l in half:
                hi, qq, pp, g1 = l >> 5, (l >> 2) & 3, l & 3, (l >> 4) & 1
                row = 16 * st + 4 * hi + qq + plus8
                ch = 4 * i + 2 * g1 + (pp >> 1)
                addrs.append(v_off(DP, row, ch) + 8 * (pp & 1))
            worst_v = max(worst_v, ways(addrs, 8, 64))
    # staging writes: ds_write_b128, 8 consecutive lanes per LDS cycle, banks mod 32
    NCH = DP // 8
    worst_wk = worst_wv = 0
    for i in range(64 * NCH // 256):
        for base in range(0, 256, 8):
            cs = [base + j + 256 * i for j in range(8)]
            worst_wk = max(worst_wk, ways([k_off(DP, c // NCH, c % NCH) for c in cs], 16, 32))
            worst_wv = max(worst_wv, ways([v_off(DP, c // NCH, c % NCH) for c in cs], 16, 32))
    # injectivity of the images
    for f in (k_off, v_off):
        seen = {f(DP, r, c) for r in range(64) for c in range(NCH)}
        assert len(seen) == 64 * NCH and max(seen) < 64 * DP * 2, (DP, f.__name__)
    print(f"DP={DP:3d}: K ds_read_b128 {worst_k}-way, V ds_read_b64_tr_b16 {worst_v}-way, "
          f"K write {worst_wk}-way, V write {worst_wv}-way")


# ---- dual-use images of csrc/fa_bwd_16.hip: rows of 2*DP bytes read both by rows (ds_read_b128) and transposed
def d_off(DP, row, ch):
    f = (((row & 3) << 2) | ((row >> 2) & 3)) if DP >= 128 else (((row >> 2) & 3) | (((row >> 1) & 1) << 2))
    return 2 * DP * row + 16 * (ch ^ f)


for DP in (256, 128, 64):
    worst_row = worst_tr = 0
    for ks in range(DP // 16):
        for g in B128_GROUPS:
            worst_row = max(worst_row, ways([d_off(DP, l & 31, 2 * ks + (l >> 5)) for l in g], 16, 64))
    for i, s2, plus8 in itertools.product(range(DP // 32), range(2), (0, 8)):
        for half in (range(0, 32), range(32, 64)):
            addrs = []
            for l in half:
                hi, qq, pp, g1 = l >> 5, (l >> 2) & 3, l & 3, (l >> 4) & 1
                addrs.append(d_off(DP, 16 * s2 + 4 * hi + qq + plus8, 4 * i + 2 * g1 + (pp >> 1)) + 8 * (pp & 1))
            worst_tr = max(worst_tr, ways(addrs, 8, 64))
    seen = {d_off(DP, r, c) for r in range(32) for c in range(DP // 8)}
    assert len(seen) == 32 * DP // 8 and max(seen) < 32 * 2 * DP
    print(f"dual image (DP={DP}): row ds_read_b128 {worst_row}-way, transposed ds_read_b64_tr_b16 {worst_tr}-way")
