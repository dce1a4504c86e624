#!/usr/bin/env python3
"""Generate csrc/fa_fwd16_w64_body.inc: the hand-placed instruction stream of one key/value-tile iteration of the
64-rows-per-wave forward kernel (fa_fwd16_w64.hip).

Why a generator: every MFMA is an inline-asm statement whose operand classes pin O^T and Q^T to accumulator registers
("a") and the scores to arch VGPRs ("v"), and the VALU / LDS work of the online softmax is placed BETWEEN those
statements in a fixed order (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost': an MFMA
32x32x16 holds the SIMD's vector issue for 8 of its 32 cycles, so ~24 cycles of other issue fit in each gap).
The placement is computed here (earliest-deadline-first under a per-gap cycle budget) and written out as plain C++.

One iteration i (tile = 64 keys, wave = 64 query rows = 2 q-blocks):
  gaps  0..31  S_new = K(i) Q^T           (32 MFMA, kb-major so K(kb=0) registers free up at gap 16)
  gaps 32..63  O^T += V(i-1)^T P(i-1)^T   (32 MFMA, 16-key-step-major)
  fillers: exp/sum/pack of S_old -> P(i-1); transposed V(i-1) fragment reads; K(i) fragment reads a few gaps ahead
           of their MFMA (the first NPRE are issued by the includer right after the previous tile's barrier);
           row max of S_new, deferred-max decision, e = s*c - m in place.
Sections in the output, selected by W64_PART: 0 = first tile of a segment (no S_old), 1 / 2 = steady state (odd /
even tile: the two score sets swap roles), 3 / 4 = drain (no S_new).  The includer defines W64_VOFF, W64_KOFF,
W64_MFMA, W64_CVT and the variables the statements name (kf, vf, oacc, l4, mx, nm, c2).
"""
import os
import sys
from pathlib import Path

# timing experiments only (results become wrong): W64_ABL=EXP,ADD,... drops those filler kinds, W64_OUT overrides the
# output path so a variant library can be built beside the real one
ABL = set(filter(None, os.environ.get("W64_ABL", "").split(",")))

BUDGET = int(os.environ.get("W64_BUDGET", "24"))   # issue cycles available beside one MFMA
VR_EARLY, VR_LATE = (int(x) for x in os.environ.get("W64_VREAD", "10,6").split(","))   # V fragment read: gaps before its MFMA
KR_EARLY, KR_LATE = (int(x) for x in os.environ.get("W64_KREAD", "12,8").split(","))   # K fragment read
MIDBAR = os.environ.get("W64_MIDBAR", "0") == "1"  # per-tile barrier in the middle of the PV phase (see mid_barrier_streams)
EARLY_MAX = os.environ.get("W64_EARLY_MAX", "1") == "1"  # row max of key-block 0 during the QK of key-block 1, e = s*c - m spread to the end
class Cfg:
    """16-bit kernels: S = K Q^T is 32 MFMAs (8 k-steps of 16); int8 kernel (fa_fwd_w64_i8): 16 MFMAs
    (v_mfma_i32_32x32x32_i8, 4 k-steps of 32).  The integer scores never pass through v_cvt_f32_i32: the first
    MFMA of a score block accumulates onto a register tile holding the bits of 1.5 * 2^23 in every element, so the
    int32 result s + 0x4B400000 IS the float 12582912 + s (|s| <= 127 * 127 * 128 < 2^21 stays inside the mantissa);
    row max runs on those floats (monotone in s) and the one v_fma_f32 per score that applies scale and reference max
    takes the bias out with its addend (nmb = -m - 12582912 c).  W64_I8_BIAS=0 (lab, with -DW64_I8_NOBIAS) keeps the
    explicit in-place conversion (I2F)."""
    def __init__(self, i8):
        self.i8 = i8
        self.i2f = i8 and os.environ.get("W64_I8_BIAS", "1") == "0"
        self.NQK = 16 if i8 else 32      # MFMAs of the QK^T phase = first gap index of the PV phase
        self.KS = 4 if i8 else 8         # k-steps per 32-key block
        self.HALF = self.NQK // 2        # QK^T MFMAs per key block
        self.NG = self.NQK + 32          # gaps per tile
        self.KDMA = 2 if i8 else 4       # 1-KiB LDS-DMA pieces of a K tile per wave
        self.KB_BYTES = 4096 if i8 else 8192  # LDS bytes of one 32-key block of the K tile image


C = Cfg(False)

COST = {"UPDK": 4, "UPDV": 4, "BAR": 16, "KPRE": 4, "I2F": 4, "MASK": 8, "DMAK": 8, "DMAV": 8, "NOP": 32, "EXP": 8, "ADD": 4, "CVT": 4, "VREAD": 8, "KREAD": 4, "MAX": 4, "DEC": 44, "FMA": 4}
for _kv in filter(None, os.environ.get("W64_COST", "").split(",")):  # lab: W64_COST=EXP:16,ADD:4 overrides the model
    COST[_kv.split(":")[0]] = int(_kv.split(":")[1])

NPRE = 4  # K fragments (kb=0, ks<NPRE) read before the iteration starts (by the includer, after the barrier)


# ---- register map.  The kernel is compiled with amdgpu_num_vgpr(128), which on gfx950 confines the COMPILER to
# v[0:127] + a[0:127]; the upper halves are ours and are addressed by literal number in the asm text:
#   score tile (kb, qb) of set A at v[128 + 16 t : +15], of set B at v[192 + 16 t : +15], t = 2 kb + qb; the packed
#   P fragment of 16-key step `st` overwrites the first 4 registers of the matching 8-register half of its tile
#   (in-place compaction by v_cvt_pk);  O^T block (qb, db) at a[128 + 16 (4 qb + db) : +15].
# The Q^T fragments are compiler values constrained to the accumulator class ("a"): 64 of the compiler's a[0:127].
S_BASE, O_BASE = 128, 128
UPD_DELAY = int(os.environ.get("W64_DMA_UPD_DELAY", "4"))  # gaps between a DMA instruction and the update of its offset VGPR
DMA_MOD = os.environ.get("W64_DMA_MOD", "")  # cache-policy bits of the LDS-DMA loads, e.g. " nt" / " sc0" / " sc1"
DMASTAMP = os.environ.get("W64_LAB_DMASTAMP") == "1"  # lab: clock stamps around the first K DMA instruction and a control
NOUPD = os.environ.get("W64_LAB_NOUPD") == "1"  # lab: the DMA source offsets never advance (timing only)
QF_CLASS = os.environ.get("W64_LAB_QF", "a")  # lab: register class of the Q^T fragments


def base(setname, kb, qb):
    return S_BASE + (0 if setname == "a" else 64) + 16 * (2 * kb + qb)


def tup(setname, kb, qb):
    b = base(setname, kb, qb)
    return f"v[{b}:{b + 15}]"


def oreg(qb, db):
    r = O_BASE + 16 * (4 * qb + db)
    return f"a[{r}:{r + 15}]"


class Roles:
    def __init__(self, new, old):
        self.new, self.old = new, old


def qk_mfma(R, kb, ks, qb):
    t = tup(R.new, kb, qb)
    c = "0" if ks == 0 else t
    if ks == 0 and C.i8 and not C.i2f:
        return f'asm volatile(W64_MFMA_QK " {t}, %0, %1, %2" :: "v"(kf[{kb}][{ks}]), "{QF_CLASS}"(qf[{qb}][{ks}]), "v"(bias16));'
    return f'asm volatile(W64_MFMA_QK " {t}, %0, %1, {c}" :: "v"(kf[{kb}][{ks}]), "{QF_CLASS}"(qf[{qb}][{ks}]));'


def pv_mfma(R, st, db, qb):
    b = base(R.old, st >> 1, qb) + 8 * (st & 1)
    o = oreg(qb, db)
    return f'asm volatile(W64_MFMA " {o}, %0, v[{b}:{b + 3}], {o}" :: "v"(vf[{st}][{db}]));'


def op_text(R, op):
    kind = op[0]
    if kind == "EXP":
        _, kb, qb, r = op
        v = base(R.old, kb, qb) + r
        return f'asm volatile("v_exp_f32 v{v}, v{v}");'
    if kind == "ADD":
        _, kb, qb, r = op
        v = base(R.old, kb, qb) + r
        return f'asm volatile("v_add_f32 %0, %0, v{v}" : "+v"(l4[{qb}][{r & 3}]));'
    if kind == "CVT":
        _, kb, qb, r = op
        b = base(R.old, kb, qb)
        d = b + 8 * (r >> 3) + ((r & 7) >> 1)
        return f'asm volatile(W64_CVT " v{d}, v{b + r}, v{b + r + 1}");'
    if kind == "VREAD":
        _, st, db = op
        return f"vf[{st}][{db}] = v_frag(W64_VOFF, {db}, {st});"
    if kind == "KREAD":
        _, kb, ks = op
        return f"kf[{kb}][{ks}] = k_frag(W64_KOFF + {kb * C.KB_BYTES}, {ks});"
    if kind == "MAX":
        _, kb, qb, r, first = op
        v = base(R.new, kb, qb) + r
        c = (r >> 1) & 1  # two running maxima per q-block: consecutive ops never depend on each other
        if first:
            return f'asm volatile("v_max_f32 %0, v{v}, v{v + 1}" : "=v"(mx[{qb}][{c}]));'
        return f'asm volatile("v_max3_f32 %0, %0, v{v}, v{v + 1}" : "+v"(mx[{qb}][{c}]));'
    if kind == "BAR":
        return "W64_ITER_BARRIER();"
    if kind == "KPRE":
        return f"kf[0][{op[1]}] = k_frag(W64_KNEXT, {op[1]});"
    if kind == "I2F":
        _, kb, qb, r = op
        v = base(R.new, kb, qb) + r
        return f'asm volatile("v_cvt_f32_i32 v{v}, v{v}");'
    if kind == "MASK":
        # masking tiles: score (kb, qb, r) is masked  <=>  dmask[qb] + c > 0 with the element constant
        # c = 32 kb - 32 qb + (r & 3) + 8 (r >> 2) and the lane value dmask[qb] = max(causal term 64 t + 4 hi - row0 - ql
        # [key > row], key-tail term 64 t + 4 hi - Skv + 1 + 32 qb [key >= Skv]); masked scores become -inf before the max
        _, kb, qb, r = op
        v = base(R.new, kb, qb) + r
        c = 32 * kb - 32 * qb + (r & 3) + 8 * (r >> 2)
        return (f'asm volatile("v_cmp_lt_i32 vcc, {-c}, %0\\n\\tv_cndmask_b32 v{v}, v{v}, %1, vcc" :: "v"(dmask[{qb}]), "v"(neg_inf) : "vcc");')
    if kind == "DMAK" and DMASTAMP and op[1] == 0:
        return ('{ asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long d0_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); '
                'asm volatile("s_mov_b32 m0, %0\\n\\ts_nop 0\\n\\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_wave_k + W64_KDST + 0), "v"(kdma[0]), "s"(k_srd) : "memory"); '
                'const unsigned long long d1_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); st_wait += d1_ - d0_; }')
    if kind == "UPDK" and DMASTAMP and op[1] == 0:
        return ('{ asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long d0_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); '
                'asm volatile("s_mov_b32 m0, %0\\n\\ts_nop 0" ::"s"(lds_wave_k) : "memory"); '
                'const unsigned long long d1_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); st_bar += d1_ - d0_; } kdma[0] += k_step;')
    if kind == "DMAK":
        _, j = op
        return (f'asm volatile("s_mov_b32 m0, %0\\n\\ts_nop 0\\n\\tbuffer_load_dwordx4 %1, %2, 0 offen{DMA_MOD} lds" ::"s"(lds_wave_k + W64_KDST + {j * 1024}), '
                f'"v"(kdma[{j}]), "s"(k_srd) : "memory");')
    if kind == "DMAV":
        _, j = op
        return (f'asm volatile("s_mov_b32 m0, %0\\n\\ts_nop 0\\n\\tbuffer_load_dwordx4 %1, %2, 0 offen{DMA_MOD} lds" ::"s"(lds_wave + W64_VDST + {j * 1024}), '
                f'"v"(vdma[{j}]), "s"(v_srd) : "memory");')
    if kind == "UPDK":  # the source offset of piece j moves on to the next tile, a few gaps behind its DMA instruction
        # (a VALU write to a VGPR that an issued VMEM instruction still has to read would wait for it; measured neutral)
        return "" if NOUPD else f"kdma[{op[1]}] += k_step;"
    if kind == "UPDV":
        return "" if NOUPD else f"vdma[{op[1]}] += v_step;"
    if kind == "DEC":
        return "W64_DECIDE();"
    if kind == "NOP":
        return 'asm volatile("s_nop 15\\n\\ts_nop 15");  // MFMA result -> VALU read distance when no MFMA follows'
    if kind == "FMA":
        _, kb, qb, r = op
        v = base(R.new, kb, qb) + r
        nm = "nmb" if C.i8 and not C.i2f else "nm"
        return f'asm volatile("v_fma_f32 v{v}, v{v}, %0, %1" :: "s"(c2), "v"({nm}[{qb}]));'
    raise ValueError(kind)


def exp_streams():
    """exp -> row sum -> pack of S_old, one stream per score block (kb, qb), skewed by one pair so an exp result is
    not consumed by the next instruction; the 16-key step st = 2 kb + (r >> 3) must be packed before its PV MFMAs."""
    streams = []
    for kb in (0, 1):
        for qb in (0, 1):
            ops = []
            prev = None
            for r in list(range(0, 16, 2)) + [None]:
                if r is not None:
                    dl = C.NQK + 8 * (2 * kb + (r >> 3)) - 2
                    ops.append((("EXP", kb, qb, r), 0, dl))
                    ops.append((("EXP", kb, qb, r + 1), 0, dl))
                if prev is not None:
                    dl = C.NQK + 8 * (2 * kb + (prev >> 3)) - 2
                    ops.append((("ADD", kb, qb, prev), 0, dl))
                    ops.append((("ADD", kb, qb, prev + 1), 0, dl))
                    ops.append((("CVT", kb, qb, prev), 0, dl))
                prev = r
            streams.append(ops)
    return streams


def bar_gap():
    return C.NQK + 16  # MIDBAR: behind the PV MFMAs of 16-key steps 0 and 1


def vread_stream(have_new=True):
    ops = []
    for st in range(4):
        for db in range(4):
            use = C.NQK + st * 8 + db * 2
            early, late = max(0, use - VR_EARLY), use - VR_LATE
            if MIDBAR and have_new:  # every V read of the tile sits before the barrier (the slot is refilled behind it)
                late = min(late, bar_gap() - 2)
                early = min(early, late - 3)
            ops.append((("VREAD", st, db), early, late))
    return ops


def mid_barrier_streams(have_old):
    """MIDBAR: the per-tile s_barrier sits in the MIDDLE of the PV phase instead of at the end of the tile.  The MFMAs
    right behind it (PV of 16-key steps 2, 3) have their operands in registers already, so nothing waits on LDS
    there, and the first K fragments of the NEXT tile (visible only after this barrier) are read under those MFMAs
    instead of in front of the next tile's first MFMA.  WAR safety: all V reads of this tile come before the
    barrier (the V slot is refilled by the next tile's DMA), all K reads are in the first half anyway."""
    if have_old:
        g = bar_gap()
        return [[(("BAR",), g, g)], [(("KPRE", ks), g + 1, g + 10) for ks in range(min(NPRE, C.KS))]]
    last = C.NG - 1  # first tile of a segment (no PV): barrier + prefetch close the part
    return [[(("BAR",), last, last)] + [(("KPRE", ks), last, last) for ks in range(min(NPRE, C.KS))]]


def kread_stream():
    ops = []
    for kb in (0, 1):
        for ks in range(C.KS):
            if kb == 0 and ks < min(NPRE, C.KS):
                continue
            use = kb * C.HALF + ks * 2
            ops.append((("KREAD", kb, ks), max(0, use - KR_EARLY), max(0, use - KR_LATE)))
    return ops


def dma_stream():
    """LDS-DMA of K(i+1) and V(i), issued in the first gaps of tile i (their ring slots were released by the barrier
    that ended tile i-1) so that the VMEM issue overlaps MFMA execution; waited on at the end of tile i."""
    # first gap of each instruction (window of 3 gaps).  Two LDS-DMA instructions in one gap stall the MFMA behind
    # them: back-to-back placement -7 %, one every 2 gaps baseline, one every 4 gaps +6.5 % (same-box A/B at FLUX);
    # W64_DMA_K / W64_DMA_V (W64_DMA_K_I8 / _V_I8) override for experiments.
    sfx = "_I8" if C.i8 else ""
    kpos = [1, 4] if C.i8 else [1, 9, 17, 25]
    vpos = [2, 6, 8, 10] if C.i8 else [5, 13, 21, 29]
    if os.environ.get("W64_DMA_K" + sfx):
        kpos = [int(x) for x in os.environ["W64_DMA_K" + sfx].split(",")]
    if os.environ.get("W64_DMA_V" + sfx):
        vpos = [int(x) for x in os.environ["W64_DMA_V" + sfx].split(",")]
    ops = [(("DMAK", j), kpos[j], kpos[j] + 2) for j in range(C.KDMA)] + [(("DMAV", j), vpos[j], vpos[j] + 2) for j in range(4)]
    ops.sort(key=lambda o: o[1])
    return ops


def dma_update_stream(dma_ops):
    """offset updates, UPD_DELAY gaps behind the deadline of their DMA instruction"""
    last = C.NG - 1
    ups = [(("UPDK" if o[0][0] == "DMAK" else "UPDV", o[0][1]), min(last, o[2] + UPD_DELAY), min(last, o[2] + UPD_DELAY + 3)) for o in dma_ops]
    return ups


def start_streams(have_new, mfma_follows=True, masked=False):
    """(int8: convert the integer scores ->) (masking tiles: mask ->) row max of S_new -> decision -> e = s*c - m in
    place.  Returns several streams (each consumed in order); cross-stream order is enforced by disjoint gap windows:
    convert(kb) | mask(kb) | max(kb) | decision | the four blocks of e = s*c - m in parallel.  Four independent max
    chains (two per q-block) are interleaved so consecutive ops never depend on each other."""
    if not have_new:
        return []
    last = C.NG - 1
    H = C.HALF
    if not mfma_follows:  # first tile of a segment: nothing to hide under, plain order
        ops = []
        for kb in (0, 1):
            ready = kb * H + (H - 2) + 2
            if kb == 1:
                ops.append((("NOP",), C.NQK, last))
            for qb in (0, 1):
                for r in range(16):
                    if C.i2f:
                        ops.append((("I2F", kb, qb, r), ready + qb, last))
            if masked:
                for qb in (0, 1):
                    for r in range(16):
                        ops.append((("MASK", kb, qb, r), ready + qb, last))
            for r in range(0, 16, 2):
                for qb in (0, 1):
                    ops.append((("MAX", kb, qb, r, kb == 0 and r < 4), ready + qb, last))
        ops.append((("DEC",), C.NQK + 2, last))
        for kb in (0, 1):
            for qb in (0, 1):
                for r in range(16):
                    ops.append((("FMA", kb, qb, r), C.NQK + 4, last))
        return [ops]
    # gap windows (start, deadline) per stage; kb = 0 scores are complete at gap HALF, kb = 1 at gap NQK
    if not C.i8:
        if masked:
            w = {"mask": {0: (16, 30), 1: (32, 44)}, "max": {0: (31, 36), 1: (45, 50)}, "dec": (51, 52), "fma": (53, 62)}
        else:
            w = {"max": {0: (16, 31), 1: (32, 40)}, "dec": (41, 43), "fma": (44, 61)}
    elif C.i2f:
        if masked:
            w = {"i2f": {0: (9, 13), 1: (17, 21)}, "mask": {0: (14, 20), 1: (22, 29)}, "max": {0: (21, 24), 1: (30, 33)},
                 "dec": (34, 35), "fma": (36, 46)}
        else:
            w = {"i2f": {0: (9, 14), 1: (17, 22)}, "max": {0: (15, 19), 1: (23, 27)}, "dec": (28, 29), "fma": (30, 45)}
    else:
        if masked:
            w = {"mask": {0: (9, 17), 1: (17, 26)}, "max": {0: (18, 21), 1: (27, 30)}, "dec": (31, 32), "fma": (33, 46)}
        else:
            w = {"max": {0: (9, 16), 1: (17, 24)}, "dec": (25, 26), "fma": (27, 45)}
    streams = []
    for kb in (0, 1):
        ready = kb * H + (H - 2) + 2
        if C.i2f:
            for qb in (0, 1):
                streams.append([(("I2F", kb, qb, r), max(w["i2f"][kb][0], ready + qb), w["i2f"][kb][1]) for r in range(16)])
        if masked:
            for qb in (0, 1):
                streams.append([(("MASK", kb, qb, r), max(w["mask"][kb][0], ready + qb), w["mask"][kb][1]) for r in range(16)])
        mx = []
        for r in range(0, 16, 2):
            for qb in (0, 1):
                mx.append((("MAX", kb, qb, r, kb == 0 and r < 4), max(w["max"][kb][0], ready + qb), w["max"][kb][1]))
        streams.append(mx)
    streams.append([(("DEC",), w["dec"][0], w["dec"][1])])
    for kb in (0, 1):
        for qb in (0, 1):
            streams.append([(("FMA", kb, qb, r), w["fma"][0], w["fma"][1]) for r in range(16)])
    return streams


def schedule(streams, gaps):
    """EDF under a per-gap budget.  streams: list of [ (op, earliest, deadline) ... ] each consumed in order."""
    pos = [0] * len(streams)
    out = [[] for _ in range(gaps)]
    # a stream is consumed in order, so an op inherits the tightest deadline of everything queued behind it
    tight = []
    for st_ in streams:
        t = list(st_)
        for k in range(len(t) - 2, -1, -1):
            if t[k + 1][2] < t[k][2]:
                t[k] = (t[k][0], t[k][1], t[k + 1][2])
        tight.append(t)
    streams = tight
    for g in range(gaps):
        used = 0
        while True:
            best = None
            for si, s in enumerate(streams):
                if pos[si] >= len(s):
                    continue
                op, earliest, deadline = s[pos[si]]
                if earliest > g:
                    continue
                remaining = sum(COST[o[0][0]] for o in s[pos[si]:])
                # how far this stream is behind an even spread up to its last deadline
                last_dl = min(gaps - 1, max(d for _, _, d in s[pos[si]:]))
                need_rate = remaining / max(1, (last_dl - g + 1))
                forced = deadline <= g
                key = (0 if forced else 1, -need_rate)
                if best is None or key < best[0]:
                    best = (key, si, forced)
            if best is None:
                break
            _, si, forced = best
            op, earliest, deadline = streams[si][pos[si]]
            c = COST[op[0]]
            if not forced and used + c > BUDGET and used > 0:
                break
            if not forced and used >= BUDGET:
                break
            out[g].append(op)
            used += c
            pos[si] += 1
    for si, s in enumerate(streams):
        assert pos[si] == len(s), f"stream {si} not fully placed ({pos[si]}/{len(s)})"
    return out


def check_part(placed, have_new, have_old, masked):
    """Data-flow self-check of one scheduled part (gap g = after MFMA g): every consumer sits behind its producer.
    (A schedule that packed a P fragment one gap late shows up on the GPU as garbage in exactly the O^T blocks whose
    MFMAs came first - cost a long bisect once.)"""
    pos = {}
    for g, ops in enumerate(placed):
        for k, op in enumerate(ops):
            pos[op] = (g, k)
    def before(a, b):
        return a in pos and b in pos and pos[a] < pos[b]
    for kb in (0, 1):
        for qb in (0, 1):
            if have_old:
                for r in range(16):
                    assert before(("EXP", kb, qb, r), ("ADD", kb, qb, r)), ("ADD before EXP", kb, qb, r)
                for r in range(0, 16, 2):
                    cv = ("CVT", kb, qb, r)
                    assert before(("EXP", kb, qb, r), cv) and before(("EXP", kb, qb, r + 1), cv), ("CVT before EXP", cv)
                    assert before(("ADD", kb, qb, r), cv) and before(("ADD", kb, qb, r + 1), cv), ("CVT before ADD", cv)
                    # in-place compaction: pair (r, r+1) lands in register 8*(r>>3) + (r&7)/2 of the tile, which must
                    # already have been consumed as a score (its own ADD and the CVT that read it)
                    dst = 8 * (r >> 3) + ((r & 7) >> 1)
                    if dst not in (r, r + 1):
                        assert before(("ADD", kb, qb, dst), cv), ("CVT overwrites unread score", cv)
                        assert before(("CVT", kb, qb, dst & ~1), cv), ("CVT overwrites unpacked score", cv)
                    st = 2 * kb + (r >> 3)
                    first_use = C.NQK + 8 * st  # first PV MFMA of this 16-key step (any d-block, any q-block)
                    assert pos[cv][0] < first_use, ("P fragment packed after its first PV MFMA", cv, pos[cv], first_use)
            if have_new:
                last_mfma = kb * C.HALF + (C.HALF - 2) + qb
                for r in range(0, 16, 2):
                    mxop = [o for o in pos if o[0] == "MAX" and o[1:4] == (kb, qb, r)][0]
                    assert pos[mxop][0] > last_mfma, ("row max reads an unfinished score tile", mxop)
                    if C.i2f:
                        assert before(("I2F", kb, qb, r), mxop) and before(("I2F", kb, qb, r + 1), mxop), ("MAX before I2F", mxop)
                        assert pos[("I2F", kb, qb, r)][0] > last_mfma, ("convert on an unfinished score tile", kb, qb, r)
                        if masked:
                            assert before(("I2F", kb, qb, r), ("MASK", kb, qb, r)), ("MASK before I2F", kb, qb, r)
                            assert before(("I2F", kb, qb, r + 1), ("MASK", kb, qb, r + 1)), ("MASK before I2F", kb, qb, r + 1)
                    if masked:
                        assert before(("MASK", kb, qb, r), mxop) and before(("MASK", kb, qb, r + 1), mxop), ("MAX before MASK", mxop)
                    assert before(mxop, ("DEC",)), ("decision before MAX", mxop)
                for r in range(16):
                    assert before(("DEC",), ("FMA", kb, qb, r)), ("FMA before decision", kb, qb, r)
                    if masked:
                        assert pos[("MASK", kb, qb, r)][0] > last_mfma, ("mask on an unfinished score tile", kb, qb, r)
    if MIDBAR and have_new:
        bar = pos[("BAR",)]
        for op, at in pos.items():
            if op[0] in ("VREAD", "KREAD", "DMAK", "DMAV"):
                assert at < bar, ("before the barrier", op)
            if op[0] == "KPRE":
                assert at > bar, ("next tile's K fragments are visible only behind the barrier", op)
    if have_old:
        for st in range(4):
            for db in range(4):
                assert pos[("VREAD", st, db)][0] < C.NQK + st * 8 + db * 2, ("V fragment read after its MFMA", st, db)
    if have_new:
        for kb in (0, 1):
            for ks in range(C.KS):
                if ("KREAD", kb, ks) in pos:
                    assert pos[("KREAD", kb, ks)][0] < kb * C.HALF + ks * 2, ("K fragment read after its MFMA", kb, ks)


def emit_part(lines, R, have_new, have_old, masked=False):
    mf = []
    if have_new and os.environ.get("W64_LAB_QK_ORDER") == "il":  # lab (timing only): four accumulators round-robin
        for ks in range(C.KS):
            for kb in (0, 1):
                for qb in (0, 1):
                    mf.append(qk_mfma(R, kb, ks, qb))
    elif have_new:
        for kb in (0, 1):
            for ks in range(C.KS):
                for qb in (0, 1):
                    mf.append(qk_mfma(R, kb, ks, qb))
    else:
        mf += [None] * C.NQK
    if have_old:
        for st in range(4):
            for db in range(4):
                for qb in (0, 1):
                    mf.append(pv_mfma(R, st, db, qb))
    else:
        mf += [None] * 32
    streams = []
    if have_old:
        streams += exp_streams()
        streams.append(vread_stream(have_new))
    if have_new:
        streams += start_streams(True, have_old, masked)
        streams.append(kread_stream())
        dmas = dma_stream()
        streams.append(dmas)
        streams.append(dma_update_stream(dmas))
        if MIDBAR:
            streams += mid_barrier_streams(have_old)
    placed = schedule(streams, C.NG)
    if not ABL:
        check_part(placed, have_new, have_old, masked)
    cyc = 0
    for g in range(C.NG):
        if mf[g] is not None:
            lines.append(mf[g])
        for op in placed[g]:
            if op[0] not in ABL:
                lines.append("    " + op_text(R, op))
        fill = sum(COST[o[0]] for o in placed[g])
        cyc += max(32 if mf[g] else 0, (8 if mf[g] else 0) + fill)
        if mf[g] is not None or placed[g]:
            lines.append(f"__builtin_amdgcn_sched_barrier(0);  // gap {g}: filler issue {fill} cyc")
    lines.append(f"// modelled issue time of this part: {cyc} cycles")


def emit_helpers(lines):
    """Helpers that touch the asm-owned O^T registers a[128:255] by literal number."""
    a = lines.append
    a("// GENERATED by tools/gen_w64_body.py -- helpers that address the asm-owned O^T registers a[128:255].")
    a("__device__ __forceinline__ void zero_o() {")
    a("    asm volatile(" + " ".join(f'"v_accvgpr_write_b32 a{O_BASE + r}, 0\\n\\t"' for r in range(128)) + ' "s_nop 0" ::: "memory", "v255", "a255");')
    a("}")
    for qb in (0, 1):
        b0 = O_BASE + qb * 64
        a(f"// O^T of q-block {qb} *= alpha (rare path of the deferred max)")
        a(f"__device__ __forceinline__ void scale_o{qb}(float alpha) {{")
        a("    float t0, t1, t2, t3;")
        for r in range(b0, b0 + 64, 4):
            a(f'    asm volatile("v_accvgpr_read_b32 %0, a{r}\\n\\tv_accvgpr_read_b32 %1, a{r + 1}\\n\\tv_accvgpr_read_b32 %2, a{r + 2}\\n\\t"'
              f' "v_accvgpr_read_b32 %3, a{r + 3}\\n\\ts_nop 1\\n\\tv_mul_f32 %0, %0, %4\\n\\tv_mul_f32 %1, %1, %4\\n\\tv_mul_f32 %2, %2, %4\\n\\t"'
              f' "v_mul_f32 %3, %3, %4\\n\\ts_nop 1\\n\\tv_accvgpr_write_b32 a{r}, %0\\n\\tv_accvgpr_write_b32 a{r + 1}, %1\\n\\t"'
              f' "v_accvgpr_write_b32 a{r + 2}, %2\\n\\tv_accvgpr_write_b32 a{r + 3}, %3"'
              ' : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(alpha));')
        a("}")
        a(f"// 16 registers of O^T block (q-block {qb}, d-block DB) -> floats")
        a(f"template <int DB> __device__ __forceinline__ void read_o{qb}(float (&o)[16]) {{")
        for db in range(4):
            a(f"    if constexpr (DB == {db}) {{")
            for r in range(0, 16, 4):
                rr = b0 + 16 * db + r
                a(f'        asm volatile("v_accvgpr_read_b32 %0, a{rr}\\n\\tv_accvgpr_read_b32 %1, a{rr + 1}\\n\\t"'
                  f' "v_accvgpr_read_b32 %2, a{rr + 2}\\n\\tv_accvgpr_read_b32 %3, a{rr + 3}"'
                  f' : "=v"(o[{r}]), "=v"(o[{r + 1}]), "=v"(o[{r + 2}]), "=v"(o[{r + 3}]));')
            a("    }")
        a("}")


def emit_body(out):
    lines = ["// GENERATED by tools/gen_w64_body.py -- do not edit; see that file for the placement rules.",
             "#if W64_PART == 0  // first tile of a segment: S -> set A"]
    emit_part(lines, Roles("a", "b"), True, False)
    lines.append("#elif W64_PART == 1  // steady state, odd tile: S -> set B, P from set A")
    emit_part(lines, Roles("b", "a"), True, True)
    lines.append("#elif W64_PART == 2  // steady state, even tile: S -> set A, P from set B")
    emit_part(lines, Roles("a", "b"), True, True)
    lines.append("#elif W64_PART == 3  // drain: P of the last tile (set A), then its PV")
    emit_part(lines, Roles("b", "a"), False, True)
    lines.append("#elif W64_PART == 4  // drain, last tile in set B")
    emit_part(lines, Roles("a", "b"), False, True)
    lines.append("#elif W64_PART == 5  // masking tile (causal diagonal / ragged last key tile), first tile of a segment")
    emit_part(lines, Roles("a", "b"), True, False, masked=True)
    lines.append("#elif W64_PART == 6  // masking tile, odd")
    emit_part(lines, Roles("b", "a"), True, True, masked=True)
    lines.append("#elif W64_PART == 7  // masking tile, even")
    emit_part(lines, Roles("a", "b"), True, True, masked=True)
    lines.append("#endif")
    out.write_text("\n".join(lines) + "\n")
    print("wrote", out, len(lines), "lines")


def main():
    global C
    csrc = Path(__file__).resolve().parent.parent / "universal-metal-flash-attention_amd" / "csrc"
    helpers = []
    emit_helpers(helpers)
    (csrc / "fa_fwd16_w64_regs.inc").write_text("\n".join(helpers) + "\n")
    C = Cfg(False)
    emit_body(Path(os.environ["W64_OUT"]) if os.environ.get("W64_OUT") else csrc / "fa_fwd16_w64_body.inc")
    C = Cfg(True)
    emit_body(Path(os.environ["W64_OUT_I8"]) if os.environ.get("W64_OUT_I8") else csrc / "fa_fwd_w64_i8_body.inc")


if __name__ == "__main__":
    sys.exit(main())
