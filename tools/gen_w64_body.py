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

BUDGET = int(os.environ.get("W64_BUDGET", "28"))   # issue cycles available beside one MFMA
VR_EARLY, VR_LATE = (int(x) for x in os.environ.get("W64_VREAD", "10,6").split(","))   # V fragment read: gaps before its MFMA
KR_EARLY, KR_LATE = (int(x) for x in os.environ.get("W64_KREAD", "12,8").split(","))   # K fragment read
MIDBAR = os.environ.get("W64_MIDBAR", "0") == "1"  # per-tile barrier in the middle of the PV phase (see mid_barrier_streams)
# issue cycles of fillers placed BEFORE the first MFMA of a tile body: that MFMA waits for K fragments read right behind
# the tile's barrier (LDS latency, ~150 cycles); softmax work of the previous tile runs in that shadow instead of behind it
PRE = {"16": int(os.environ.get("W64_PRE", "0")), "i8": int(os.environ.get("W64_PRE_I8", "0")), "f8": int(os.environ.get("W64_PRE_F8", "0"))}
ROTATE = int(os.environ.get("W64_LAB_ROTATE", "0"))  # lab, TIMING ONLY (results wrong): steady-state bodies start with their last ROTATE gaps
FSTAMP = [int(x) for x in os.environ.get("W64_LAB_FSTAMP", "").split(",") if x]  # lab: up to 8 wait-free s_memtime stamps in front of these gaps
GAPSTAMP = [int(x) for x in os.environ.get("W64_LAB_GAPSTAMP", "").split(",") if x]  # lab: clock stamps in front of these gaps (steady-state parts)
SPEC = os.environ.get("W64_SPEC", "1") == "1"  # steady-state tiles: e = s*c - m against the CURRENT reference first, row max of e, cheap decision (spec_streams)
EARLY_MAX = os.environ.get("W64_EARLY_MAX", "1") == "1"  # row max of key-block 0 during the QK of key-block 1, e = s*c - m spread to the end
MA_DEPTH = int(os.environ.get("W64_MA_DEPTH", "2"))  # additive-mask bodies (MASKA): mask fragments read this many 4-score groups ahead of their use (per score block)
class Cfg:
    """16-bit kernels: S = K Q^T is 32 MFMAs (8 k-steps of 16); int8 kernel (fa_fwd_w64_i8): 16 MFMAs
    (v_mfma_i32_32x32x32_i8, 4 k-steps of 32).  The integer scores never pass through v_cvt_f32_i32: the first
    MFMA of a score block accumulates onto a register tile holding the bits of 1.5 * 2^23 in every element, so the
    int32 result s + 0x4B400000 IS the float 12582912 + s (|s| <= 127 * 127 * 128 < 2^21 stays inside the mantissa);
    row max runs on those floats (monotone in s) and the one v_fma_f32 per score that applies scale and reference max
    takes the bias out with its addend (nmb = -m - 12582912 c).  W64_I8_BIAS=0 (lab, with -DW64_I8_NOBIAS) keeps the
    explicit in-place conversion (I2F)."""
    def __init__(self, i8, f8=False, d64=False, madd=False):
        self.i8 = i8
        # additive fp16 mask tensors (fa_fwd16_w64_bias_*, round 6): a body file of its own -- the masking bodies read the wave's mask tile from LDS (MLD) and add
        # mask / scale with one v_fma_mix_f32 per score; EVERY body carries the eight LDS-DMA instructions of the next listed tile's mask image (DMAM)
        self.madd = madd
        self.f8 = f8                     # fp8 (e4m3) P and V: O^T += V^T P^T on v_mfma_scale_f32_32x32x64_f8f6f4 (see Cfg8 notes)
        # head_dim 64 (16-bit kernels only, fa_fwd16_w64d64): half the k-steps of S = K Q^T and two d-blocks of O^T instead of
        # four -- 32 MFMAs per 64-key tile for the same softmax work, rows of 128 bytes in the K / V tile images (the int8
        # kernels' K geometry).  The O^T register map is head_dim 128's with d-blocks 0 and 1 of each q-block in use.
        self.d64 = d64
        assert not (d64 and (i8 or f8)) and not (madd and (i8 or f8))
        self.i2f = i8 and os.environ.get("W64_I8_BIAS", "1") == "0"
        # row sums on the matrix pipe (16-bit P kernels): l += sum of the ROUNDED P fragment by v_mfma_f32_4x4x4_16b against an
        # all-ones operand (a lane-local sum: block b = lane / 4, column j = lane % 4 -> the lane's own four values; 8-cycle
        # instruction).  16 of them per tile replace 64 v_add_f32, and l is the sum of exactly the values P V consumes.
        # OFF by default (lab option W64_MSUM=1 with -DW64_MSUM_ON=1): same-box A/B, every shape slower than the v_add form
        # (FLUX -3.5 %, B4 H16 S8192 causal -3.8 %, int8 -1 %): alternating with 32x32x16 MFMAs a 4x4x4 costs the matrix
        # pipe 16 cycles, 256 per tile, and the loop is not VALU-issue bound enough to win them back (profiles/r3/lab_notes.md).
        self.msum = (not f8) and os.environ.get("W64_MSUM", "0") == "1"
        self.NQK = 16 if (i8 or d64) else 32      # MFMAs of the QK^T phase = first gap index of the PV phase
        self.KS = 4 if (i8 or d64) else 8         # k-steps per 32-key block
        self.HALF = self.NQK // 2        # QK^T MFMAs per key block
        self.NDB = 2 if d64 else 4       # 32-row d-blocks of O^T
        self.PVS = 2 * self.NDB          # PV MFMAs per 16-key step (d-blocks x two q-blocks)
        self.NPV = 20 if f8 else 4 * self.PVS  # gap slots of the PV phase (a 64-cycle fp8 MFMA takes two 32-cycle slots)
        self.NG = self.NQK + self.NPV    # gaps per tile
        self.KDMA = 2 if (i8 or d64) else 4       # 1-KiB LDS-DMA pieces of a K tile per wave
        self.VDMA = 2 if (f8 or d64) else 4       # ... of a V tile
        self.KB_BYTES = 4096 if (i8 or d64) else 8192  # LDS bytes of one 32-key block of the K tile image
        # per-gap filler budget (cost-model cycles).  The tile's filler work must FIT: what does not is forced into its deadline
        # gap, and the int8 body had 248 cycles piled into gap 45, the bf16 body 128 into gap 61 -- right in front of the
        # tile's barrier, where nothing overlaps them (gap stamps, profiles/r2/lab_notes.md).  Defaults = total / gaps, rounded up.
        self.budget = int(os.environ.get("W64_BUDGET_F8", "40")) if f8 else int(os.environ.get("W64_BUDGET_I8", "36")) if i8 else \
            int(os.environ.get("W64_BUDGET_D64", "48")) if d64 else BUDGET

    def pv_gap(self, qb, db):
        """gap slot of the fp8 PV MFMA of (q-block, d-block); db = 4: the row-sum MFMA of the q-block"""
        return self.NQK + 10 * qb + 2 * db


COST = {"MLD": 8, "DMAM": 8, "MSUM": 8, "LCHK": 12, "DEC2": 20, "MXINIT": 8, "MAXE": 4, "CVT8": 4, "VREAD8": 8, "UPDK": 4, "UPDV": 4, "BAR": 16, "KPRE": 4, "I2F": 4, "MASK": 8, "DMAK": 8, "DMAV": 8, "NOP": 32, "EXP": 8, "ADD": 4, "CVT": 4, "VREAD": 8, "KREAD": 4, "MAX": 4, "DEC": 44, "FMA": 4}
for _kv in filter(None, os.environ.get("W64_COST", "").split(",")):  # lab: W64_COST=EXP:16,ADD:4 overrides the model
    COST[_kv.split(":")[0]] = int(_kv.split(":")[1])

C = Cfg(False)

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


LAB_SHAPE16 = os.environ.get("W64_LAB_SHAPE16") == "1"  # lab, TIMING ONLY (results are garbage of the right distribution): every
# 32x32x16 MFMA becomes two 16x16x32 on the same operand registers, each accumulator quarter fed by every other k-step -- the
# same FLOPs, LDS reads, vector work and operand statistics; measures what the MFMA shape is worth at the power cap


def qk_mfma(R, kb, ks, qb):
    t = tup(R.new, kb, qb)
    c = "0" if ks == 0 else t
    if LAB_SHAPE16 and not C.i8:
        b = base(R.new, kb, qb)
        out = []
        for q in (2 * (ks & 1), 2 * (ks & 1) + 1):
            tq = f"v[{b + 4 * q}:{b + 4 * q + 3}]"
            out.append(f'asm volatile(W64_MFMA16 " {tq}, %0, %1, {"0" if ks < 2 else tq}" :: "v"(kf[{kb}][{ks}]), "{QF_CLASS}"(qf[{qb}][{ks}]));')
        return " ".join(out)
    if ks == 0 and C.f8:
        return f'asm volatile(W64_MFMA_QK " {t}, %0, %1, v[{F8_BIAS}:{F8_BIAS + 15}]" :: "v"(kf[{kb}][{ks}]), "{QF_CLASS}"(qf[{qb}][{ks}]));'
    if ks == 0 and C.i8 and not C.i2f:
        # literal bias tile (the kernel is compiled with amdgpu_num_vgpr(112)): as a compiler value hipcc parks it in AGPRs
        # and copies it back with 16 v_accvgpr_read in front of EVERY tile as soon as anything else in the kernel changes
        return f'asm volatile(W64_MFMA_QK " {t}, %0, %1, v[{I8_BIAS}:{I8_BIAS + 15}]" :: "v"(kf[{kb}][{ks}]), "{QF_CLASS}"(qf[{qb}][{ks}]));'
    return f'asm volatile(W64_MFMA_QK " {t}, %0, %1, {c}" :: "v"(kf[{kb}][{ks}]), "{QF_CLASS}"(qf[{qb}][{ks}]));'


# ---- fp8 variant (Cfg.f8): SageAttention2-style P V.  Per 64-key tile and q-block the WHOLE key range is ONE MFMA k-range
# (v_mfma_scale_f32_32x32x64_f8f6f4: 64 cycles, twice the bf16 rate per flop; operand maps measured by
# tools/lab/f8_probe.hip: lane l holds row/col l % 32 and k-slots 32 (l / 32) + byte).  The B operand P^T is the score
# accumulator packed in place: byte j = 4 m + b of lane half h is the score of key b + 8 (m & 3) + 4 h + 32 (m >> 2), i.e.
# register m of the packed fragment holds scores 4 m' .. 4 m' + 3 (m' = m & 3) of key block m >> 2 -- two v_cvt_pk_fp8_f32
# per register -- and the eight packed registers of a q-block sit in the first eight registers of its key-block-0 tile.
# The A operand V^T comes from the quantiser already in that k order (fa_quant.hip v8 image), two ds_read_b128 per
# d-block, shared by both q-blocks; its per-tile power-of-two scale rides in the instruction's scale_a operand.
# The row sum l = sum_k P is a ninth MFMA per q-block against an all-ones A operand (the MFMA pipe is half idle in
# this variant, the VALU is not): it sums exactly the rounded P that P V uses and removes 64 v_add per tile.
# Register ownership of the fp8 kernel: it is compiled with amdgpu_num_vgpr(96), so v[96:127] and a[96:127] are ours too.
# The row-sum accumulators MUST be literal registers: as compiler values ("+a") hipcc copied them between basic blocks
# right behind the asm statement, i.e. before the 64-cycle MFMA had written them (it pads nothing for inline asm) -- rows
# lost whole tiles of their sum (measured: O = 1.0 ... 1.33 for V = 1).
# msum kernels are compiled with amdgpu_num_vgpr(118): v[118:127] are asm-owned as well
MS_L = 118       # v[118:121] / v[122:125]: row-sum accumulators of q-block 0 / 1 (four identical columns each)
MS_ONES = 126    # v[126:127]: 1.0 in all four 16-bit k-slots, the A operand of the row-sum MFMA
I8_BIAS = 112    # v[112:127] of fa_fwd_w64_i8 (amdgpu_num_vgpr(112)): the int8 score bias tile (1.5 * 2^23 in every element)
F8_BIAS = 96     # v[96:111]  the int8 score bias tile (1.5 * 2^23 in every element)
F8_ONES = 112    # v[112:119] fp8 1.0 in every k-slot: A operand of the row-sum MFMA
F8_SONE = 120    # v120       E8M0 scale bytes of 1.0
F8_VSC = 121     # v121       E8M0 scale bytes of the V tile in use (set per tile by the includer)
F8_L = 96        # a[96:111] / a[112:127]: row-sum accumulators of q-block 0 / 1


def pv8_mfma(R, db, qb):
    b = base(R.old, 0, qb)
    o = oreg(qb, db)
    return (f'asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 {o}, %0, v[{b}:{b + 7}], {o}, v{F8_VSC}, v{F8_SONE} op_sel_hi:[0,0,0]" '
            f':: "v"(vf8[{db}]));')


def l8_mfma(R, qb):
    b = base(R.old, 0, qb)
    la = f"a[{F8_L + 16 * qb}:{F8_L + 16 * qb + 15}]"
    return (f'asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 {la}, v[{F8_ONES}:{F8_ONES + 7}], v[{b}:{b + 7}], {la}, '
            f'v{F8_SONE}, v{F8_SONE} op_sel_hi:[0,0,0]");')


def pv_mfma(R, st, db, qb):
    b = base(R.old, st >> 1, qb) + 8 * (st & 1)
    o = oreg(qb, db)
    if LAB_SHAPE16:
        r = O_BASE + 16 * (4 * qb + db)
        out = []
        for q in (2 * (st & 1), 2 * (st & 1) + 1):
            oq = f"a[{r + 4 * q}:{r + 4 * q + 3}]"
            out.append(f'asm volatile(W64_MFMA16 " {oq}, %0, v[{b}:{b + 3}], {oq}" :: "v"(vf[{st}][{db}]));')
        return " ".join(out)
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
    if kind == "MSUM":
        _, st, qb, h = op
        b = base(R.old, st >> 1, qb) + 8 * (st & 1) + 2 * h
        la = f"v[{MS_L + 4 * qb}:{MS_L + 4 * qb + 3}]"
        return f'asm volatile(W64_MSUM " {la}, v[{MS_ONES}:{MS_ONES + 1}], v[{b}:{b + 1}], {la}");'
    if kind == "CVT8":
        _, kb, qb, m, hi = op
        src = base(R.old, kb, qb) + 4 * m + (2 if hi else 0)
        d = base(R.old, 0, qb) + 4 * kb + m
        sel = " op_sel:[0,0,1]" if hi else ""
        return f'asm volatile("v_cvt_pk_fp8_f32 v{d}, v{src}, v{src + 1}{sel}");'
    if kind == "VREAD8":
        return f"vf8[{op[1]}] = v8_frag(W64_VOFF, {op[1]});"
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
        one = (f'asm volatile("v_cmp_lt_i32 vcc, {-c}, %0\\n\\tv_cndmask_b32 v{v}, v{v}, %1, vcc" :: "v"(dmask[{qb}]), "v"(neg_inf) : "vcc");')
        if C.i8:
            if C.f8:
                return one
            # (round 6) bool mask tensors on the int8 kernel: the bit form of the 16-bit kernels (the biased integer score, a float, or -inf)
            mt8 = (f'asm volatile("v_bfe_i32 %0, %1, {16 * kb + r}, 1\\n\\tv_bfi_b32 v{v}, %0, v{v}, %2" : "=&v"(mtmp) : "v"(mwc[{qb}]), "v"(neg_inf));')
            return f'if constexpr (MASKT) {{ {mt8} }} else {{ {one} }}'
        # sliding-window instantiations (WINDOW): the other band edge as well, score masked <=> dleft + c < 0
        two = (f'asm volatile("v_cmp_lt_i32 vcc, {-c}, %0\\n\\tv_cndmask_b32 v{v}, v{v}, %1, vcc\\n\\tv_cmp_gt_i32 vcc, {-c}, %2\\n\\t'
               f'v_cndmask_b32 v{v}, v{v}, %1, vcc" :: "v"(dmask[{qb}]), "v"(neg_inf), "v"(dleft) : "vcc");')
        # mask-tensor instantiations (MASKT): bit 16 kb + r of the lane's word mwc[qb] (fa_aux.hip mask_pack_kernel) says whether the
        # score attends: sign-extend the bit to a select mask, keep the score or take -inf
        mt = (f'asm volatile("v_bfe_i32 %0, %1, {16 * kb + r}, 1\\n\\tv_bfi_b32 v{v}, %0, v{v}, %2" : "=&v"(mtmp) : "v"(mwc[{qb}]), "v"(neg_inf));')
        # additive fp16 mask tensors (MASKA, round 6): the lane's four mask values of score group g = r >> 2 (keys 8 g + 4 hi ... + 3 of key block kb: 8 bytes of
        # the wave's mask tile in LDS, read by the MLD op) are added to the raw scores as mask / scale -- one v_fma_mix_f32 per score (f16 source half picked
        # by op_sel), so that the row max and e = s * c - m see s + mask / scale, i.e. (s * scale + mask) * log2 e
        # (temporaries per q-block, the eight groups of its two key blocks in ONE ordered stream: ma_slot)
        if C.madd:
            return (f'asm volatile("v_fma_mix_f32 v{v}, %0, %1, v{v} op_sel:[{r & 1},0,0] op_sel_hi:[1,0,0]" :: "v"(mta[{qb}][{ma_slot(kb, r >> 2)}][{(r & 3) >> 1}]), "s"(ma_k));')
        return f'if constexpr (MASKT) {{ {mt} }} else if constexpr (WINDOW) {{ {two} }} else {{ {one} }}'
    if kind == "MLD":  # MASKA: the mask fragment of score group g of block (kb, qb) from the wave's mask tile ring (compiler-visible LDS read: hipcc counts the wait)
        _, kb, qb, g = op
        return f"mta[{qb}][{ma_slot(kb, g)}] = ma_read(W64_MOFF + {4096 * qb}, {16 * (4 * kb + g)});"
    if kind == "DMAM":  # MASKA: piece j (8 rows) of the NEXT listed tile's mask image -> the other ring slot; skipped (wave-uniform) when this wave runs that tile unmasked
        _, j = op
        # (ONE running scalar for the source offset and the ring base + an immediate for M0: as sixteen distinct scalar operands per tile the pieces cost the
        # kernel scalar-register spills into vector lanes, and those vector registers)
        return (f'if (ma_col >= 0) {{ asm volatile("s_add_u32 m0, %0, %4\\n\\ts_nop 0\\n\\tbuffer_load_dwordx4 %1, %2, %3 offen{DMA_MOD} lds" ::"s"(ma_ring), '
                f'"v"(ma_voff), "s"(ma_srd), "s"(ma_soff), "n"(W64_MDST + {j * 1024}) : "memory", "scc"); ma_soff += ma_rs8; }}')
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
    if kind == "DEC2":
        return "W64_DECIDE2();"
    if kind == "MXINIT":
        return "mx[0][0] = mx[0][1] = mx[1][0] = mx[1][1] = neg_inf;"
    if kind == "MAXE":  # row max of e = s*c - m (after the FMA): always the accumulating form, mx starts at -inf
        _, kb, qb, r = op
        v = base(R.new, kb, qb) + r
        return f'asm volatile("v_max3_f32 %0, %0, v{v}, v{v + 1}" : "+v"(mx[{qb}][{(r >> 1) & 1}]));'
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
                    dl = C.NQK + C.PVS * (2 * kb + (r >> 3)) - 2
                    ops.append((("EXP", kb, qb, r), 0, dl))
                    ops.append((("EXP", kb, qb, r + 1), 0, dl))
                if prev is not None:
                    dl = C.NQK + C.PVS * (2 * kb + (prev >> 3)) - 2
                    if not C.msum:
                        ops.append((("ADD", kb, qb, prev), 0, dl))
                        ops.append((("ADD", kb, qb, prev + 1), 0, dl))
                    ops.append((("CVT", kb, qb, prev), 0, dl))
                prev = r
            streams.append(ops)
    return streams


def msum_stream():
    """row-sum MFMAs of the packed P fragments: two per (16-key step, q-block) (a 4x4x4 B operand is two registers = four
    values); a fragment is complete one gap before its first P V MFMA (exp_streams deadline), and the set is overwritten
    by the next tile's scores, so everything sits inside the P V phase.  The last one keeps >= 3 MFMA issues to the end
    of the body: the includer's reads of l (v_cmp in the lazy check, the segment's final read) need the result landed."""
    ops = []
    for st in range(4):
        for h in (0, 1):
            for qb in (0, 1):  # consecutive ones go to different accumulators; the scheduler puts at most one in a gap
                ops.append((("MSUM", st, qb, h), C.NQK + C.PVS * st, C.NG - 4))
    return ops


def exp8_streams():
    """fp8 variant: exp and pack of S_old, ONE ordered stream per q-block (key block 0 first: key block 1 packs into
    registers 4..7 of the key-block-0 tile, which must have been consumed); the pack of a 4-score group trails the
    exponentials of the next group; everything of a q-block before its first P V MFMA."""
    streams = []
    for qb in (0, 1):
        dl = C.pv_gap(qb, 0) - 1
        ops, pend = [], None
        for kb in (0, 1):
            for m in range(4):
                for r in range(4 * m, 4 * m + 4):
                    ops.append((("EXP", kb, qb, r), 0, dl))
                if pend is not None:
                    ops += [(("CVT8",) + pend + (0,), 0, dl), (("CVT8",) + pend + (1,), 0, dl)]
                pend = (kb, qb, m)
        ops += [(("CVT8",) + pend + (0,), 0, dl), (("CVT8",) + pend + (1,), 0, dl)]
        streams.append(ops)
    return streams


def vread8_stream():
    return [(("VREAD8", db), max(0, C.pv_gap(0, db) - 12), C.pv_gap(0, db) - 4) for db in range(4)]


def bar_gap():
    return C.NQK + 2 * C.PVS  # MIDBAR: behind the PV MFMAs of 16-key steps 0 and 1


def vread_stream(have_new=True):
    ops = []
    for st in range(4):
        for db in range(C.NDB):
            use = C.NQK + st * C.PVS + db * 2
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
    if C.f8:
        sfx, kpos, vpos = "_F8", [1, 5], [9, 13]
    if C.d64:
        sfx, kpos, vpos = "_D64", [1, 9], [5, 13]
    if os.environ.get("W64_DMA_K" + sfx):
        kpos = [int(x) for x in os.environ["W64_DMA_K" + sfx].split(",")]
    if os.environ.get("W64_DMA_V" + sfx):
        vpos = [int(x) for x in os.environ["W64_DMA_V" + sfx].split(",")]
    ops = [(("DMAK", j), kpos[j], kpos[j] + 2) for j in range(C.KDMA)] + [(("DMAV", j), vpos[j], vpos[j] + 2) for j in range(C.VDMA)]
    if C.madd:  # the eight 1-KiB pieces of the next listed tile's mask image, in the gaps the K / V pieces leave (W64_DMA_M overrides)
        # (head_dim 64: half the gaps -- the pieces sit in every other gap between the two K and two V pieces)
        mpos = [int(x) for x in os.environ.get("W64_DMA_M_D64" if C.d64 else "W64_DMA_M", "3,7,11,15,18,21,24,27" if C.d64 else "3,7,11,15,19,23,27,31").split(",")]
        ops += [(("DMAM", j), mpos[j], min(C.NG - 2, mpos[j] + 2)) for j in range(8)]
    ops.sort(key=lambda o: o[1])
    return ops


def dma_update_stream(dma_ops):
    """offset updates, UPD_DELAY gaps behind the deadline of their DMA instruction"""
    last = C.NG - 1
    ups = [(("UPDK" if o[0][0] == "DMAK" else "UPDV", o[0][1]), min(last, o[2] + UPD_DELAY), min(last, o[2] + UPD_DELAY + 3)) for o in dma_ops if o[0][0] != "DMAM"]
    return ups


def mask_ops(kb, qb, st, dl):
    """MASK ops of score block (kb, qb) in register order"""
    return [(("MASK", kb, qb, r), st, dl) for r in range(16)]


def ma_slot(kb, g):
    """additive-mask bodies: which of the q-block's MA_DEPTH temporaries holds the mask fragment of score group g of key block kb (groups 4 kb + g run in order)"""
    return (4 * kb + g) % MA_DEPTH


def madd_chain(qb, per_kb, mld_from):
    """Additive-mask bodies: ONE ordered stream per q-block over both key blocks -- the mask fragments are read MA_DEPTH groups ahead of the v_fma_mix that
    consume them (a fragment does not depend on the scores: the read of key block 1's first groups is issued while key block 0 is still being masked), and a
    temporary is rewritten only behind the group that used it.  per_kb[kb] = (earliest, deadline, ops that follow the block's MASK ops); mld_from: earliest gap of a read."""
    groups = [(kb, g) for kb in (0, 1) for g in range(4)]
    last = max(per_kb[kb][1] for kb in (0, 1))
    ops = [(("MLD", kb, qb, g), mld_from, last) for kb, g in groups[:MA_DEPTH]]
    for n, (kb, g) in enumerate(groups):
        st, dl, tail = per_kb[kb]
        ops += [(("MASK", kb, qb, r), st, dl) for r in range(4 * g, 4 * g + 4)]
        if n + MA_DEPTH < len(groups):
            k2, g2 = groups[n + MA_DEPTH]
            ops.append((("MLD", k2, qb, g2), mld_from, last))
        if g == 3:
            ops += tail
    return ops


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
            if masked and not C.madd:
                for qb in (0, 1):
                    ops += mask_ops(kb, qb, ready + qb, last)
            if masked and C.madd and kb == 1:  # (both key blocks at once, behind the NOP: one ordered chain per q-block, the two interleaved group by group)
                ch = [madd_chain(qb, {0: (C.NQK, last, []), 1: (C.NQK, last, [])}, C.NQK) for qb in (0, 1)]
                for a, b in zip(ch[0], ch[1]):
                    ops += [a, b]
            if masked and C.madd and kb == 0:
                continue  # (the row max of key block 0 follows its masking: below, behind key block 1's)
            for r in range(0, 16, 2):
                for qb in (0, 1):
                    for kbm in ((0, 1) if (masked and C.madd) else (kb,)):
                        ops.append((("MAX", kbm, qb, r, kbm == 0 and r < 4), ready + qb, last))
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
    elif C.f8:
        if masked:
            w = {"mask": {0: (9, 14), 1: (17, 21)}, "max": {0: (15, 17), 1: (22, 24)}, "dec": (25, 26), "fma": (27, 35)}
        else:
            w = {"max": {0: (9, 16), 1: (17, 24)}, "dec": (25, 26), "fma": (27, 35)}
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
                streams.append(mask_ops(kb, qb, max(w["mask"][kb][0], ready + qb), w["mask"][kb][1]))
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


def lazy_streams(masked):
    """Steady-state tiles of the LAZY reference mode (every body but the fp8 one; thresholds per P format in the includer): no row max at all.  e = s*c - m against the reference
    of the previous tiles as soon as a score block's MFMAs are done; whether the reference has to move is read off the
    row sums the matrix pipe delivers anyway (includer: W64_LAZY_CHECK on l after the body -- bf16 P and the fp32
    accumulators have fp32's exponent range, so a stale reference costs no accuracy until l nears 2^100; the includer
    rebases by an exact power of two long before, and a segment that still overflows is re-run with the max chain)."""
    last = C.NG - 1
    streams = []
    if masked and C.madd:
        for qb in (0, 1):
            per = {kb: (kb * C.HALF + C.HALF + qb, last, [(("FMA", kb, qb, r), kb * C.HALF + C.HALF + qb, last) for r in range(16)]) for kb in (0, 1)}
            streams.append(madd_chain(qb, per, max(0, C.HALF - 6)))
        return streams
    for kb in (0, 1):
        ready = kb * C.HALF + C.HALF
        for qb in (0, 1):
            st, dl = ready + qb, last
            ops = []
            if masked:
                ops += mask_ops(kb, qb, st, dl)
            ops += [(("FMA", kb, qb, r), st, dl) for r in range(16)]
            streams.append(ops)
    return streams


def spec_streams(masked):
    """Steady-state tiles, speculative order: the reference max m of the PREVIOUS tiles is known when the tile starts, so
    e = s*c - m is applied to a score block as soon as its MFMAs are done (no wait for this tile's row max), the row max
    is taken of e (monotone in s), and the per-tile decision shrinks to "any e above tau?" (W64_DECIDE2).  Only when it
    fires (rare) the includer moves the reference and shifts the stored e by the difference (fix_e_*).  The old order
    (max of the raw scores -> full decision -> fma) put a ~25-instruction dependent chain plus all 64 fmas behind the
    last score MFMA: 40-56 us of 356-432 at B1 H16 S8192 in the int8 kernels (timing-only ablations, lab notes r2).
    One ordered stream per score block: (int8 convert ->) (mask ->) fma of a register pair, its max one pair later."""
    last = C.NG - 1
    dec_at = last - 1
    streams = [[(("MXINIT",), 0, max(1, C.HALF - 2))]]
    if masked and C.madd:
        for qb in (0, 1):
            per = {}
            for kb in (0, 1):
                st, dl = kb * C.HALF + C.HALF + qb, dec_at - 1
                tail, pend = [], None
                for r in range(0, 16, 2):
                    tail += [(("FMA", kb, qb, r), st, dl), (("FMA", kb, qb, r + 1), st, dl)]
                    if pend is not None:
                        tail.append((("MAXE", kb, qb, pend), st, dl))
                    pend = r
                tail.append((("MAXE", kb, qb, pend), st, dl))
                per[kb] = (st, dl, tail)
            streams.append(madd_chain(qb, per, max(0, C.HALF - 6)))
        streams.append([(("DEC2",), dec_at, dec_at)])
        return streams
    for kb in (0, 1):
        ready = kb * C.HALF + C.HALF
        for qb in (0, 1):
            st, dl = ready + qb, dec_at - 1
            ops = []
            if C.i2f:
                ops += [(("I2F", kb, qb, r), st, dl) for r in range(16)]
            if masked:
                ops += mask_ops(kb, qb, st, dl)
            pend = None
            for r in range(0, 16, 2):
                ops += [(("FMA", kb, qb, r), st, dl), (("FMA", kb, qb, r + 1), st, dl)]
                if pend is not None:
                    ops.append((("MAXE", kb, qb, pend), st, dl))
                pend = r
            ops.append((("MAXE", kb, qb, pend), st, dl))
            streams.append(ops)
    streams.append([(("DEC2",), dec_at, dec_at)])
    return streams


GAP_CAP = {"MSUM": 1}  # at most this many ops of a kind in one gap (a second row-sum MFMA would queue behind the first in the matrix pipe)


def schedule(streams, gaps, pre_budget=0, budget=None):
    """EDF under a per-gap budget.  streams: list of [ (op, earliest, deadline) ... ] each consumed in order.
    Returns gaps + 1 lists: the first is the pre-slot (fillers ahead of MFMA 0, budget pre_budget, ops runnable at gap 0)."""
    gap_budget = budget if budget is not None else C.budget
    BUDGET = gap_budget
    pos = [0] * len(streams)
    out = [[] for _ in range(gaps + 1)]
    # a stream is consumed in order, so an op inherits the tightest deadline of everything queued behind it
    tight = []
    for st_ in streams:
        t = list(st_)
        for k in range(len(t) - 2, -1, -1):
            if t[k + 1][2] < t[k][2]:
                t[k] = (t[k][0], t[k][1], t[k + 1][2])
        tight.append(t)
    streams = tight
    for slot in range(gaps + 1):
        g = max(0, slot - 1)          # slot 0 = the pre-slot: what may run at gap 0, nothing forced
        BUDGET = pre_budget if slot == 0 else gap_budget
        if slot == 0 and pre_budget <= 0:
            continue
        used = 0
        kinds = {}
        while True:
            best = None
            for si, s in enumerate(streams):
                if pos[si] >= len(s):
                    continue
                op, earliest, deadline = s[pos[si]]
                if earliest > g or (slot == 0 and op[0] not in ("EXP", "CVT", "CVT8", "ADD", "MXINIT")):
                    continue
                if kinds.get(op[0], 0) >= GAP_CAP.get(op[0], 1 << 30) and not (deadline <= g and slot > 0):
                    continue
                remaining = sum(COST[o[0][0]] for o in s[pos[si]:])
                # how far this stream is behind an even spread up to its last deadline
                last_dl = min(gaps - 1, max(d for _, _, d in s[pos[si]:]))
                need_rate = remaining / max(1, (last_dl - g + 1))
                forced = deadline <= g and slot > 0
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
            out[slot].append(op)
            used += c
            kinds[op[0]] = kinds.get(op[0], 0) + 1
            pos[si] += 1
    for si, s in enumerate(streams):
        assert pos[si] == len(s), f"stream {si} not fully placed ({pos[si]}/{len(s)})"
    return out


def check_part(placed, have_new, have_old, masked, pre=(), lazy=False):
    """Data-flow self-check of one scheduled part (gap g = after MFMA g): every consumer sits behind its producer.
    (A schedule that packed a P fragment one gap late shows up on the GPU as garbage in exactly the O^T blocks whose
    MFMAs came first - cost a long bisect once.)"""
    pos = {}
    for k, op in enumerate(pre):
        pos[op] = (-1, k)
    for g, ops in enumerate(placed):
        for k, op in enumerate(ops):
            pos[op] = (g, k)
    def before(a, b):
        return a in pos and b in pos and pos[a] < pos[b]
    if have_old and C.f8:
        for qb in (0, 1):
            first_use = C.pv_gap(qb, 0)
            for kb in (0, 1):
                for m in range(4):
                    lo, hi = ("CVT8", kb, qb, m, 0), ("CVT8", kb, qb, m, 1)
                    for r in range(4 * m, 4 * m + 4):
                        assert before(("EXP", kb, qb, r), lo if r < 4 * m + 2 else hi), ("pack before EXP", kb, qb, r)
                    assert before(lo, hi), ("high word packed before the low word (which rewrites the register's low half only)", lo)
                    assert pos[hi][0] < first_use, ("P fragment packed after its first PV MFMA", hi, pos[hi], first_use)
                    # in-place compaction: destination register 4 kb + m of the key-block-0 tile must already have been
                    # consumed as a score: it is score 4 kb + m of key block 0, packed by group (4 kb + m) >> 2
                    dst = 4 * kb + m
                    if not (kb == 0 and m == 0):
                        g = dst >> 2
                        assert before(("CVT8", 0, qb, g, 1 if dst & 2 else 0), lo), ("pack overwrites an unpacked score", lo)
        for db in range(4):
            assert pos[("VREAD8", db)][0] < C.pv_gap(0, db), ("V fragment read after its MFMA", db)
    for kb in (0, 1):
        for qb in (0, 1):
            if have_old and not C.f8:
                for r in range(16):
                    if not C.msum:
                        assert before(("EXP", kb, qb, r), ("ADD", kb, qb, r)), ("ADD before EXP", kb, qb, r)
                for r in range(0, 16, 2):
                    cv = ("CVT", kb, qb, r)
                    assert before(("EXP", kb, qb, r), cv) and before(("EXP", kb, qb, r + 1), cv), ("CVT before EXP", cv)
                    if not C.msum:
                        assert before(("ADD", kb, qb, r), cv) and before(("ADD", kb, qb, r + 1), cv), ("CVT before ADD", cv)
                    else:  # the row-sum MFMA of this packed pair: a later gap than the pack (a big MFMA issues in between), >= 3 MFMA issues before the body ends
                        ms = ("MSUM", 2 * kb + (r >> 3), qb, (r & 7) >> 2)
                        assert pos[cv][0] < pos[ms][0] <= C.NG - 4, ("row-sum MFMA placement", ms, pos[cv], pos[ms])
                    # in-place compaction: pair (r, r+1) lands in register 8*(r>>3) + (r&7)/2 of the tile, which must
                    # already have been consumed as a score (its own ADD and the CVT that read it)
                    dst = 8 * (r >> 3) + ((r & 7) >> 1)
                    if dst not in (r, r + 1):
                        if not C.msum:
                            assert before(("ADD", kb, qb, dst), cv), ("CVT overwrites unread score", cv)
                        assert before(("CVT", kb, qb, dst & ~1), cv), ("CVT overwrites unpacked score", cv)
                    st = 2 * kb + (r >> 3)
                    first_use = C.NQK + C.PVS * st  # first PV MFMA of this 16-key step (any d-block, any q-block)
                    assert pos[cv][0] < first_use, ("P fragment packed after its first PV MFMA", cv, pos[cv], first_use)
            if have_new and lazy:
                last_mfma = kb * C.HALF + (C.HALF - 2) + qb
                for r in range(16):
                    f = ("FMA", kb, qb, r)
                    assert pos[f][0] > last_mfma, ("fma on an unfinished score tile", f)
                    if masked:
                        assert before(("MASK", kb, qb, r), f) and pos[("MASK", kb, qb, r)][0] > last_mfma, ("mask order", kb, qb, r)
                assert not any(o[0] in ("MAX", "MAXE", "DEC", "DEC2", "MXINIT") for o in pos), "lazy body carries a max chain"
            elif have_new and ("DEC2",) in pos:
                last_mfma = kb * C.HALF + (C.HALF - 2) + qb
                for r in range(16):
                    f = ("FMA", kb, qb, r)
                    assert pos[f][0] > last_mfma, ("fma on an unfinished score tile", f)
                    assert before(("MXINIT",), ("MAXE", kb, qb, r & ~1)), ("row max before its -inf start", kb, qb, r)
                    assert before(f, ("MAXE", kb, qb, r & ~1)), ("row max of e before the fma", f)
                    assert before(("MAXE", kb, qb, r & ~1), ("DEC2",)), ("decision before the row max", kb, qb, r)
                    if masked:
                        assert before(("MASK", kb, qb, r), f) and pos[("MASK", kb, qb, r)][0] > last_mfma, ("mask order", kb, qb, r)
                    if C.i2f:
                        assert before(("I2F", kb, qb, r), f) and pos[("I2F", kb, qb, r)][0] > last_mfma, ("convert order", kb, qb, r)
            elif have_new:
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
    if C.madd and masked and have_new:
        for qb in (0, 1):
            groups = [(kb, g) for kb in (0, 1) for g in range(4)]
            for n, (kb, g) in enumerate(groups):
                ld = ("MLD", kb, qb, g)
                for r in range(4 * g, 4 * g + 4):
                    assert before(ld, ("MASK", kb, qb, r)), ("mask fragment read behind its use", ld)
                if n >= MA_DEPTH:  # the temporary's previous tenant must have been consumed
                    kp, gp = groups[n - MA_DEPTH]
                    assert before(("MASK", kp, qb, 4 * gp + 3), ld), ("mask fragment read overwrites an unconsumed one", ld)
    if MIDBAR and have_new:
        bar = pos[("BAR",)]
        for op, at in pos.items():
            if op[0] in ("VREAD", "KREAD", "DMAK", "DMAV"):
                assert at < bar, ("before the barrier", op)
            if op[0] == "KPRE":
                assert at > bar, ("next tile's K fragments are visible only behind the barrier", op)
    if have_old and not C.f8:
        for st in range(4):
            for db in range(C.NDB):
                assert pos[("VREAD", st, db)][0] < C.NQK + st * C.PVS + db * 2, ("V fragment read after its MFMA", st, db)
    if have_new:
        for kb in (0, 1):
            for ks in range(C.KS):
                if ("KREAD", kb, ks) in pos:
                    assert pos[("KREAD", kb, ks)][0] < kb * C.HALF + ks * 2, ("K fragment read after its MFMA", kb, ks)


def emit_part(lines, R, have_new, have_old, masked=False, lazy=False):
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
    if have_old and C.f8:
        for qb in (0, 1):
            for db in range(4):
                mf += [pv8_mfma(R, db, qb), None]   # 64-cycle instruction: two gap slots
            mf += [l8_mfma(R, qb), None]
    elif have_old:
        for st in range(4):
            for db in range(C.NDB):
                for qb in (0, 1):
                    mf.append(pv_mfma(R, st, db, qb))
    else:
        mf += [None] * C.NPV
    streams = []
    if have_old and C.f8:
        streams += exp8_streams()
        streams.append(vread8_stream())
    elif have_old:
        streams += exp_streams()
        streams.append(vread_stream(have_new))
        if C.msum:
            streams.append(msum_stream())
    if have_new:
        if lazy:
            assert have_old and not C.f8
            streams += lazy_streams(masked)
        else:
            streams += spec_streams(masked) if (SPEC and have_old) else start_streams(True, have_old, masked)
        streams.append(kread_stream())
        dmas = dma_stream()
        streams.append(dmas)
        streams.append(dma_update_stream(dmas))
        if MIDBAR:
            streams += mid_barrier_streams(have_old)
    pre_budget = PRE["f8" if C.f8 else "i8" if C.i8 else "16"] if (have_new and have_old) else 0
    budget = None
    if (C.msum or lazy) and have_new and have_old:
        # the filler work of an msum / lazy body is 15-30 % below the budget the other bodies were tuned with: spread it evenly
        # (total / gaps, rounded up to a whole 4-cycle issue slot) instead of front-loading the tile
        total = sum(COST[o[0][0]] for st_ in streams for o in st_)
        budget = 4 * (-(-total // (4 * C.NG)))
        budget = int(os.environ.get("W64_BUDGET_LAZY" if lazy else "W64_BUDGET_MSUM", budget))
    slots = schedule(streams, C.NG, pre_budget, budget)
    pre, placed = slots[0], slots[1:]
    if not ABL:
        check_part(placed, have_new, have_old, masked, pre, lazy)
    for op in pre:
        if op[0] not in ABL:
            lines.append("    " + op_text(R, op))
    if pre:
        lines.append(f"__builtin_amdgcn_sched_barrier(0);  // pre-slot: {sum(COST[o[0]] for o in pre)} cyc of fillers ahead of the first MFMA")
    cyc = 0
    order = list(range(C.NG))
    if ROTATE and have_new and have_old:
        order = order[C.NG - ROTATE:] + order[:C.NG - ROTATE]
    for g in order:
        if GAPSTAMP and have_new and have_old and g in GAPSTAMP:
            lines.append(f"W64_GSTAMP({GAPSTAMP.index(g)});")
        if FSTAMP and have_new and have_old and not masked and g in FSTAMP:
            lines.append(f"W64_FSTAMP({FSTAMP.index(g)});")
        if mf[g] is not None:
            lines.append(mf[g])
        for op in placed[g]:
            if op[0] not in ABL:
                lines.append("    " + op_text(R, op))
        fill = sum(COST[o[0]] for o in placed[g])
        busy = mf[g] is not None or (C.f8 and g >= C.NQK and g > 0 and mf[g - 1] is not None)  # second slot of a 64-cycle MFMA
        cyc += max(32 if busy else 0, (8 if mf[g] else 0) + fill)
        if mf[g] is not None or placed[g]:
            lines.append(f"__builtin_amdgcn_sched_barrier(0);  // gap {g}: filler issue {fill} cyc")
    if FSTAMP and have_new and have_old and not masked:
        lines.append(f"W64_FSTAMP_END({len(FSTAMP)});")
    lines.append(f"// modelled issue time of this part: {cyc} cycles")


def emit_helpers(lines):
    """Helpers that touch the asm-owned O^T registers a[128:255] by literal number."""
    a = lines.append
    a("// GENERATED by tools/gen_w64_body.py -- helpers that address the asm-owned O^T registers a[128:255].")
    a("// The clobbers of v254 / a254 are what makes the kernel descriptor allocate the whole register file (next_free_vgpr = 255")
    a("// -> 256 + 256 after the hardware's granule; tests/test_w64_build_contract.py reads it back from the assembly).  hipcc")
    a("// says \"clobber list contains reserved registers\": under amdgpu_num_vgpr(128) every register from 128 up is reserved FROM")
    a("// THE ALLOCATOR -- which is the point: those registers are the generated stream's, the compiler never names them (the")
    a("// same test checks that too).  The diagnostic is therefore expected here and only here, and switched off here and only here.")
    a("#pragma clang diagnostic push")
    a("#pragma clang diagnostic ignored \"-Winline-asm\"")
    a("__device__ __forceinline__ void zero_o() {")
    a("    asm volatile(" + " ".join(f'"v_accvgpr_write_b32 a{O_BASE + r}, 0\\n\\t"' for r in range(128)) + ' "s_nop 0" ::: "memory", "v254", "a254");')
    a("}")
    a("#pragma clang diagnostic pop")
    a("// ---- matrix-pipe row sums (tools/gen_w64_body.py MS_*): literal v[118:127], kernels compiled with amdgpu_num_vgpr(118)")
    a("__device__ __forceinline__ void ms_init_ones(unsigned bits) {")
    a(f'    asm volatile("v_mov_b32 v{MS_ONES}, %0\\n\\tv_mov_b32 v{MS_ONES + 1}, %0\\n\\ts_nop 1" :: "s"(bits) : "memory");')
    a("}")
    a("__device__ __forceinline__ void ms_zero_l() {")
    a("    asm volatile(" + " ".join(f'"v_mov_b32 v{MS_L + r}, 0\\n\\t"' for r in range(8)) + ' "s_nop 1" ::: "memory");')
    a("}")
    a("// row sums *= alpha (rare paths: deferred max moved / lazy rebase); the four columns of an accumulator are identical and")
    a("// keep accumulating independently, so all are scaled.  Callers sit behind the s_nops that follow the tile's last MFMA.")
    a("__device__ __forceinline__ void ms_scale_l(float a0, float a1) {")
    a("    asm volatile(" + " ".join(f'"v_mul_f32 v{MS_L + r}, v{MS_L + r}, %{r >> 2}\\n\\t"' for r in range(8)) + ' "s_nop 1" :: "v"(a0), "v"(a1) : "memory");')
    a("}")
    a("__device__ __forceinline__ void ms_read_l(float& l0, float& l1) {")
    a(f'    asm volatile("v_mov_b32 %0, v{MS_L}\\n\\tv_mov_b32 %1, v{MS_L + 4}" : "=v"(l0), "=v"(l1));')
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


def emit_helpers_fix(lines):
    """Rare path of the speculative order: the reference max moved, shift the stored e = s*c - m of the NEW score set."""
    a = lines.append
    for setname in ("a", "b"):
        a(f"// e += d (d = old reference - new reference <= 0) on every score register of set {setname.upper()}: q-block 0 with d0, 1 with d1")
        a(f"__device__ __forceinline__ void fix_e_{setname}(float d0, float d1) {{")
        for qb in (0, 1):
            txt = " ".join(f'"v_add_f32 v{base(setname, kb, qb) + r}, v{base(setname, kb, qb) + r}, %0\\n\\t"' for kb in (0, 1) for r in range(16))
            a(f'    asm volatile({txt} "s_nop 0" :: "v"(d{qb}));')
        a("}")


def emit_helpers_f8(lines):
    """Helpers of the fp8 kernel (compiled with amdgpu_num_vgpr(96)): its literal v[96:127] / a[96:127]."""
    a = lines.append
    a("// GENERATED by tools/gen_w64_body.py -- literal registers of fa_fwd_w64_i8f8 (see F8_* there).")
    a("__device__ __forceinline__ void f8_init_consts() {")
    a("    asm volatile(" + " ".join(f'"v_mov_b32 v{F8_BIAS + r}, 0x4b400000\\n\\t"' for r in range(16)) +
      " " + " ".join(f'"v_mov_b32 v{F8_ONES + r}, 0x38383838\\n\\t"' for r in range(8)) +
      f' "v_mov_b32 v{F8_SONE}, 0x7f7f7f7f\\n\\tv_mov_b32 v{F8_VSC}, 0x7f7f7f7f\\n\\ts_nop 1" ::: "memory");')
    a("}")
    a("// fa_fwd_w64_i8 (compiled with amdgpu_num_vgpr(112)): its literal bias tile")
    a("__device__ __forceinline__ void i8_init_bias() {")
    a("    asm volatile(" + " ".join(f'"v_mov_b32 v{I8_BIAS + r}, 0x4b400000\\n\\t"' for r in range(16)) + ' "s_nop 1" ::: "memory");')
    a("}")
    a("__device__ __forceinline__ void f8_set_vscale(unsigned e8) {")
    a(f'    asm volatile("v_mov_b32 v{F8_VSC}, %0\\n\\ts_nop 1" :: "s"(e8) : "memory");')
    a("}")
    a("__device__ __forceinline__ void f8_zero_l() {")
    a("    asm volatile(" + " ".join(f'"v_accvgpr_write_b32 a{F8_L + r}, 0\\n\\t"' for r in range(32)) + ' "s_nop 0" ::: "memory");')
    a("}")
    a("// row sums *= alpha (rare path of the deferred max); only register 0 of each accumulator is ever read back, but all")
    a("// sixteen keep accumulating, so all are scaled")
    a("__device__ __forceinline__ void f8_scale_l(float a0, float a1) {")
    a("    float t0, t1, t2, t3;")
    for qb in (0, 1):
        for r in range(F8_L + 16 * qb, F8_L + 16 * qb + 16, 4):
            a(f'    asm volatile("v_accvgpr_read_b32 %0, a{r}\\n\\tv_accvgpr_read_b32 %1, a{r + 1}\\n\\tv_accvgpr_read_b32 %2, a{r + 2}\\n\\t"'
              f' "v_accvgpr_read_b32 %3, a{r + 3}\\n\\ts_nop 1\\n\\tv_mul_f32 %0, %0, %4\\n\\tv_mul_f32 %1, %1, %4\\n\\tv_mul_f32 %2, %2, %4\\n\\t"'
              f' "v_mul_f32 %3, %3, %4\\n\\ts_nop 1\\n\\tv_accvgpr_write_b32 a{r}, %0\\n\\tv_accvgpr_write_b32 a{r + 1}, %1\\n\\t"'
              f' "v_accvgpr_write_b32 a{r + 2}, %2\\n\\tv_accvgpr_write_b32 a{r + 3}, %3"'
              f' : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(a{qb}));')
    a("}")
    a("__device__ __forceinline__ void f8_read_l(float& l0, float& l1) {")
    a(f'    asm volatile("v_accvgpr_read_b32 %0, a{F8_L}\\n\\tv_accvgpr_read_b32 %1, a{F8_L + 16}" : "=v"(l0), "=v"(l1));')
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
    if not C.f8:  # (fp8 P tops out at 448: no room for a stale reference)
        lines.append("#elif W64_PART == 8  // LAZY reference (no row max), steady state, odd tile")
        emit_part(lines, Roles("b", "a"), True, True, lazy=True)
        lines.append("#elif W64_PART == 9  // LAZY, even tile")
        emit_part(lines, Roles("a", "b"), True, True, lazy=True)
        lines.append("#elif W64_PART == 10  // LAZY, masking tile, odd")
        emit_part(lines, Roles("b", "a"), True, True, masked=True, lazy=True)
        lines.append("#elif W64_PART == 11  // LAZY, masking tile, even")
        emit_part(lines, Roles("a", "b"), True, True, masked=True, lazy=True)
    lines.append("#endif")
    out.write_text("\n".join(lines) + "\n")
    print("wrote", out, len(lines), "lines")


def main():
    global C
    csrc = Path(__file__).resolve().parent.parent / "universal-metal-flash-attention_amd" / "csrc"
    helpers = []
    emit_helpers(helpers)
    emit_helpers_fix(helpers)
    emit_helpers_f8(helpers)
    (csrc / "fa_fwd16_w64_regs.inc").write_text("\n".join(helpers) + "\n")
    C = Cfg(False)
    emit_body(Path(os.environ["W64_OUT"]) if os.environ.get("W64_OUT") else csrc / "fa_fwd16_w64_body.inc")
    C = Cfg(True)
    emit_body(Path(os.environ["W64_OUT_I8"]) if os.environ.get("W64_OUT_I8") else csrc / "fa_fwd_w64_i8_body.inc")
    C = Cfg(True, f8=True)
    emit_body(Path(os.environ["W64_OUT_I8F8"]) if os.environ.get("W64_OUT_I8F8") else csrc / "fa_fwd_w64_i8f8_body.inc")
    C = Cfg(False, d64=True)
    emit_body(Path(os.environ["W64_OUT_D64"]) if os.environ.get("W64_OUT_D64") else csrc / "fa_fwd16_w64d64_body.inc")
    C = Cfg(False, madd=True)
    emit_body(Path(os.environ["W64_OUT_BIAS"]) if os.environ.get("W64_OUT_BIAS") else csrc / "fa_fwd16_w64_bias_body.inc")
    C = Cfg(False, d64=True, madd=True)
    emit_body(Path(os.environ["W64_OUT_BIAS_D64"]) if os.environ.get("W64_OUT_BIAS_D64") else csrc / "fa_fwd16_w64d64_bias_body.inc")


if __name__ == "__main__":
    sys.exit(main())
