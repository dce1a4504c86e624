// fa_fwd16_w64.hip -- bf16 / fp16 forward at head_dim 128, "one wave per SIMD" structure (no mask; causal or not;
// Sq >= 256, Skv >= 64); every other case stays on fa_fwd16 (fa_fwd_16.hip).
//
// Replaces the same Metal `attention` forward dispatch as fa_fwd_16.hip (MFABridge.swift:2240-2248, 2525-2541).
//
// Why a second structure: in fa_fwd16 a wave owns 32 query rows, so every 16 MFMAs it re-reads a whole K and V tile
// from LDS and its softmax VALU work runs back to back with its MFMAs (measured: time ~ MFMA + VALU, ~45 % MFMA
// busy).  Here (cdna_hip_programming.md, "4-wave, one-wave-per-SIMD, persistent structure"):
//   * workgroup = 4 waves = 256 query rows; a wave owns 64 rows (two 32-row q-blocks) and the whole 512-register
//     file: O^T (64 x 128 fp32), Q^T and the current K tile's fragments live in accumulator registers that only
//     inline-asm MFMAs / ds_reads touch; the compiler allocates the arch VGPRs (scores, P, V^T fragments).
//   * per 64-key tile one pass of 64 MFMAs: S(i) = K(i) Q^T, then O^T += V(i-1)^T P(i-1)^T; the softmax of tile
//     i-1, the transposed V reads, the K fragment reads of tile i+1 and the row max of tile i are placed in the
//     issue gaps between those MFMAs by tools/gen_w64_body.py (fa_fwd16_w64_body.inc).
//   * deferred max (T13): the reference max moves only when a row max exceeds it by > 2^6; O (in AGPRs) is then
//     rescaled by a rare v_accvgpr_read/mul/write pass after the pending tile's PV.
//   * K/V tiles arrive by LDS-DMA into 2-slot rings (K three tiles ahead, V one), one barrier per tile.
//   * persistent grid (one workgroup per CU): whole rounds of items first (lockstep per XCD, K/V read into its L2
//     once), then the remaining items are shared "stream-K" style, so 384 items on 256 CUs (the FLUX shape) cost
//     1.5 item-times instead of 2; an item cut by a slice boundary is folded by the last of its parts to arrive,
//     in index order (bitwise reproducible), through part_buf.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "fa_fwd16_w64_params.h"

namespace umfa {

#define W64_I8 0
#define W64_BODY_INC "fa_fwd16_w64_body.inc"
#define W64_T __bf16
#define W64_MFMA "v_mfma_f32_32x32x16_bf16"
#define W64_MFMA_QK "v_mfma_f32_32x32x16_bf16"
#define W64_MSUM "v_mfma_f32_4x4x4_16b_bf16"   /* row sums: lane-local sum of four P values against an all-ones operand */
#define W64_ONES_BITS 0x3f803f80u              /* bf16 1.0 twice */
#define W64_LAZY_PARTS 1                        /* lazy reference mode, bf16 P: fp32's exponent range (2 = fp16 P: tighter thresholds) */
#define W64_CVT "v_cvt_pk_bf16_f32"
#define W64_KERNEL fa_fwd16_w64_bf16
#include "fa_fwd16_w64_kernel.inc"
#undef W64_T
#undef W64_MFMA
#undef W64_MFMA_QK
#undef W64_MSUM
#undef W64_ONES_BITS
#undef W64_LAZY_PARTS
#undef W64_CVT
#undef W64_KERNEL

// bf16 Q / K / V with the P V product in fp16 -- the DEFAULT bf16 forward (option pv_fp16): S = K Q^T on the bf16 MFMA, P rounded to fp16 (v_cvt_pk_f16_f32: 11 bits instead of bf16's 8, which
// is what puts the bf16-input forward inside the north-star's 1e-3), O^T += V^T P^T on the fp16 MFMA against an fp16 image of V
// (the runtime's cast pre-pass, fa_aux.hip: V * 2^-e with one power of two per (batch, head) slab, so that no bf16 value leaves fp16's
// range; 2^e comes back in the epilogue's 1 / l: W64_VSC); the lazy reference with fp16's thresholds.
// (Round 4 also built the conversion INSIDE this kernel -- register-staged V tiles, 48 vector instructions + 4 ds_write per tile
// per wave: +23 % cycles per tile, +15 % loop time at the FLUX shape (in-kernel stamps 167.6 vs 145.3 us,
// profiles/r4/lab_notes.md): every workgroup converts every V tile again, 16 x redundantly at FLUX; the pre-pass converts once
// at HBM speed.  An epilogue check of the outputs was dropped too: two more live registers there cost the TILE LOOP 76
// accumulator-register moves and the kernel 8-13 %.)
#define W64_T __bf16
#define W64_MFMA "v_mfma_f32_32x32x16_f16"
#define W64_MFMA_QK "v_mfma_f32_32x32x16_bf16"
#define W64_MSUM "v_mfma_f32_4x4x4_16b_f16"
#define W64_ONES_BITS 0x3c003c00u
#define W64_LAZY_PARTS 2
#define W64_CVT "v_cvt_pk_f16_f32"
#define W64_KERNEL fa_fwd16_w64_bf16pv16
#undef W64_VSC
#define W64_VSC 1
#include "fa_fwd16_w64_kernel.inc"
#undef W64_VSC
#define W64_VSC 0
#undef W64_T
#undef W64_MFMA
#undef W64_MFMA_QK
#undef W64_MSUM
#undef W64_ONES_BITS
#undef W64_LAZY_PARTS
#undef W64_CVT
#undef W64_KERNEL

#define W64_T _Float16
#define W64_MFMA "v_mfma_f32_32x32x16_f16"
#define W64_MFMA_QK "v_mfma_f32_32x32x16_f16"
#define W64_MSUM "v_mfma_f32_4x4x4_16b_f16"
#define W64_ONES_BITS 0x3c003c00u              /* fp16 1.0 twice */
#define W64_LAZY_PARTS 2                        /* fp16 P: the lazy mode with fp16's thresholds (see the kernel) */
#define W64_CVT "v_cvt_pk_f16_f32"
#define W64_KERNEL fa_fwd16_w64_f16
#include "fa_fwd16_w64_kernel.inc"
#undef W64_T
#undef W64_MFMA
#undef W64_CVT
#undef W64_KERNEL
#undef W64_MFMA_QK
#undef W64_BODY_INC

// head_dim 64: the same structure (tools/gen_w64_body.py Cfg(d64=True)): 32 MFMAs per 64-key tile, rows of 128 bytes in the
// tile images, d-blocks 0 and 1 of head_dim 128's O^T register map
#undef W64_DP  /* (the kernel text defaults it to 128) */
#define W64_DP 64
#define W64_BODY_INC "fa_fwd16_w64d64_body.inc"
#undef W64_MSUM
#undef W64_ONES_BITS
#undef W64_LAZY_PARTS
#define W64_T __bf16
#define W64_MFMA "v_mfma_f32_32x32x16_bf16"
#define W64_MFMA_QK "v_mfma_f32_32x32x16_bf16"
#define W64_MSUM "v_mfma_f32_4x4x4_16b_bf16"
#define W64_ONES_BITS 0x3f803f80u
#define W64_LAZY_PARTS 1
#define W64_CVT "v_cvt_pk_bf16_f32"
#define W64_KERNEL fa_fwd16_w64d64_bf16
#include "fa_fwd16_w64_kernel.inc"
#undef W64_T
#undef W64_MFMA
#undef W64_MFMA_QK
#undef W64_MSUM
#undef W64_ONES_BITS
#undef W64_LAZY_PARTS
#undef W64_CVT
#undef W64_KERNEL
/* bf16 operands, fp16 P V against the fp16 image of V: the default bf16 forward at head_dim 64 */
#define W64_T __bf16
#define W64_MFMA "v_mfma_f32_32x32x16_f16"
#define W64_MFMA_QK "v_mfma_f32_32x32x16_bf16"
#define W64_MSUM "v_mfma_f32_4x4x4_16b_f16"
#define W64_ONES_BITS 0x3c003c00u
#define W64_LAZY_PARTS 2
#define W64_CVT "v_cvt_pk_f16_f32"
#define W64_KERNEL fa_fwd16_w64d64_bf16pv16
#undef W64_VSC
#define W64_VSC 1
#include "fa_fwd16_w64_kernel.inc"
#undef W64_VSC
#define W64_VSC 0
#undef W64_T
#undef W64_MFMA
#undef W64_MFMA_QK
#undef W64_MSUM
#undef W64_ONES_BITS
#undef W64_LAZY_PARTS
#undef W64_CVT
#undef W64_KERNEL
#define W64_T _Float16
#define W64_MFMA "v_mfma_f32_32x32x16_f16"
#define W64_MFMA_QK "v_mfma_f32_32x32x16_f16"
#define W64_MSUM "v_mfma_f32_4x4x4_16b_f16"
#define W64_ONES_BITS 0x3c003c00u
#define W64_LAZY_PARTS 2
#define W64_CVT "v_cvt_pk_f16_f32"
#define W64_KERNEL fa_fwd16_w64d64_f16
#include "fa_fwd16_w64_kernel.inc"
#undef W64_T
#undef W64_MFMA
#undef W64_MFMA_QK
#undef W64_CVT
#undef W64_KERNEL
#undef W64_BODY_INC
#undef W64_DP
#undef W64_I8
// (W64_MSUM / W64_ONES_BITS / W64_LAZY_PARTS of the fp16 family stay defined for the int8 kernels below: fp16 P there too)

// runtime-quantised variant: int8 QK^T, fp16 PV
#define W64_I8 1
#undef W64_VSC
#define W64_VSC 2                               /* the V image's 2^e per dense (batch, head) slab, nullable */
#define W64_BODY_INC "fa_fwd_w64_i8_body.inc"
#define W64_T _Float16
#define W64_MFMA "v_mfma_f32_32x32x16_f16"
#define W64_MFMA_QK "v_mfma_i32_32x32x32_i8"
#define W64_CVT "v_cvt_pk_f16_f32"
#define W64_KERNEL fa_fwd_w64_i8
#include "fa_fwd16_w64_kernel.inc"
#undef W64_T
#undef W64_MFMA
#undef W64_MFMA_QK
#undef W64_CVT
#undef W64_KERNEL
#undef W64_BODY_INC

// runtime-quantised, fp8 P V (opt-in fast mode, quant_mode 3): int8 QK^T, fp8 e4m3 P and V
#undef W64_VSC
#define W64_VSC 0                               /* (the fp8 image carries a power of two per TILE in the MFMA's scale operand) */
#undef W64_F8
#define W64_F8 1
#undef W64_LAZY_PARTS
#define W64_LAZY_PARTS 0                        /* fp8 P tops out at 448: deferred max only */
#define W64_BODY_INC "fa_fwd_w64_i8f8_body.inc"
#define W64_T _Float16
#define W64_MFMA "v_mfma_f32_32x32x16_f16"
#define W64_MFMA_QK "v_mfma_i32_32x32x32_i8"
#define W64_CVT "v_cvt_pk_f16_f32"
#define W64_KERNEL fa_fwd_w64_i8f8
#include "fa_fwd16_w64_kernel.inc"
#undef W64_T
#undef W64_MFMA
#undef W64_MFMA_QK
#undef W64_CVT
#undef W64_KERNEL
#undef W64_F8
#undef W64_I8
#undef W64_BODY_INC


// Softmax reference policy of a launch (kernels.h SoftmaxRef, set by umfa_set_option): tau of the max-chain tile bodies and
// whether the bf16 kernels run their lazy bodies.  Measured accuracy / time of the regimes: DESIGN.md §3.2.
#ifndef W64_LAZY_FP16_DEFAULT
#define W64_LAZY_FP16_DEFAULT 1
#endif
static void w64_softmax_policy(int in_prec, float* tau, uint32_t* lazy) {
    const int mode = tuning().sm_mode.load(std::memory_order_relaxed);
    const float t = tuning().sm_tau.load(std::memory_order_relaxed);
    *tau = mode == SM_EXACT ? 0.0f : t;
    // bf16 P: lazy by default.  fp16 P (fp16 kernels, the int8 kernel): the lazy bodies exist with fp16's thresholds
    *lazy = (mode == SM_LAZY || (mode == SM_DEFAULT && (in_prec == P_BF16 || W64_LAZY_FP16_DEFAULT))) ? 1u : 0u;
}

static int w64_cu_count() { return device_cu_count(); }

// Sliding window without a mask tensor (MK_WINDOW; with `causal` the right edge is the diagonal): the WINDOW instantiations
// sweep, per 256-row block, only the key tiles of its band -- w64_window_tiles of them, the same count for every block.
static bool w64_is_window(const FwdParams& p) { return p.mask_kind == MK_WINDOW; }
// (clamped to values that change nothing: key >= row - left holds for every row once left >= Sq, key <= row + right once
// right >= Skv; the kernel adds them to 32-bit row / key differences)
static uint32_t w64_win_left(const FwdParams& p) { return p.win_left < p.Sq ? p.win_left : p.Sq; }
static uint32_t w64_win_right(const FwdParams& p) { return p.causal ? 0u : (p.win_right < p.Skv ? p.win_right : p.Skv); }
static uint32_t w64_tiles_per_item(const FwdParams& p) {
    const uint32_t T = (p.Skv + 63) / 64;
    if (!w64_is_window(p)) return T;
    // key tiles the band of a 256-row block touches: blocks start on tile boundaries (q0 = 256 j), so the band [q0 - left, q0 + 255 + right]
    // runs from tile q0 / 64 - ceil(left / 64) to tile q0 / 64 + floor((255 + right) / 64) -- 20 tiles for +-512, not the 21 of "band / 64 + 1"
    const uint64_t tw = ((uint64_t)w64_win_left(p) + 63) / 64 + (255ull + w64_win_right(p)) / 64 + 1;
    return tw < T ? (uint32_t)tw : T;
}

static uint32_t w64_grid(const FwdParams& p);

// ---- routing cost model ------------------------------------------------------------------------------------------------------------
// Predicted microseconds of one launch, per structure.  The FORMS follow the kernels' schedules; the constants are least-squares fits
// (log time) to launches measured on MI355X at 256 CUs, both kernels forced in turn (tools/fit_route_model.py prints this table from
// profiles/r6/routing_*.jsonl together with its residuals: median error 4-9 %, which is why the comparison below is only trusted as a
// comparison -- both predictions share most of their inputs).  Other CU counts enter through the schedule arithmetic (rounds, slices),
// not through the constants.
//   w64 (one persistent workgroup per CU):  t0 + cast + segments c_seg + tile_steps t_step + folded_parts c_fold
//   128-row (R co-resident workgroups per CU, dispatched as slots free up):
//        uniform items   full rounds of R at tau1 f(R) per tile, the rest at f(rest)          + c_item per round (+ c_tail per split part)
//        causal items    max(throughput of all tiles at f(R) / R  + c_tail longest, the longest item alone)
//   cast = cast_a + 2 V bytes / cast_tbps for bf16 operands (the fp16 image of V: every w64 launch; 128-row launches from 16 MB on)
struct W64Cost { float t0, c_seg, t_step, c_fold, cast_a, cast_tbps; };
struct R128Cost { float t0, c_item, tau1, f2, f3, cast_a, cast_tbps, c_tail, g16; };  // g16: tau1 factor without the bf16 -> fp16 conversion of V (fp16 operands)
// [head_dim 64 | 128][full | causal]
static const W64Cost kW64Cost[2][2] = {
    {{1.9654f, 5.6596f, 1.0164f, 1.8670f, 4.8672f, 4.1271f}, {9.3795f, 0.0000f, 1.1237f, 48.8617f, 2.7924f, 2.2478f}},
    {{0.4243f, 10.5808f, 1.5482f, 3.2266f, 3.4018f, 3.8643f}, {0.0000f, 6.7763f, 1.4334f, 52.2072f, 2.5095f, 2.6358f}}};
static const R128Cost kR128Cost[2][2] = {
    {{1.9627f, 3.9851f, 0.6098f, 2.0000f, 1.0000f, 0.0000f, 20.0000f, 1.3083f, 0.9054f}, {0.2449f, 5.9446f, 0.7894f, 1.4608f, 1.0000f, 0.0000f, 20.0000f, 0.6456f, 0.8727f}},
    {{0.0000f, 8.3967f, 1.0624f, 2.0000f, 2.8549f, 5.1811f, 20.0000f, 3.2320f, 1.0191f}, {0.0000f, 8.4229f, 1.2922f, 1.7176f, 1.0000f, 0.0000f, 12.7398f, 0.4806f, 0.8916f}}};
// balanced causal pairs (128-row kernel, round 6): microseconds of a round with one / two workgroups per CU as a + b h, h = key tiles per workgroup (half a pair)
struct CbalCost { float a1, b1, a2, b2; };
static const CbalCost kCbalCost[2] = {{8.5243f, 1.1660f, 11.3665f, 1.3999f}, {11.7677f, 1.5652f, 21.1185f, 1.7844f}};
// (fit of round 6, 1420 launches of profiles/r6/routing_random_*.jsonl -- the 128-row kernel changed: balanced causal pairs, the split-KV fold's read-ahead: each
// model's error against the measurement: median 4-9 %, 90th percentile 11-24 %, the paired launches 1.5-3 % / 5-10 %; routing by the pair: within 5 % of the faster
// kernel on 97.8 % of the launches, worst 1.17 x -- the round-5 constants, measured on the same launches: 90.6 %, worst 1.33 x)

static double route_v_megabytes(const FwdParams& p) {  // distinct V slabs (broadcast batch / head strides: one slab)
    return (double)(p.vs[0] == 0 ? 1u : p.B) * (p.vs[1] == 0 ? 1u : p.H) * p.Skv * p.D * 2.0 * 1e-6;
}

double fwd_w64_predict_us(const FwdParams& p) {
    const W64Cost& c = kW64Cost[p.D == 64 ? 0 : 1][p.causal ? 1 : 0];
    const uint64_t nqb = (p.Sq + 255) / 256, items = (uint64_t)p.B * p.H * nqb, T = (p.Skv + 63) / 64;
    const uint64_t G = w64_grid(p);
    double segs, steps, fold = 0.0;
    if (p.causal) {
        const uint64_t jobs = (uint64_t)p.B * p.H * ((nqb + 1) / 2), rounds = (jobs + G - 1) / G;
        uint64_t longest = 0;  // tile steps of the longest mirrored pair (j, nqb - 1 - j)
        for (uint64_t j = 0; j < (nqb + 1) / 2; ++j) {
            const uint64_t a = std::min<uint64_t>(T, 4 * j + 4), m = nqb - 1 - j, b2 = m != j ? std::min<uint64_t>(T, 4 * m + 4) : 0;
            longest = std::max(longest, a + b2);
        }
        segs = 2.0 * rounds;
        steps = (double)(rounds * longest);
    } else {
        const uint64_t full = items / G, rem = items % G;
        segs = (double)(full + (rem ? 1 : 0));
        steps = (double)(full * T + (rem ? (rem * T + G - 1) / G : 0));
        if (rem) fold = std::max((double)G / (double)rem - 1.0, 0.0);  // the folding workgroup reads the other parts one after the other
    }
    const double cast = (p.in_prec == P_BF16 && p.pv16) ? c.cast_a + 2.0 * route_v_megabytes(p) / c.cast_tbps : 0.0;
    return c.t0 + cast + segs * c.c_seg + steps * c.t_step + fold * c.c_fold;
}

double fwd_16_predict_us(const FwdParams& p) {
    const R128Cost& c = kR128Cost[p.D == 64 ? 0 : 1][p.causal ? 1 : 0];
    const uint64_t cus = (uint64_t)w64_cu_count();
    const uint64_t nqb = (p.Sq + 127) / 128, items = (uint64_t)p.B * p.H * nqb, T = (p.Skv + 63) / 64;
    const uint32_t R = (p.D == 128 && !p.causal) ? 3u : 2u;  // (32-key tiles at head_dim 128: three resident workgroups)
    const double fr[4] = {0.0, 1.0, c.f2, c.f3};
    const double tau1 = (p.in_prec == P_BF16 && p.pv16) ? c.tau1 : c.tau1 * c.g16;
    const FwdSplitPlan plan = fwd_16_split_plan(p);
    const uint64_t k = plan.nsplit > 1 ? plan.nsplit : 1;
    const double n = (double)(items * k) / (double)cus;
    double body;
    if (plan.cbal) {
        // balanced causal pairs (round 6): every workgroup sweeps half a pair = h tiles; rounds of two workgroups per CU.  Piecewise-linear fit of
        // profiles/r6/cbal_matrix.jsonl (the unpaired form's f2 does not carry over: here BOTH workgroups of a CU are busy all the way)
        const uint64_t lastq = nqb - 1;
        const double h = 0.5 * (double)(std::min<uint64_t>(T, 2) + std::min<uint64_t>(T, (lastq * 128 + 128 + 63) / 64));
        const CbalCost& cb = kCbalCost[p.D == 64 ? 0 : 1];
        const double one = cb.a1 + cb.b1 * h, two = cb.a2 + cb.b2 * h;
        const uint64_t full = (uint64_t)(n / 2.0);
        const double rest = n - 2.0 * (double)full;
        body = ((double)full * two + (rest > 1e-9 ? (rest <= 1.0 ? one : two) : 0.0)) * (p.in_prec == P_BF16 && p.pv16 ? 1.0 : (double)c.g16) - c.t0;  // (the intercepts include the launch)
    } else if (p.causal) {
        uint64_t tot = 0, longest = 0;
        for (uint64_t qb = 0; qb < nqb; ++qb) {
            const uint64_t len = std::min<uint64_t>(T, (qb * 128 + 128 + 63) / 64);
            tot += len;
            longest = std::max(longest, len);
        }
        tot *= (uint64_t)p.B * p.H;
        if (n <= (double)R) {
            const int r = std::max(1, (int)std::ceil(n));
            body = c.c_item + (double)longest * tau1 * fr[r];
        } else {
            const double thr = (double)tot / (double)cus * tau1 * fr[R] / R + n * c.c_item / R;
            body = std::max(thr + c.c_tail * (double)longest * tau1, c.c_item + (double)longest * tau1 * fr[R]);
        }
    } else {
        const double L = (double)T / (double)k;
        const uint64_t full = (uint64_t)(n / R);
        const int rest = (int)std::ceil(n - (double)(full * R) - 1e-9);
        body = (double)full * (c.c_item + L * tau1 * fr[R]) + (rest > 0 ? c.c_item + L * tau1 * fr[rest] : 0.0) + (k > 1 ? c.c_tail * (double)k : 0.0);
    }
    const bool cast_on = p.in_prec == P_BF16 && p.pv16 && route_v_megabytes(p) * 1e6 >= (double)((size_t)16 << 20) && p.Sq >= 1024;  // (runtime.hip's rule)
    const double cast = cast_on ? c.cast_a + 2.0 * route_v_megabytes(p) / c.cast_tbps : 0.0;
    return c.t0 + cast + body;
}

bool fwd_w64_supported(const FwdParams& p) {
    if (tuning().no_w64.load(std::memory_order_relaxed) || !fwd_16_supported(p)) return false;
    if ((p.D != 128 && p.D != 64) || (p.mask_kind != MK_NONE && p.mask_kind != MK_WINDOW && p.mask_kind != MK_BOOL && p.mask_kind != MK_F16 && p.mask_kind != MK_BF16 && p.mask_kind != MK_F32)) return false;
    if (p.mask_kind == MK_F16 || p.mask_kind == MK_BF16 || p.mask_kind == MK_F32) {
        // ADDITIVE fp16 / bf16 mask tensors (MASKA instantiations, round 6; the reference's additive masks: MFABridge.swift:157-242): head_dim 128 and 64, the fp16-P-V
        // families, no causal flag / rotation on top; the wave's mask tile comes by LDS-DMA straight from the caller's tensor, so: keys contiguous,
        // 16-byte aligned rows, Sq and Skv whole 64-row / 64-key tiles; at least one 256-row block per CU (whole blocks in rounds + a shared remainder,
        // as for bool masks).  Everything else (bf16 / fp32 masks, ragged shapes, few blocks) stays on the 128-row kernel.
        if (tuning().no_w64_mask.load(std::memory_order_relaxed) || tuning().no_w64_bias.load(std::memory_order_relaxed) || !p.mask || (p.D != 128 && p.D != 64) || p.rope_cos || p.causal) return false;
        if (p.in_prec == P_BF16 && !p.pv16) return false;
        // whole 64 x 64 tiles -- or (end of round 6) a ragged shape through the classification pass's PADDED fp16 copy (fa_aux.hip mask_classify_body: keys past Skv and rows
        // past Sq at -inf): Sq from 1024 on (a ragged last block wastes its empty waves, as for the unmasked kernels), Skv >= 64, the mask's 16-byte chunks entirely inside or
        // outside it (Skv a multiple of 8 with 16-bit masks, of 4 with fp32 ones)
        const bool ragged = !p.mask_padded && (p.Sq % 64 != 0 || p.Skv % 64 != 0);  // (a padded copy: the pass has dealt with the shape; its chunk rule was the SOURCE's)
        if (p.Sq < 256 || p.Skv < 64 || (p.Sq % 64 != 0 && p.Sq < 1024) || ((p.Skv + 63) / 64) > 1024u) return false;
        // (any Skv, any row alignment since the very end of the round: the pass reads such rows element by element)
        // (a ragged fp16 mask is always read by the pass -- it needs the padded copy --, like every bf16 mask: up to 1 GiB of copy, below)
        if (ragged && tuning().no_w64_ragged_mask.load(std::memory_order_relaxed)) return false;
        if (p.out_prec != P_FP32 && p.out_prec != p.in_prec) return false;
        // contiguous keys; an fp16 mask the kernel reads IN PLACE needs 16-byte aligned rows (mask_needs_copy says when it is not read in place: the copy is aligned by construction)
        if (p.ms[3] != 1) return false;
        if (!mask_needs_copy(p)) {
            if (((uintptr_t)p.mask & 15) != 0 || (p.ms[0] & 7) != 0 || (p.ms[1] & 7) != 0 || (p.ms[2] & 7) != 0) return false;
        } else if (tuning().no_w64_ragged_mask.load(std::memory_order_relaxed) && p.mask_kind == MK_F16) {
            return false;  // (the option keeps fp16 masks that would need the copy on the 128-row kernel)
        }
        if (p.mask_kind != MK_F32 && p.ms[2] != 0 && (uint64_t)p.ms[2] * 2 * 64 > 0x7fffffffull) return false;  // (a wave's 64 rows behind one 32-bit descriptor; fp32: the kernel reads the dense copy)
        // bf16 masks: the classification pass also writes the dense fp16 copy the kernel reads (bf16's significands fit fp16's; fa_aux.hip) -- up to 1 GiB of it
        if (mask_copy_bytes(p) > ((size_t)1 << 30)) return false;  // (0 for an fp16 mask on whole tiles)
        // fp32 masks (end of round 6): the same copy, taken by the bias kernel only when the pass finds it EXACT -- the verdict is a device word, so the call
        // enqueues the 128-row kernel on the caller's tensor as well and the kernels guard themselves (FwdParams::guard).  Only masks the pass may read
        // (its bytes within twice the call's tensor traffic: a dense per-head fp32 bias is read once, by the 128-row kernel, as before).
        if (p.mask_kind == MK_F32) {
            if (tuning().no_w64_f32_mask.load(std::memory_order_relaxed)) return false;
            // the size rule of every mask pre-pass (fa_aux.hip mask_flags_worthwhile: 2 x the call's tensor bytes; 8 x for a mask with a batch and no head dimension);
            // lab option f32_mask_ratio > 0 puts its own constant in its place
            const int ratio = tuning().f32_mask_ratio.load(std::memory_order_relaxed);
            if (ratio > 0) {
                const uint64_t Bm = p.ms[0] != 0 ? p.B : 1, Hm = p.ms[1] != 0 ? p.H : 1, eb = 2;
                const uint64_t mask_bytes = Bm * Hm * p.Sq * p.Skv * 4, qkvo = (uint64_t)p.B * p.H * p.D * ((uint64_t)p.Sq * (eb + 4) + 2ull * p.Skv * eb);
                if (mask_bytes > (uint64_t)ratio * qkvo) return false;
            } else if (!mask_flags_worthwhile(p)) {
                return false;
            }
        }
        if (w64_grid(p) > 512u) return false;
        if (tuning().force_w64.load(std::memory_order_relaxed)) return true;
        const uint64_t blocks = (uint64_t)p.B * p.H * ((p.Sq + 255) / 256), cus = (uint64_t)w64_cu_count();
        return blocks >= cus;
    }
    if (p.mask_kind == MK_BOOL) {
        // bool mask tensors (MASKT instantiations): head_dim 128 and (round 5) 64, the fp16-P-V families (bf16 operands by default, fp16 operands), no
        // rotation on top (a causal flag is folded into the packed mask); whole items per workgroup, so at least one item per CU
        if (tuning().no_w64_mask.load(std::memory_order_relaxed) || !p.mask || (p.D != 128 && p.D != 64) || p.rope_cos) return false;  // (a causal flag is folded into the packed bits)
        if (p.in_prec == P_BF16 && !p.pv16) return false;
        if (p.Skv < 64 || p.Sq < 256 || (p.Sq % 256 != 0 && p.Sq < 1024) || ((p.Skv + 63) / 64) > 1024u) return false;  // (a block's tile list sits in 4 KiB of LDS)
        if (p.out_prec != P_FP32 && p.out_prec != p.in_prec) return false;
        // at least one block per CU; or -- fewer blocks, every block then shared between workgroups (w64_grid) -- a mask WITHOUT a row
        // dimension (key padding, [B, 1 | H, 1, Skv]) and the unmasked kernel's 10 tile steps per CU.  Measured with fewer blocks than CUs
        // (profiles/r4/mask_w64_few_blocks.jsonl, this kernel / 128-row kernel): padding 1.15-1.49 x, dense random per-head 1.7-2.4 x, but
        // block-diagonal and window TENSORS 0.75-0.96 x (short lists: two parts + a fold per block) -- and how dense a [Sq, Skv] mask is
        // the host cannot know without reading it back
        const uint64_t blocks = (uint64_t)p.B * p.H * ((p.Sq + 255) / 256), cus = (uint64_t)w64_cu_count();
        if (w64_grid(p) > 512u) return false;  // (the pre-pass and the kernel keep the shared blocks' running sums in 2 KiB of LDS: fa_aux.hip mask_pack_prepare refuses more)
        if (tuning().force_w64.load(std::memory_order_relaxed)) return true;
        // causal + mask (the causal flag folded into the packed bits: fa_aux.hip causal_word) runs here when asked for, not by default: measured
        // level with the 128-row kernel (profiles/r5/causal_mask_timing.txt: 0.93-1.06 x at FLUX size, B4 H16 S4096, B2 H16 S8192 with key
        // padding / four documents) -- the lists of a causal launch are as unequal as its blocks, and whole blocks per workgroup balance them no
        // better than the 128-row kernel's dispatch order does
        if (p.causal) return false;
        if (p.in_prec == P_BF16 && p.Sq < 1024) return false;  // (the fp16 image of V is re-read by too few q-blocks: see below)
        // (1024 <= Sq < 2048, masks shared by every (batch, head) or without a row dimension: B4 H12 S1536 window 78 us against 62, B4 H8 S1536 padding 72 / 65,
        // B1 H64 S1024 window 48 / 44 -- the pass and the per-block prologues against short lists)
        if (p.in_prec == P_BF16 && p.Sq < 2048 && ((p.ms[0] == 0 && p.ms[1] == 0) || p.ms[2] == 0)) return false;
        // (random-size audit, routing_random_masks_*.jsonl: with WHOLE blocks on `blocks` workgroups -- nothing cut, some CUs idle -- DENSE masks with a row
        // dimension win here too from three eighths of a block per CU (random per-head masks B4 H6 S1536 68 us against 112, B2 H4 S4096 139 / 162), sparse
        // structured ones lose as much (block-diagonal, 8 documents, B4 H6 S1536 47 / 30).  The host cannot see the density; it can see the shape: a mask with
        // a batch or head dimension of its own is not a window or a document mask, one shared by every (batch, head) usually is)
        return blocks >= cus || (p.ms[2] == 0 && blocks * ((p.Skv + 63) / 64) >= cus * 10) || ((p.ms[0] != 0 || p.ms[1] != 0) && blocks * 8 >= cus * 3);
    }
    if (w64_is_window(p) && p.rope_cos) return false;  // window instantiations: no fused rotation
    if (p.D == 64 && p.rope_cos) return false;  // the fused Q rotation exists at head_dim 128 only
    // rows are processed in blocks of 256: a ragged last block wastes its empty waves, so small ragged Sq stay on
    // the 128-row kernel; any Skv >= 64 works (a partial last key tile runs the masking variant of the tile body)
    if (p.Skv < 64 || p.Sq < 256 || (p.Sq % 256 != 0 && p.Sq < 1024)) return false;
    if (p.out_prec != P_FP32 && p.out_prec != p.in_prec) return false;
    // Which of the two structures?  A cost model (below: fwd_w64_predict_us / fwd_16_predict_us), not a rule per regime: predicted
    // microseconds of THIS launch on each kernel from its schedule -- rounds, tile steps per workgroup, segments, folds, the V cast pass --
    // with per-(head_dim, causal) constants fitted to ~1100 measured random launches (tools/fit_route_model.py, profiles/r5/routing_*.jsonl).
    // Rounds 3-4 had ~35 thresholds here, each fitted to the regime in which it was found; on the same records the model's choice is
    // within 5 % of the faster kernel more often than theirs was (the fit script prints both).  force_w64 lifts it (parity tests on
    // small shapes).
    if (!tuning().force_w64.load(std::memory_order_relaxed)) {
        if (w64_is_window(p)) {
            // windows: the band's tile steps are what there is to share; no measurements of both structures over random windows exist, so
            // the one rule of round 3 stays (cut items need 10 steps per CU)
            const uint64_t cus = (uint64_t)w64_cu_count(), nqb = (p.Sq + 255) / 256;
            const uint64_t items = (uint64_t)p.B * p.H * nqb, steps = items * w64_tiles_per_item(p);
            if (items % cus != 0 && steps < cus * 10) return false;
        } else if (!(fwd_w64_predict_us(p) < fwd_16_predict_us(p))) {
            return false;
        }
    }
    return true;
}

static uint32_t w64_grid(const FwdParams& p) {
    if (p.mask_kind == MK_BOOL || p.mask_kind == MK_F16 || p.mask_kind == MK_BF16) {  // mask tensors: one workgroup per CU at most; whole blocks in rounds, the last n % grid blocks cut along their tile lists
        const uint64_t items = (uint64_t)p.B * p.H * ((p.Sq + 255) / 256);
        const uint64_t cus = (uint64_t)w64_cu_count();
        if (const int gi = tuning().w64_grid.load(std::memory_order_relaxed)) {  // lab / tests: force the number of workgroups
            if (gi > 0 && (uint64_t)gi <= cus && (uint64_t)gi <= items && gi <= 512) return (uint32_t)gi;
        }
        const uint64_t gmax = cus < 512 ? cus : 512;  // (the shared blocks' running sums sit in 2 KiB of LDS)
        if (items < cus) return (uint32_t)((p.ms[2] == 0 && items * ((p.Skv + 63) / 64) >= cus * 10) ? gmax : items);  // few blocks, key-padding mask: every block shared
        return (uint32_t)gmax;
    }
    const bool pairs = p.causal && !w64_is_window(p);  // (a causal window is a window with right = 0: linear schedule)
    uint64_t total = (uint64_t)p.B * p.H * ((p.Sq + 255) / 256) * w64_tiles_per_item(p);  // (item, key tile) steps
    if (pairs) total = (uint64_t)p.B * p.H * (((p.Sq + 255) / 256 + 1) / 2);  // jobs = mirrored pairs of q-blocks
    const uint32_t cus = (uint32_t)w64_cu_count();
    if (const int gi = tuning().w64_grid.load(std::memory_order_relaxed)) {  // lab: force the number of workgroups
        const uint32_t g = (uint32_t)gi;
        if (gi > 0 && g <= cus && g <= total) return g;
    }
    if (!pairs) {
        // few items (the strong-scaling shards of a problem: B1 H3 S4096 = 48 items): a whole number of EQUAL parts per item
        // (grid = items x floor(CUs / items): the slices total * w / G then start and end on part boundaries, every
        // workgroup has one segment and one prologue) instead of CUs ragged slices.  Graph-replayed us, aligned / ragged:
        // H3 47.3 / 51.7, H6 71.5 / 72.7, H12 (one part: whole items) 109.1 / 110.0.
        const uint64_t items = (uint64_t)p.B * p.H * ((p.Sq + 255) / 256);
        if (items < cus && cus / items >= 2 && items * (cus / items) <= total) return (uint32_t)(items * (cus / items));
        // more than half an item per CU: whole items on `items` workgroups (some CUs idle) against cutting every item for all CUs and folding
        // it -- the cut costs ~28 us of prologues and folds, the idle CUs T (1 - items / CUs) tile steps of ~1.6 us.  Graph-replayed us, whole /
        // cut / 128-row kernel (profiles/r4/grid_probe.jsonl, grid_probe_more.jsonl): S 2048 (T = 32): 144 items 56 / 59 / 70, 176: 58 / 66 / 72, 216
        // (S 2304): 73 / 86 / 85; S 4096 (T = 64): 176 items 100 / 102 / 133, 192: 106 / 106 / 137, 240: 123 / 135 / 149; head_dim 64, 216 items 48 / 54 / 51
        if (items < cus && 2 * (uint64_t)w64_tiles_per_item(p) * (cus - items) < 35 * cus) return (uint32_t)items;
    }
    return total < cus ? (uint32_t)total : cus;  // never more workgroups than steps: every slice is non-empty
}

uint32_t fwd_w64_grid(const FwdParams& p) { return w64_grid(p); }

FwdW64Plan fwd_w64_plan(const FwdParams& p) {
    FwdW64Plan plan;
    const uint32_t items = p.B * p.H * ((p.Sq + 255) / 256);
    (void)items;
    plan.cnt_bytes = ((size_t)w64_grid(p) * sizeof(uint32_t) + 255) & ~(size_t)255;  // < grid shared items (causal: none)
    plan.buf_bytes = (size_t)2 * w64_grid(p) * (4 * 2 * 17 * 1024);
    return plan;
}

template <typename KFN>
static hipError_t launch_w64_kernel(KFN kfn, const FwdParams& p, const W64Params& wp, hipStream_t stream) {
    const uint32_t grid = w64_grid(p);
    // K/V rings + per-wave output staging + flag / ticket words + the mask kernels' bit-word ring, tile list and shared-block running sums
    const size_t lds = 65536 + 4 * 32 * (512 + 16) + 64 + 4096 + 8192 + 2048 + 64;
    if (hipError_t e = ensure_dynamic_lds((const void*)kfn, lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds, stream, wp);
    return hipGetLastError();
}

// part_cnt must be zero on entry (the runtime zeroes it when it allocates; the kernel leaves it zero).
hipError_t launch_fwd_w64(const FwdParams& p, float* part_buf, uint32_t* part_cnt, hipStream_t stream, const char** name) {
    if (!fwd_w64_supported(p) || !part_buf || !part_cnt) return hipErrorNotSupported;
    W64Params wp;
    wp.q = p.q; wp.k = p.k; wp.v = p.v; wp.o = p.o; wp.lse = p.lse;
    for (int i = 0; i < 3; ++i) { wp.qs[i] = p.qs[i]; wp.ks[i] = p.ks[i]; wp.vs[i] = p.vs[i]; }
    wp.B = p.B; wp.H = p.H; wp.Sq = p.Sq; wp.Skv = p.Skv;
    wp.scale = p.scale;
    wp.n_items = p.B * p.H * ((p.Sq + 255) / 256);
    wp.T = (p.Skv + 63) / 64;
    wp.part_buf = part_buf;
    wp.part_cnt = part_cnt;
    w64_softmax_policy(p.in_prec, &wp.tau, &wp.lazy);
    wp.skew = (uint32_t)tuning().w64_skew.load(std::memory_order_relaxed);
    // (skew = 0xffffffff would run the remainder of a multi-round launch as one more round of whole items (kernel: whole_rem).  Measured,
    // whole / cut: 464 items S 4096 242 / 235, 480 items S 2048 138 / 142, 448 items 134 / 130 -- a wash, so the launcher never asks for it;
    // lab option w64_skew = -1)
    if (tuning().w64_skew.load(std::memory_order_relaxed) == -1) wp.skew = 0xffffffffu;
    wp.rope_cos = p.rope_cos; wp.rope_sin = p.rope_sin; wp.rope_tb = p.rope_tb;
    wp.vsc = p.vsc; wp.vsc_bs = p.vsc_bs; wp.vsc_hs = p.vsc_hs;
    if (p.in_prec == P_BF16 && p.pv16 && !p.vsc) return hipErrorInvalidValue;  // the fp16 image of V comes with its slab exponents (runtime.hip)
    wp.Tw = wp.T; wp.win_left = wp.win_right = 0;
    const bool rope = p.rope_cos != nullptr, window = w64_is_window(p), fp32o = p.out_prec == P_FP32, maska = p.mask_kind == MK_F16, maskt = p.mask_kind == MK_BOOL || maska;
    wp.mask = nullptr; wp.mask_s[0] = wp.mask_s[1] = wp.mask_s[2] = 0;
    wp.guard = maska ? p.guard : nullptr;  // (fp32 masks: the launch runs iff the classification pass found the fp16 copy exact -- FwdParams::guard)
    if (p.guard && (!maska || p.guard_want != 0u)) return hipErrorInvalidValue;
    if (maskt) {
        if ((!maska && !p.mk_bits) || !p.mk_list || !p.mk_cnt) return hipErrorInvalidValue;  // runtime.hip packs / classifies the mask first (launch_mask_pack, launch_mask_classify)
        if (maska) { wp.mask = p.mask; wp.mask_s[0] = p.ms[0]; wp.mask_s[1] = p.ms[1]; wp.mask_s[2] = p.ms[2]; }
        wp.mk_bits = p.mk_bits; wp.mk_list = p.mk_list; wp.mk_cnt = p.mk_cnt;
        wp.mk_bs = p.mk_bs; wp.mk_hs = p.mk_hs; wp.mk_nrb64 = p.mk_nrb64;
        wp.mk_prefix = p.mk_prefix;
        // the max chain, unless the mask has no row dimension (key padding: a listed tile holds a key for every row, the lazy bodies are as safe as
        // without a mask); with one, which rows have keys in a segment is not arithmetic
#ifndef W64_LAB_MASK_FORCE_LAZY  // (lab, timing only -- NOT safe for masks with a row dimension in general: what would the lazy bodies buy a mask whose listed tiles are all open?)
        // (additive masks: the same rule -- a mask without a row dimension gives every row of a block the same keys in every listed tile, whatever the values;
        // with one, finite terms move a row's reference tile by tile and a steep bias would overflow the lazy bodies' stale reference segment after segment:
        // the max chain follows them)
        if (p.ms[2] != 0 || p.causal || tuning().no_w64_mask_lazy.load(std::memory_order_relaxed)) wp.lazy = 0;
#endif
    }
    if (rope && p.out_prec != p.in_prec) return hipErrorNotSupported;  // fused-RoPE instantiations: O in the operand type only (runtime.hip asks first)
    if (window) {
        wp.Tw = w64_tiles_per_item(p);
        wp.win_left = (int32_t)w64_win_left(p);
        wp.win_right = (int32_t)w64_win_right(p);
    }
    // one kernel family = one (operand type, P V type, head_dim); its instantiations: <OUT, causal>, <OUT, false, false, window>,
    // and at head_dim 128 <operand type, causal, rope>
#define W64_FAMILY(FAM, T16, HAS_ROPE)                                                                                         \
    do {                                                                                                                       \
        if constexpr (HAS_ROPE >= 2) {  /* families with mask-tensor instantiations (3: head_dim 64, no fused rotation) */      \
            if (maskt) return fp32o ? launch_w64_kernel(FAM<float, false, false, false, true>, p, wp, stream) : launch_w64_kernel(FAM<T16, false, false, false, true>, p, wp, stream); \
        }                                                                                                                      \
        if constexpr (HAS_ROPE == 1 || HAS_ROPE == 2) {                                                                        \
            if (rope) return p.causal ? launch_w64_kernel(FAM<T16, true, true>, p, wp, stream) : launch_w64_kernel(FAM<T16, false, true>, p, wp, stream); \
        }                                                                                                                      \
        if (window) return fp32o ? launch_w64_kernel(FAM<float, false, false, true>, p, wp, stream) : launch_w64_kernel(FAM<T16, false, false, true>, p, wp, stream); \
        if (p.causal) return fp32o ? launch_w64_kernel(FAM<float, true>, p, wp, stream) : launch_w64_kernel(FAM<T16, true>, p, wp, stream); \
        return fp32o ? launch_w64_kernel(FAM<float, false>, p, wp, stream) : launch_w64_kernel(FAM<T16, false>, p, wp, stream); \
    } while (0)
    static const char* const names[3][2][3] = {
        {{"fa_fwd16_w64<bf16,128>", "fa_fwd16_w64<bf16,128,rope>", "fa_fwd16_w64<bf16,128,window>"},
         {"fa_fwd16_w64<bf16,64>", "fa_fwd16_w64<bf16,64,rope>", "fa_fwd16_w64<bf16,64,window>"}},
        {{"fa_fwd16_w64<bf16,128,pv16>", "fa_fwd16_w64<bf16,128,pv16,rope>", "fa_fwd16_w64<bf16,128,pv16,window>"},
         {"fa_fwd16_w64<bf16,64,pv16>", "fa_fwd16_w64<bf16,64,pv16,rope>", "fa_fwd16_w64<bf16,64,pv16,window>"}},
        {{"fa_fwd16_w64<fp16,128>", "fa_fwd16_w64<fp16,128,rope>", "fa_fwd16_w64<fp16,128,window>"},
         {"fa_fwd16_w64<fp16,64>", "fa_fwd16_w64<fp16,64,rope>", "fa_fwd16_w64<fp16,64,window>"}}};
    if (p.D == 64 && rope) return hipErrorNotSupported;
    const int fam = p.in_prec == P_BF16 ? (p.pv16 ? 1 : 0) : 2;
    if (maskt && fam == 0) return hipErrorNotSupported;
    if (maska && fam == 0) return hipErrorNotSupported;
    if (maska) {  // the additive-mask families: a translation unit of their own (generated bodies of their own)
        *name = p.D == 64 ? (fam == 1 ? "fa_fwd16_w64<bf16,64,pv16,bias>" : "fa_fwd16_w64<fp16,64,bias>") : (fam == 1 ? "fa_fwd16_w64<bf16,128,pv16,bias>" : "fa_fwd16_w64<fp16,128,bias>");
        return launch_fwd_w64_bias(wp, fam, fp32o, w64_grid(p), 65536 + 4 * 32 * (512 + 16) + 64 + 4096 + 8192 + 2048 + 64, stream, (int)p.D);
    }
    *name = maska ? (fam == 1 ? "fa_fwd16_w64<bf16,128,pv16,bias>" : "fa_fwd16_w64<fp16,128,bias>") : maskt ? (p.D == 64 ? (fam == 1 ? "fa_fwd16_w64<bf16,64,pv16,mask>" : "fa_fwd16_w64<fp16,64,mask>")
                               : (fam == 1 ? "fa_fwd16_w64<bf16,128,pv16,mask>" : "fa_fwd16_w64<fp16,128,mask>"))
                  : names[fam][p.D == 64 ? 1 : 0][rope ? 1 : window ? 2 : 0];
    if (p.D == 64) {
        if (fam == 0) W64_FAMILY(fa_fwd16_w64d64_bf16, __bf16, 0);
        if (fam == 1) W64_FAMILY(fa_fwd16_w64d64_bf16pv16, __bf16, 3);
        W64_FAMILY(fa_fwd16_w64d64_f16, _Float16, 3);
    }
    if (fam == 0) W64_FAMILY(fa_fwd16_w64_bf16, __bf16, 1);
    if (fam == 1) W64_FAMILY(fa_fwd16_w64_bf16pv16, __bf16, 2);
    W64_FAMILY(fa_fwd16_w64_f16, _Float16, 2);
#undef W64_FAMILY
}

// ---- runtime-quantised variant ---------------------------------------------------------------------------------
bool fwd_w64_i8_supported(const FwdParams& p) {
    if (tuning().no_w64.load(std::memory_order_relaxed) || p.D != 128) return false;
    if (!(p.scale > 0.0f)) return false;
    if (p.Skv < 64 || p.Sq < 256 || (p.Sq % 256 != 0 && p.Sq < 1024)) return false;
    if (p.mask_kind == MK_BOOL && p.mask) {
        // (round 6) a bool mask tensor the caller handed over as it is (umfa_quantized_forward_masked_stream, MFABridge+Quantized.swift:227-358 with
        // mfa_prepare_mask's semantics): the MASKT instantiation of the int8 kernel -- packed bits, tile classes, block lists as on the 16-bit kernels.
        // The 16-bit route's shape rules (fwd_w64_supported): one 256-row block per CU, or a mask with a batch / head dimension of its own from 3/8
        if (tuning().no_w64_mask.load(std::memory_order_relaxed) || p.causal) return false;
        if (((p.Skv + 63) / 64) > 1024u || w64_grid(p) > 512u) return false;
        if (tuning().force_w64.load(std::memory_order_relaxed)) return true;
        const uint64_t blocks = (uint64_t)p.B * p.H * ((p.Sq + 255) / 256), cus = (uint64_t)w64_cu_count();
        return blocks >= cus || (p.ms[2] == 0 && blocks * ((p.Skv + 63) / 64) >= cus * 10) || ((p.ms[0] != 0 || p.ms[1] != 0) && blocks * 8 >= cus * 3);
    }
    return p.mask_kind == MK_NONE && !p.mask;
}

hipError_t launch_fwd_w64_i8(const FwdParams& p, const QuantViews& v, float* part_buf, uint32_t* part_cnt, hipStream_t stream) {
    if (!fwd_w64_i8_supported(p) || !part_buf || !part_cnt || v.dpq != 128) return hipErrorNotSupported;
    W64I8Params wp;
    wp.q8 = v.q8; wp.k8 = v.k8; wp.v16 = (const _Float16*)v.v16;
    wp.v8 = v.v8; wp.v_e8 = v.v_e8;
    wp.q_scale = v.q_scale; wp.k_scale = v.k_scale;
    wp.o = p.o; wp.lse = p.lse;
    wp.B = p.B; wp.H = p.H; wp.Sq = p.Sq; wp.Skv = p.Skv; wp.nqblk = v.nqblk; wp.nkblk = v.nkblk;
    wp.scale = p.scale;
    wp.n_items = p.B * p.H * ((p.Sq + 255) / 256);
    wp.T = (p.Skv + 63) / 64;
    wp.part_buf = part_buf;
    wp.part_cnt = part_cnt;
    w64_softmax_policy(P_FP16, &wp.tau, &wp.lazy);  // fp16 P
    wp.skew = (uint32_t)tuning().w64_skew.load(std::memory_order_relaxed);
    wp.vsc = v.v8 ? nullptr : p.vsc;
    const uint32_t grid = w64_grid(p);
    const bool f8 = v.v8 != nullptr, maskt = p.mask_kind == MK_BOOL;
    wp.mk_bits = wp.mk_list = wp.mk_cnt = nullptr; wp.mk_bs = wp.mk_hs = wp.mk_nrb64 = 0;
    if (maskt) {
        if (f8 || p.causal || !p.mk_bits || !p.mk_list || !p.mk_cnt) return hipErrorInvalidValue;  // the runtime packs the mask first (launch_mask_pack)
        wp.mk_bits = p.mk_bits; wp.mk_list = p.mk_list; wp.mk_cnt = p.mk_cnt;
        wp.mk_bs = p.mk_bs; wp.mk_hs = p.mk_hs; wp.mk_nrb64 = p.mk_nrb64;
        if (p.ms[2] != 0 || tuning().no_w64_mask_lazy.load(std::memory_order_relaxed)) wp.lazy = 0;  // (a mask with a row dimension: the max chain, as on the 16-bit kernels)
    }
    // K / V rings + output staging + flag words (+ the mask instantiation's bit-word ring, tile list and shared-block running sums: launch_w64_kernel's figure)
    const size_t lds = maskt ? 65536 + 4 * 32 * (512 + 16) + 64 + 4096 + 8192 + 2048 + 64 : 65536 + 4 * 32 * (512 + 16) + 16;
    auto kfn = f8 ? (p.causal ? fa_fwd_w64_i8f8<float, true> : fa_fwd_w64_i8f8<float, false>)
                  : maskt ? fa_fwd_w64_i8<float, false, true>
                          : (p.causal ? fa_fwd_w64_i8<float, true> : fa_fwd_w64_i8<float, false>);
    if (hipError_t e = ensure_dynamic_lds((const void*)kfn, lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds, stream, wp);
    return hipGetLastError();
}

}  // namespace umfa
