// fa_fwd_16_kernel.h -- the bf16/fp16 MFMA forward kernel template (see fa_fwd_16.hip for the design notes).
// Kept in a header so that tools/fwd_lab.hip can instantiate ONE variant for ablation/tuning builds.
#pragma once
#include <type_traits>

#include "fa_common.h"

namespace umfa {

typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
#define LDS_AS __attribute__((address_space(3)))

template <typename T> struct Mma16;
template <> struct Mma16<__bf16> {
    typedef bf16x8 V8;
    typedef bf16x4_t V4;
    static __device__ __forceinline__ f32x16 mma(V8 a, V8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ V4 tr_read(const char* lds) {
        return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((V4 LDS_AS*)(lds));
    }
    // The same MFMA with a VGPR destination and an AGPR-resident B operand, for accumulators the vector unit reads right
    // after (S, dP of the backward).  In a kernel that needs AGPRs at all hipcc selects the AGPR-destination form for EVERY
    // builtin MFMA, and each such value then costs a v_accvgpr_read per register before a VALU instruction can touch it.
    // Inline asm: hipcc pads nothing after it -- the CALLER keeps >= 2 MFMA issues between the last write of an
    // accumulator and its first vector read (XDL write -> VALU read: 18 wait states at 16 passes).
    static __device__ __forceinline__ void mma_v_first(f32x16& c, V8 a, V8 b) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "a"(b));
    }
    static __device__ __forceinline__ void mma_v(f32x16& c, V8 a, V8 b) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));
    }
};
template <> struct Mma16<_Float16> {
    typedef f16x8 V8;
    typedef f16x4_t V4;
    static __device__ __forceinline__ f32x16 mma(V8 a, V8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ V4 tr_read(const char* lds) {
        return __builtin_bit_cast(V4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 LDS_AS*)(lds)));
    }
    static __device__ __forceinline__ void mma_v_first(f32x16& c, V8 a, V8 b) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "a"(b));
    }
    static __device__ __forceinline__ void mma_v(f32x16& c, V8 a, V8 b) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));
    }
};

// LDS images.  A tile is [64 keys][DP] 16-bit elements; `ch` indexes 16-byte chunks of a row.
// K is read by rows (ds_read_b128, 16 lanes = 16 different keys at one chunk): spread the 16
// keys over the 16 chunk slots of a 256-byte bank row.
template <int DP> __device__ __forceinline__ constexpr int k_off(int row, int ch) {
    int sw = DP >= 128 ? (ch ^ (row & 15)) : DP == 64 ? (ch ^ ((row >> 1) & 7)) : (ch ^ ((row >> 2) & 3));
    return row * (2 * DP) + 16 * sw;
}
// V is read transposed (ds_read_b64_tr_b16: a 32-lane half reads 4 consecutive keys x 64 bytes):
// put the 4 keys in 4 different 64-byte bank segments.
template <int DP> __device__ __forceinline__ constexpr int v_off(int row, int ch) {
    int sw = DP >= 128 ? (ch ^ ((row & 3) << 2)) : DP == 64 ? (ch ^ (((row >> 1) & 1) << 2)) : ch;
    return row * (2 * DP) + 16 * sw;
}

__device__ __forceinline__ float max_xor32(float x) {
    // max over lane and lane^32 without LDS: v_permlane32_swap gives {x[l & 31], x[32 + (l & 31)]}
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// DMA = K/V tiles go global -> LDS directly (buffer_load ... lds, "LDS-DMA"): no staging VGPRs, no ds_write and no
// per-tile address VALU.  Requires head_dim == DP (a 16-byte chunk past D would belong to the next row).
// The loads are inline asm on purpose: for the builtin hipcc (ROCm 7.2) waits vmcnt(0) before every later LDS
// read (no alias information), which serialises the tile; here the only wait is ours, before the tile's barrier.
// BN = keys per tile (64, or 32: half the LDS and fewer live registers -> a third resident workgroup per CU).
// bf16 pair -> fp16 pair (round to nearest even; exact for 2^-17 <= |x| < 65536, +-inf beyond fp16's range -- the converting
// kernels notice that in their outputs and sweep again with the scaled form below)
__device__ __forceinline__ unsigned bf16x2_to_f16x2(unsigned x) {
    unsigned lo = x << 16, hi = x & 0xffff0000u, d;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(d) : "v"(lo), "v"(hi));
    return d;
}
// ... of x * mul, mul a power of two (exact unless the product leaves fp16's range at the small end)
__device__ __forceinline__ unsigned bf16x2_to_f16x2_scaled(unsigned x, float mul) {
    const float lo = __uint_as_float(x << 16) * mul, hi = __uint_as_float(x & 0xffff0000u) * mul;
    unsigned d;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(d) : "v"(lo), "v"(hi));
    return d;
}
// the power of two that brings a slab's largest |v| (bf16 bits; inf / NaN count as the largest finite exponent) into [2^15, 2^16):
// the rule of the cast pre-pass (fa_aux.hip vscale_exponent)
__device__ __forceinline__ int vscale_exponent_of(unsigned amax_bits) {
    if (amax_bits == 0) return 0;
    int E = (int)(amax_bits >> 7);
    E = E > 254 ? 254 : E;
    const int e = (E ? E - 127 : -126) - 15;  // (bf16's largest significand times 2^15 is 65280: inside fp16)
    return e < -100 ? -100 : e;
}

// PV16 (T = bf16 only; FwdParams::pv16, the default bf16 forward): the second product runs in fp16 -- P is rounded to fp16 (11
// bits instead of bf16's 8: the bf16-input forward inside 1e-3) and V is fp16:
//   PV16 = 1  V is converted bf16 -> fp16 inside the kernel: with LDS-DMA staging the tile lands as bf16 and every wave converts the
//             quarter its own DMA instructions filled, in place, before the tile's barrier (stage_write); with register staging on
//             the way into LDS.  Launches whose V tiles are not re-read often enough for a cast pre-pass to pay (runtime.hip).
//   PV16 = 2  p.v already points at an fp16 image of V (the runtime's cast pre-pass, as for fa_fwd16_w64): V staged like K.
//             Long launches: the conversion is 24 (head_dim 64) ... 48 (128) vector instructions per tile per wave in a kernel
//             that is vector-bound, and every workgroup repeats it.
//
// KS = 2 ("key-split", head_dim 64 only, round 4): the workgroup has EIGHT waves -- wave = 4 kh + rw; row-wave rw owns the
// same 32 query rows as before, key half kh the 32-key block kh of every 64-key tile, with a running max / sum / O^T of its
// own; the two halves of a row-wave meet once, in LDS, after the sweep (the split-KV fold's arithmetic without its fences).
// Why: at head_dim 64 a tile's softmax is 17 vector instructions per MFMA, and a launch like BASELINE config 2 (512 items
// = one round of two workgroups per CU) has two waves per SIMD that mostly take turns on it (vector unit 54 % busy, matrix
// pipe 23 %, profiles/r4/lab_notes.md section 2).  There are no more ROWS to make waves of; halving each wave's keys gives four
// waves per SIMD (<= 128 registers each at this head_dim), and a causal diagonal tile is skipped by the half it does not touch.
//
// PIPE = 1 ("software-pipelined", head_dim 64, round 4): S(t+1) = K(t+1) Q^T is issued INSIDE tile t's softmax -- the eight
// MFMAs of the next tile's scores between the exponentials of this tile's -- so a wave's matrix work runs under its own vector
// work instead of before it.  Measured why (in-kernel s_memtime buckets, profiles/r4/lab_notes.md section 2b): at config 2 the
// longest workgroup's wave spends 1613 cycles per tile in "compute" for ~860 cycles of issue; its Q K^T phase (LDS reads + a
// chain of MFMAs, 440 cycles) and its softmax (1007) follow each other, and the second wave of the SIMD is in the same phase.
// The K stream runs one tile ahead of the V stream (same two slots each: K(t)'s slot is free once S(t) exists).
//
// RESWEEP (PV16 = 1 only): the body once more, as the second sweep of a workgroup whose V does not fit fp16 as it is -- V is converted
// as v * 2^-vexp_in and the outputs are shifted back; see the check behind the tile loop.  The kernel runs the two copies one after
// the other (the first returns the exponent, 0 = done) with the second copy's inputs laundered through an empty asm, so that nothing
// of the first copy stays live for it.  (Tried first: a loop around the sweep -- as a back-edge it cost every converting instantiation
// 12-18 registers and the three-per-CU one 92 more spills; a noinline function -- the call ABI raised every kernel to 248 registers.)
template <typename T, int DP, bool CAUSAL, bool HAS_MASK, typename OUT, bool DMA, int BN, int PV16, int KS, int PIPE, bool CBAL, bool RESWEEP, typename PRM>
__device__ __forceinline__ int fa_fwd16_body(PRM& p, const int vexp_in, const int tid_in, const uint32_t bid_in);

// threads of a workgroup / resident workgroups per CU the register budget is set for (KS = 4, the decode form: four waves, 128-key tiles -- 128 KiB of LDS at
// head_dim 128: one workgroup per CU, two at head_dim 64)
constexpr int fwd16_threads(int KS) { return KS == 4 ? 256 : 256 * KS; }
constexpr int fwd16_resident(int DP, int BN, int KS) { return KS == 4 ? (DP > 64 ? 1 : 2) : KS * (DP > 128 ? 1 : (BN == 32 ? 3 : 2)); }

template <typename T, int DP, bool CAUSAL, bool HAS_MASK, typename OUT, bool DMA = false, int BN = 64, int PV16 = 0, int KS = 1, int PIPE = 0, bool CBAL = false>
__global__ __launch_bounds__(fwd16_threads(KS), fwd16_resident(DP, BN, KS)) void fa_fwd16_kernel(FwdParams p) {
    if constexpr (HAS_MASK) {
        // guarded launches (FwdParams::guard: an fp32 mask whose fp16 copy may or may not be exact -- the other route is enqueued too): uniform, before anything else
        if (p.guard != nullptr && *p.guard != p.guard_want) return;
    }
    int e2 = fa_fwd16_body<T, DP, CAUSAL, HAS_MASK, OUT, DMA, BN, PV16, KS, PIPE, CBAL, false, const FwdParams>(p, 0, (int)threadIdx.x, blockIdx.x);
    if constexpr (PV16 == 1) {
        if (__builtin_expect(e2 != 0, 0)) {
            // the second copy reads the parameters through the kernel-argument segment (FwdParams is the only argument: offset 0), so that
            // neither `p` has its address taken nor a value of the first copy is reused
            typedef const __attribute__((address_space(4))) FwdParams KFwdParams;
            KFwdParams* pp = (KFwdParams*)__builtin_amdgcn_kernarg_segment_ptr();
            int t = (int)threadIdx.x;
            uint32_t bx = blockIdx.x;
            asm volatile("" : "+s"(pp), "+v"(t), "+s"(bx), "+v"(e2));
            e2 = __builtin_amdgcn_readfirstlane(e2);  // (workgroup-uniform by construction)
            (void)fa_fwd16_body<T, DP, CAUSAL, HAS_MASK, OUT, DMA, BN, PV16, KS, PIPE, CBAL, true, KFwdParams>(*pp, e2, t, bx);
        }
    }
}

template <typename T, int DP, bool CAUSAL, bool HAS_MASK, typename OUT, bool DMA, int BN, int PV16, int KS, int PIPE, bool CBAL, bool RESWEEP, typename PRM>
__device__ __forceinline__ int fa_fwd16_body(PRM& p, const int vexp_in, const int tid_in, const uint32_t bid_in) {
    static_assert(!RESWEEP || PV16 == 1, "the second sweep exists for the converting kernels only");
    static_assert(!CBAL || (CAUSAL && !HAS_MASK && DMA && BN == 64 && KS == 1 && !PIPE && DP <= 128), "balanced causal pairs: LDS-DMA staging, no mask tensor");
    static_assert(!PIPE || (DMA && !HAS_MASK && KS == 1 && DP <= 64 && BN == 64), "pipelined loop: head_dim <= 64, LDS-DMA staging, no mask tensor");
    static_assert(!PV16 || __is_same(T, __bf16), "PV16: bf16 operands");
    // KS = 4 ("decode form", round 6): at most 32 query rows per item (decode-like calls, after umfa_torch packed a KV head's query heads into rows) -- the FOUR waves
    // of a workgroup all serve those rows, wave w owning key quarter w of every 128-key tile, and meet in LDS behind the sweep like KS = 2's halves.  In the plain form
    // three of the four waves compute rows that do not exist and the launch waits for wave 0's chain through a whole 64-key tile (1.2-1.5 us per tile of a lone
    // workgroup); here that chain covers 32 keys per wave and step.
    static_assert(KS == 1 || (KS == 2 && DMA && !HAS_MASK && DP == 64 && BN == 64) || (KS == 4 && DMA && !HAS_MASK && !CAUSAL && !PIPE && !CBAL && BN == 128 && (DP == 64 || DP == 128)),
                  "key-split: head_dim 64, LDS-DMA staging, no mask tensor; decode form: 128-key tiles, head_dim 64 / 128, no mask, not causal");
    constexpr int NT = fwd16_threads(KS);   // threads per workgroup
    constexpr int NW = NT / 64;             // waves
    constexpr bool VCONV = PV16 == 1;  // convert V in the kernel
    typedef Mma16<T> M;
    typedef typename M::V8 V8;
    typedef Mma16<typename std::conditional<PV16 != 0, _Float16, T>::type> MP;  // the P V product
    typedef typename MP::V8 PV8;
    typedef typename std::conditional<PV16 != 0, _Float16, T>::type PT;
    constexpr int BM = KS == 4 ? 32 : 128;  // query rows per item
    constexpr int NKB = BN / 32;            // 32-key blocks per tile
    constexpr int NST = BN / 16;            // 16-key MFMA k-steps of PV per tile
    constexpr int NCH = DP / 8;             // 16-byte chunks per row
    constexpr int NKS = DP / 16;            // k-steps of QK^T
    constexpr int NDB = DP / 32;            // 32-row blocks of O^T
    constexpr int NKBW = NKB / KS, NSTW = NST / KS;  // ... of them per wave (KS = 2: a wave sees half of every tile's keys)
    constexpr int TILE_BYTES = BN * DP * 2;
    constexpr int LPT = BN * NCH / 256;     // 16-byte loads per thread per tile
    // Ring depth of the LDS-DMA staging (the code below is written for any depth: tile t + NS - 1 is requested while tile t is
    // computed, waits leave the NS - 2 younger tiles in flight).  TWO: round 4 measured four slots at head_dim 64 (config 2 sits
    // in waits 44 % of its wave cycles) -- in-kernel stamps 21.2 vs 21.4 us per launch, kernel time unchanged: the waits are not
    // the successor tile's latency, the two waves of a SIMD queue for its vector unit (profiles/r4/lab_notes.md section 2).
    // KS = 2 (key-split): FOUR -- there a wave's share of a tile is ~500 issue cycles, less than an L2 -> LDS round trip, and with
    // two slots every iteration ended waiting for the tile it had requested at its start.
#ifdef UMFA_LAB_NS
    constexpr int NS = (CBAL || CAUSAL || HAS_MASK) ? 2 : UMFA_LAB_NS;  // (lab: the plain non-causal kernels only)
#else
    constexpr int NS = KS == 2 ? 4 : 2;
#endif
    constexpr bool SPLIT_DMA = DMA && NS == 2 && !PIPE && (DP == 256 || (DP == 64 && !CAUSAL));
    static_assert(LPT >= 1, "tile too small for 256 threads");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [K slot 0 .. NS-1][V slot 0 .. NS-1]
    char* const Kbuf = smem;
    char* const Vbuf = smem + NS * TILE_BYTES;

    const int tid = tid_in, wave = tid >> 6, lane = tid & 63, ql = lane & 31, hi = lane >> 5;
#ifdef UMFA_KS_ADJ  // lab (round 6): the two key halves of a row-wave on ADJACENT waves = different SIMDs (wave w runs on SIMD w % 4; as 4 kh + rw they share one)
    const int rw = KS == 1 ? wave : (wave >> 1), kh = KS == 1 ? 0 : (wave & 1);
#else
    const int rw = KS == 1 ? wave : KS == 4 ? 0 : (wave & 3), kh = KS == 1 ? 0 : KS == 4 ? wave : (wave >> 2);  // row-wave, key half (KS = 4: key quarter)
#endif
#ifdef UMFA_LAB_STAMPS
    unsigned long long stamp[6];
    unsigned long long cbs[4] = {0, 0, 0, 0};  // CBAL: switch begin / end, fold: flag seen / payload folded
    stamp[0] = __builtin_amdgcn_s_memrealtime();
    stamp[4] = __builtin_amdgcn_s_memtime();
#define UMFA_CB_STAMP(i) cbs[i] = __builtin_amdgcn_s_memrealtime()
#else
#define UMFA_CB_STAMP(i)
#endif
    const uint32_t nqb = (p.Sq + BM - 1) / BM;
    // work item = one 128-row query block of one (batch, head).  Items [0, n_full) are processed whole;
    // the remaining ones (a partial last round of workgroups) are split into nsplit key ranges.
    uint32_t item, part = 0, nparts = 1;
    // causal launches with a split plan (fwd_16_split_plan, short launches): the HEAVY half of a head's q-blocks
    // (qb >= nqb / 2) is cut into two key ranges, the light half stays whole; heavy parts are dispatched first
    const bool causal_split = CAUSAL && p.nsplit > 1;
    const uint32_t bx = causal_split ? gridDim.x - 1 - bid_in : bid_in;
    if (bx < p.n_full) {
        item = xcd_remap(bx, p.n_full);
    } else {
        const uint32_t n_tail = gridDim.x - p.n_full;
        const uint32_t j = xcd_remap(bx - p.n_full, n_tail);
        nparts = p.nsplit;
        item = p.n_full + j / nparts;
        part = j % nparts;
    }
    uint32_t bh = item / nqb;
    uint32_t qb = item % nqb;
    // CBAL state (all workgroup-uniform): cb_role 0 = one whole q-block, 1 = part A (tiles [0, cb_a) of q-block qb; folds part B in before it
    // stores), 2 = part B (tiles [cb_a, n) of q-block qb, published at step cb_sw; then the pair's short q-block cb_qb2 whole)
    uint32_t cb_role = 0, cb_a = 0, cb_sw = 0xffffffffu, cb_qb2 = 0, cb_n2 = 0, cb_pair = 0, cb_early = 0;
    if constexpr (CBAL) {
        const uint32_t h2 = nqb >> 1, npairs = p.B * p.H * h2;
        const bool is_b = bid_in < npairs;  // parts B first: a part A waits for its part B, never the other way round, and a part B waits for nothing
        const bool is_mid = bid_in >= 2 * npairs;  // an odd number of q-blocks: the middle one (as long as half a pair) runs whole, behind the pairs
        cb_pair = is_mid ? 0 : xcd_remap(is_b ? bid_in : bid_in - npairs, npairs);
        bh = is_mid ? xcd_remap(bid_in - 2 * npairs, p.B * p.H) : cb_pair / h2;
        const uint32_t qi = is_mid ? h2 : cb_pair % h2, qj = nqb - 1 - qi;
        const uint32_t ntl = (p.Skv + BN - 1) / BN;
        auto nt_of = [&](uint32_t q_) { const uint32_t lim = (q_ * BM + BM + BN - 1) / BN; return ntl < lim ? ntl : lim; };
        const uint32_t ni = nt_of(qi), nj = nt_of(qj);
        int a = (int)((ni + nj + 1) / 2) - (int)p.cbal_delta;
        a = a < 1 ? 1 : a;
        if (is_mid) {
            qb = qi;
        } else if ((uint32_t)a >= nj) {  // nothing to cut (Skv much shorter than Sq): both q-blocks whole
            qb = is_b ? qi : qj;
        } else {
            qb = qj;
            cb_a = (uint32_t)a;
            if (!is_b) cb_role = 1;
            else if (RESWEEP && (vexp_in & 0x10000)) qb = qi;  // second sweep of a part B whose tail was fine and IS published (the pair's flag is up or
                                                              // already taken): the short q-block whole, nothing published twice
            else { cb_role = 2; cb_sw = nj - cb_a; cb_qb2 = qi; cb_n2 = ni; }
        }
    } else
    if (causal_split) {
        const uint32_t h2 = nqb >> 1, idx = item < p.n_full ? item : item - p.n_full;
        bh = idx / h2;
        qb = (item < p.n_full ? 0 : h2) + idx % h2;
    } else
    if (CAUSAL) {
        qb = nqb - 1 - qb;  // longest items first
#ifndef UMFA_LAB_NO_CAUSAL_PAIRS
        // Causal items differ in length (q-block j sweeps j + 1 of nqb key ranges), and the workgroups that share a CU
        // finish together only if their lengths add up alike.  Measured on MI355X (tools/bench_graph.py, variants of the
        // swap bit): the dispatcher co-locates workgroups that are one CU-count of an XCD (32) apart in XCD-local order,
        // not consecutive ones.  So consecutive items form a mirrored pair (j, nqb - 1 - j) and the pair 32 items
        // further on has its long and short member swapped: every CU gets one long and one short item.
        // B4 H16 S1024 D64 causal (BASELINE config 2) 27.5 -> 22.5 us, D128 44.3 -> 39.8, B2 H16 S4096 D64 120 -> 107.
        // Only for short launches (the whole grid resident, or a few rounds of short items): beyond that the
        // dispatcher refills slots as they free up and plain longest-first is the better schedule -- a long item
        // dispatched late is the tail (B1 H32 S8192 D64: 342 us longest-first vs 369 paired; head_dim 256 runs one
        // workgroup per CU and has nothing to pair).
        if ((nqb & 1) == 0 && DP <= 128 && p.n_full == nqb * p.B * p.H  // every item whole (no split tail with its own numbering)
            && (uint64_t)p.n_full * nqb <= 32768) {
                        const uint32_t pi = item >> 1, second = item & 1, flip = (item >> 5) & 1, h2 = nqb >> 1;
            const uint32_t j = pi % h2;
            bh = pi / h2;
            qb = (second ^ flip) ? j : nqb - 1 - j;
        }
#endif
    }
    const uint32_t b = bh / p.H, h = bh % p.H;
    uint32_t q_row = qb * BM + rw * 32 + ql;    // (CBAL: a part B moves on to its second q-block)
    uint32_t wave_q0 = qb * BM + rw * 32;
    const int D = (int)p.D;

    // Hardware-bounds-checked buffer loads (T8): rows past the end of a (batch, head) slab and
    // head_dim columns >= D read as zero with no branch; no address arithmetic beyond one add per load.
    const T* qp = (const T*)p.q + ((int64_t)b * p.qs[0] + (int64_t)h * p.qs[1]);
    const T* kp = (const T*)p.k + ((int64_t)b * p.ks[0] + (int64_t)h * p.ks[1]);
    const T* vp = (const T*)p.v + ((int64_t)b * p.vs[0] + (int64_t)h * p.vs[1]);
    const int q_stride_b = (int)p.qs[2] * 2, k_stride_b = (int)p.ks[2] * 2, v_stride_b = (int)p.vs[2] * 2;
    const auto q_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)qp, 0, (int)((p.Sq - 1) * (uint32_t)q_stride_b + D * 2), 0x00020000);
    const auto k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)kp, 0, (int)((p.Skv - 1) * (uint32_t)k_stride_b + D * 2), 0x00020000);
    const auto v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)vp, 0, (int)((p.Skv - 1) * (uint32_t)v_stride_b + D * 2), 0x00020000);
    constexpr int OOB = 0x7fffff00;  // an offset no slab reaches (supported(): slab < 2 GiB)

    // ---- Q^T fragments (B operand of S^T = K Q^T): lane (q, hi) holds Q[q][16 ks + 8 hi .. +7]
    V8 qf[NKS];
    auto load_q = [&](const uint32_t row) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int d0 = 16 * ks + 8 * hi;
            const int off = (row < p.Sq && d0 < D) ? (int)row * q_stride_b + d0 * 2 : OOB;
            qf[ks] = __builtin_bit_cast(V8, __builtin_amdgcn_raw_buffer_load_b128(q_rsrc, off, 0, 0));
        }
    };
    load_q(q_row);

    // ---- tile staging (register path): thread owns chunks c = tid + 256 i  ->  (row, ch); offsets are tile-invariant
    constexpr int LPTR = DMA ? 1 : LPT;
    constexpr bool VREG = !DMA;               // V tiles through registers (DMA + VCONV: converted in place in LDS, below)
    constexpr int LPTV = VREG ? LPT : 1;
    int koff[LPTR], voff[LPTV], klds[LPTR], vlds[LPTV];
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 kreg[LPTR], vreg[LPTV];
    if constexpr (VREG) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int c = tid + 256 * i, row = c / NCH, ch = c % NCH;
            const bool colok = ch * 8 < D;
            if constexpr (!DMA) {
                koff[i] = colok ? row * k_stride_b + ch * 16 : OOB;
                klds[i] = k_off<DP>(row, ch);
            }
            voff[i] = colok ? row * v_stride_b + ch * 16 : OOB;
            vlds[i] = v_off<DP>(row, ch);
        }
    }
    // ---- tile staging (LDS-DMA path).  A wave-instruction writes 1 KiB of the tile image linearly: wave w,
    // instruction j covers image bytes [(w*IPW + j) KiB, +1 KiB) = rows (w*IPW + j)*RPI ...; lane l lands in row
    // r = l / NCH, chunk slot c = l % NCH and therefore FETCHES source chunk c ^ swz(row) (rule 21: the swizzle
    // goes on the source address; k_off / v_off are involutions in the chunk index).
    constexpr int RPI = 1024 / (2 * DP);          // rows per wave-instruction
    constexpr int IPW = TILE_BYTES / 1024 / NW;  // instructions per wave per tile (K and V each)
    static_assert(!DMA || IPW >= 1, "tile too small for the workgroup's waves");
    constexpr int IPWR = DMA ? IPW : 1;
    int kdma[IPWR], vdma[IPWR];
    i32x4 k_srd, v_srd;
    unsigned lds_wave = 0;
    if constexpr (DMA) {
        const int uw = __builtin_amdgcn_readfirstlane(wave);
        const int d_r = lane / NCH, d_c = lane % NCH;
#pragma unroll
        for (int j = 0; j < IPW; ++j) {
            const int row = (uw * IPW + j) * RPI + d_r;
            kdma[j] = row * k_stride_b + (k_off<DP>(row, d_c) - row * (2 * DP));
            vdma[j] = row * v_stride_b + (v_off<DP>(row, d_c) - row * (2 * DP));
        }
        auto make_srd = [&](const T* base, uint32_t bytes) {
            const unsigned long long a = (unsigned long long)base;
            i32x4 d;
            d[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
            d[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
            d[2] = __builtin_amdgcn_readfirstlane((int)bytes);
            d[3] = 0x00020000;
            return d;
        };
        k_srd = make_srd(kp, (p.Skv - 1) * (uint32_t)k_stride_b + D * 2);
        v_srd = make_srd(vp, (p.Skv - 1) * (uint32_t)v_stride_b + D * 2);
        lds_wave = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((LDS_AS char*)smem) + uw * IPW * 1024);
        // rows past Skv are range-checked away by the hardware; start from zeros so they can never hold NaNs
#pragma unroll
        for (int i = 0; i < 2 * NS * TILE_BYTES / (NT * 16); ++i) *(i32x4*)(smem + i * (NT * 16) + tid * 16) = i32x4{0, 0, 0, 0};
        __syncthreads();
    }
    // VCONV: V is converted as v * 2^-vexp: 0 in the kernel proper, the slab's power of two in the second sweep (RESWEEP)
    // (CBAL: bit 16 of the first sweep's return value says "part B, tail already published": see the work assignment)
    const int vexp = RESWEEP ? (int)(int16_t)(vexp_in & 0xffff) : 0;
    const float vmul = __uint_as_float((unsigned)(127 - vexp) << 23);
    (void)vmul;
    // which: 1 = K tile, 2 = V tile, 3 = both.  SPLIT_DMA: the V half is issued behind the QK^T MFMAs instead of back
    // to back with the K half (LDS-DMA instructions in a row stall the MFMA behind them, profiles/r1/lab_notes.md).
    // Same-box A/B: head_dim 256 (16 DMA instructions per wave per tile) 1132 -> 1001-1029 us at B2 H24 S4096,
    // head_dim 64 non-causal +2.5 %, head_dim 128 neutral, the tiny causal head_dim-64 case -3 % (left unsplit).
    // CBAL: the sweep's steps and the key tiles they visit differ for a part B (first the tail of the long q-block, then the short one from
    // tile 0); the ring's slots go by STEP
    auto tile_of = [&](uint32_t s_) -> uint32_t {
        if constexpr (CBAL) return cb_role == 2 ? (s_ < cb_sw ? cb_a + s_ : s_ - cb_sw) : s_;
        else return s_;
    };
    auto stage_load = [&](uint32_t t, int which = 3, uint32_t tk_ahead = 0) {  // tk_ahead: the K tile requested is t + tk_ahead (PIPE)
        if constexpr (DMA) {
            const uint32_t tk = t + tk_ahead;
            const int ktile = (int)(tile_of(tk) * BN) * k_stride_b, vtile = (int)(tile_of(t) * BN) * v_stride_b;
            const unsigned kdst = lds_wave + (tk % NS) * TILE_BYTES, vdst = lds_wave + (t % NS) * TILE_BYTES + NS * TILE_BYTES;
#pragma unroll
            for (int j = 0; j < IPW; ++j) {
                if (which & 1)
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                                 ::"s"(kdst + j * 1024), "v"(kdma[j] + ktile), "s"(k_srd) : "memory");
                if (which & 2)
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                                 ::"s"(vdst + j * 1024), "v"(vdma[j] + vtile), "s"(v_srd) : "memory");
            }
        } else {
            const int ksoff = (int)(t * BN) * k_stride_b, vsoff = (int)(t * BN) * v_stride_b;
#pragma unroll
            for (int i = 0; i < LPT; ++i) {
                // the tile advance goes into voffset: soffset is excluded from the hardware range check
                kreg[i] = __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, koff[i] + ksoff, 0, 0);
                vreg[i] = __builtin_amdgcn_raw_buffer_load_b128(v_rsrc, voff[i] + vsoff, 0, 0);
            }
        }
    };
    auto stage_write = [&](int buf, bool first = false) {
        if constexpr (DMA) {
            // this wave's LDS-DMA of the tile in slot `buf` has landed (then the barrier).  The counter runs in issue order: the
            // NS - 2 tiles requested after it (2 IPW instructions each) may stay in flight -- except in the prologue (`first`),
            // which drains everything once so that the compiler's own scoreboard (the Q fragment loads) is empty when the loop starts
            if (NS == 2 || first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * 2 * IPW) : "memory");
            // the same wait as a builtin keeps hipcc's scoreboard empty too: it cannot see the LDS-DMA loads, and with
            // the Q fragment loads (issued before the loop, first used inside it) still "pending" in its model it put
            // s_waitcnt vmcnt(3) ... vmcnt(0) in front of the first MFMAs of every tile -- right behind the issue of
            // the next tile's LDS-DMA, i.e. every tile waited for its successor's prefetch (tools/trace_waits.py)
            if (NS == 2 || first) __builtin_amdgcn_s_waitcnt(0x0F70);
            if constexpr (VCONV) {
                // bf16 -> fp16 IN PLACE (both are 16-bit): every wave converts the quarter of the V tile its own LDS-DMA
                // instructions filled (lane l of instruction j owns 16 bytes at (wave IPW + j) KiB + 16 l), so the vmcnt wait
                // above is all the ordering it needs; the barrier that follows publishes the converted tile
                char* const vq = Vbuf + buf * TILE_BYTES + (wave * IPW) * 1024 + lane * 16;
                if constexpr (!RESWEEP) {
#pragma unroll
                    for (int j = 0; j < IPW; ++j) {
                        const u32x4 r = *(const u32x4*)(vq + j * 1024);
                        *(u32x4*)(vq + j * 1024) = u32x4{bf16x2_to_f16x2(r[0]), bf16x2_to_f16x2(r[1]), bf16x2_to_f16x2(r[2]), bf16x2_to_f16x2(r[3])};
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < IPW; ++j) {
                        const u32x4 r = *(const u32x4*)(vq + j * 1024);
                        *(u32x4*)(vq + j * 1024) = u32x4{bf16x2_to_f16x2_scaled(r[0], vmul), bf16x2_to_f16x2_scaled(r[1], vmul),
                                                         bf16x2_to_f16x2_scaled(r[2], vmul), bf16x2_to_f16x2_scaled(r[3], vmul)};
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < LPT; ++i) {
                *(u32x4*)(Kbuf + buf * TILE_BYTES + klds[i]) = kreg[i];
                if constexpr (VCONV) {
                    if constexpr (!RESWEEP)
                        *(u32x4*)(Vbuf + buf * TILE_BYTES + vlds[i]) = u32x4{bf16x2_to_f16x2(vreg[i][0]), bf16x2_to_f16x2(vreg[i][1]),
                                                                             bf16x2_to_f16x2(vreg[i][2]), bf16x2_to_f16x2(vreg[i][3])};
                    else
                        *(u32x4*)(Vbuf + buf * TILE_BYTES + vlds[i]) = u32x4{bf16x2_to_f16x2_scaled(vreg[i][0], vmul), bf16x2_to_f16x2_scaled(vreg[i][1], vmul),
                                                                             bf16x2_to_f16x2_scaled(vreg[i][2], vmul), bf16x2_to_f16x2_scaled(vreg[i][3], vmul)};
                } else
                    *(u32x4*)(Vbuf + buf * TILE_BYTES + vlds[i]) = vreg[i];
            }
        }
    };

    uint32_t ntiles = (p.Skv + BN - 1) / BN;
    if (CAUSAL) {
        const uint32_t lim = (qb * BM + BM + BN - 1) / BN;
        ntiles = ntiles < lim ? ntiles : lim;
    }
    uint32_t t_begin = (uint32_t)(((uint64_t)ntiles * part) / nparts);
    uint32_t t_end = (uint32_t)(((uint64_t)ntiles * (part + 1)) / nparts);
    if constexpr (CBAL) {  // steps, not tiles (tile_of)
        if (cb_role == 1) t_end = cb_a;
        else if (cb_role == 2) t_end = cb_sw + cb_n2;
    }

    f32x16 acc[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    float m = -INFINITY;
    float l4[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // four partial row sums: short dependent chains
    const float c2 = p.scale * UMFA_LOG2E;

    // V transposed read: lane = 16 g + 4 qq + pp supplies row key0 + qq, columns 32 db + 16 (g&1) + 4 pp
    const int tr_qq = (lane >> 2) & 3, tr_pp = lane & 3, tr_g1 = (lane >> 4) & 1;
    int vtr[NDB];  // per-lane byte offset of the d-block's first transposed read (rows 4 hi + qq)
#pragma unroll
    for (int i = 0; i < NDB; ++i) vtr[i] = v_off<DP>(4 * hi + tr_qq, 4 * i + 2 * tr_g1 + (tr_pp >> 1)) + 8 * (tr_pp & 1);
    // the swizzles of v_off depend on row bits 0-1 only (DP >= 128) / bit 1 (DP == 64): adding a multiple
    // of 4 rows is a pure byte offset, so every other read of the tile is vtr[i] + const
    auto v_frag = [&](const char* Vt, int i, int st) -> PV8 {
        const typename MP::V4 lo = MP::tr_read(Vt + vtr[i] + (16 * st) * (2 * DP));
        const typename MP::V4 hi4 = MP::tr_read(Vt + vtr[i] + (16 * st + 8) * (2 * DP));
        return __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
    };

    const int64_t mrow = HAS_MASK ? ((int64_t)b * p.ms[0] + (int64_t)h * p.ms[1] + (int64_t)q_row * p.ms[2]) : 0;
    // masks with contiguous rows aligned to four elements: the four keys a lane owns per register group are ONE load
    // (dword for bytes, 8 bytes for fp16 / bf16, 16 bytes for fp32)
    const int mes = p.mask_kind == MK_BOOL ? 1 : (p.mask_kind == MK_F32 ? 4 : 2);
    // (mask_padded: the runtime's realigned copy of a mask whose rows were not -- rows padded to a multiple of four keys, so a group that starts below Skv may be read whole;
    // the edge tile's per-key compare masks what lies past Skv, as always)
    const bool mvec = HAS_MASK && p.mask_kind != MK_WINDOW && p.ms[3] == 1 && ((p.Skv & 3) == 0 || p.mask_padded) && ((p.ms[0] | p.ms[1] | p.ms[2]) & 3) == 0 &&
                      ((uintptr_t)p.mask & (uintptr_t)(4 * mes - 1)) == 0;
    // mask tile flags (FwdParams::mask_flags): this wave's 32 rows are one flag row; 64 tiles per register
    const uint8_t* mf_row = nullptr;
    int mf_reg = 0;
    if (HAS_MASK && p.mask_flags && wave_q0 / 32 < p.mf_nrb)
        mf_row = p.mask_flags + ((uint64_t)b * p.mf_bs + (uint64_t)h * p.mf_hs) * p.mf_nrb * p.mf_ntiles + (uint64_t)(wave_q0 / 32) * p.mf_ntiles;

    if (HAS_MASK && p.mask_flags) {
        // trim the sweep to [first, last] tile that any of the four waves has to visit: for banded masks most of the
        // key range is never staged at all (sliding window of +-512 at S = 4096: 18 of 64 tiles per workgroup)
        uint32_t lo = 0xffffffffu, hi1 = 0;  // first visited tile, one past the last
        if (mf_row) {
            for (uint32_t t0 = t_begin & ~63u; t0 < t_end; t0 += 64) {
                const uint32_t tt = t0 + lane;
                const bool visit = tt >= t_begin && tt < t_end && mf_row[tt] != 1;
                const unsigned long long bm = __builtin_amdgcn_ballot_w64(visit);
                if (bm) {
                    const uint32_t first = t0 + (uint32_t)__builtin_ctzll(bm), last = t0 + 63u - (uint32_t)__builtin_clzll(bm);
                    lo = lo < first ? lo : first;
                    hi1 = hi1 > last + 1 ? hi1 : last + 1;
                }
            }
        }
        volatile uint32_t* red = (volatile uint32_t*)smem;  // the tile area is not in use yet
        if (lane == 0) { red[2 * wave] = lo; red[2 * wave + 1] = hi1; }
        __syncthreads();
        lo = red[0]; hi1 = red[1];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const uint32_t a = red[2 * w], b2 = red[2 * w + 1];
            lo = lo < a ? lo : a;
            hi1 = hi1 > b2 ? hi1 : b2;
        }
        __syncthreads();
        lo = __builtin_amdgcn_readfirstlane(lo);  // workgroup-uniform by construction; tell the compiler
        hi1 = __builtin_amdgcn_readfirstlane(hi1);
        if (hi1 <= lo) { t_end = t_begin; }  // nothing to visit: O = 0, LSE = -inf like a fully masked row
        else { t_begin = lo; t_end = hi1; }
        if constexpr (DMA) {  // the scratch words sit in the zero-initialised tile area: restore
            if (tid < 8) ((volatile uint32_t*)smem)[tid] = 0;
            __syncthreads();
        }
    }
    if (HAS_MASK && p.mask_kind == MK_WINDOW) {  // the workgroup's 128 rows see keys [row0 - left, row0 + 127 + right] only
        const uint32_t row0 = qb * BM;
        const uint32_t lo = row0 > p.win_left ? (row0 - p.win_left) / BN : 0;
        const uint64_t hik = (uint64_t)row0 + BM - 1 + p.win_right;
        const uint32_t hi1 = hik / BN + 1 > t_end ? t_end : (uint32_t)(hik / BN + 1);
        t_begin = lo > t_begin ? lo : t_begin;
        t_end = hi1 < t_end ? hi1 : t_end;
        if (t_end < t_begin) t_end = t_begin;
    }
    // `red`: four free words of LDS (behind the loop: the tile area -- every wave is behind the loop's last barrier)
    auto v_range_check = [&](volatile uint32_t* const red) -> int {
        // fp16's range, checked where it is free -- on this workgroup's own outputs, once per item: a V value >= 65536 went into LDS as
        // +-inf and made every output it touches inf / NaN (P >= 0: 0 * inf = NaN, never a silent finite value); outputs that are ALL
        // below 2^-11 may have met values of V under 2^-17, which fp16 no longer holds exactly (absolute error <= 2^-25, i.e. <= 2^-14 of
        // an output of 2^-11).  Either way the workgroup takes the largest |v| of its slab and sweeps its keys again with V shifted by
        // that power of two (the cast pre-pass's rule; fa_fwd16_resweep: this body once more, out of line, from its first instruction).
        // Decided from the data alone: the same under graph replay, on any stream, and nothing for the host to read.
        const float lsum = (l4[0] + l4[1]) + (l4[2] + l4[3]);
        const float lrow = lsum + xor32(lsum);
        const float inv_c = lrow > 0.0f ? 1.0f / lrow : 0.0f;
        float chk_nan = 0.0f, chk_max = 0.0f;
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float val = acc[i][r] * inv_c;
                chk_nan = __builtin_fmaf(val, 0.0f, chk_nan);
                chk_max = fmaxf(chk_max, __builtin_fabsf(val));
            }
        const bool rowok = q_row < p.Sq;  // (rows past Sq were computed on zero Q rows: their "outputs" are means of V, and say nothing)
        const unsigned bits = (__builtin_amdgcn_ballot_w64(rowok && chk_nan != chk_nan) != 0 ? 1u : 0u) |
                              (__builtin_amdgcn_ballot_w64(rowok && chk_max >= 0x1p-11f) != 0 ? 2u : 0u) |
                              (__builtin_amdgcn_ballot_w64(rowok && lrow > 0.0f) != 0 ? 4u : 0u);
        if (lane == 0) red[wave] = bits;
        __syncthreads();
        unsigned all = 0;
#pragma unroll
        for (int w2 = 0; w2 < NT / 64; ++w2) all |= red[w2];
        all = __builtin_amdgcn_readfirstlane(all);
        // non-finite, or rows with keys and nothing above 2^-11  (otherwise on: the words stay -- the tile area is not read again, and the
        // folds below write before they read)
        if (__builtin_expect((all & 1u) || ((all & 4u) && !(all & 2u)), 0)) {
            unsigned amax = 0;
            const uint32_t d8 = (uint32_t)D / 8u, nch = p.Skv * d8;
            for (uint32_t c = (uint32_t)tid; c < nch; c += NT) {
                const uint32_t row = c / d8, ch = c - row * d8;
                const u32x4 r4 = __builtin_amdgcn_raw_buffer_load_b128(v_rsrc, (int)(row * (uint32_t)v_stride_b + ch * 16u), 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned a = r4[j] & 0x7fff7fffu, m2 = (a & 0xffffu) > (a >> 16) ? (a & 0xffffu) : (a >> 16);
                    amax = amax > m2 ? amax : m2;
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned o2 = (unsigned)__shfl_xor((int)amax, off, 64);
                amax = amax > o2 ? amax : o2;
            }
            __syncthreads();
            if (lane == 0) red[wave] = amax;
            __syncthreads();
            amax = 0;
#pragma unroll
            for (int w2 = 0; w2 < NT / 64; ++w2) amax = amax > red[w2] ? amax : red[w2];
            amax = __builtin_amdgcn_readfirstlane(amax);
            __syncthreads();
            // V all zero (the outputs were right), V with inf / NaN in it (they are what they should be: non-finite), V already where the
            // shift would put it: nothing a second sweep improves
            const int e2 = vscale_exponent_of(amax);
            if (amax != 0 && amax < 0x7f80u && e2 != 0) {
                return e2;  // -> the kernel runs the RESWEEP copy (the caller returns it)
            }
        }
        return 0;
    };
    auto v_shift_back = [&]() {
        if constexpr (RESWEEP) {  // the shift comes back (exact): everything below -- fold, output -- sees the values of the unshifted V
            const float back = __uint_as_float((unsigned)(127 + vexp) << 23);
    #pragma unroll
            for (int i = 0; i < NDB; ++i)
    #pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] *= back;
        } else if constexpr (PV16 == 2) {
            // the cast pre-pass shifted this slab's V by a power of two (FwdParams::vsc): back, before anything is published
            if (p.vsc) {
                const float back = p.vsc[128 * ((size_t)b * p.vsc_bs + (size_t)h * p.vsc_hs) + 65];
    #pragma unroll
                for (int i = 0; i < NDB; ++i)
    #pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] *= back;
            }
        }
    };
    // ---- CBAL ("balanced causal pairs", round 6; short causal launches, BASELINE config 2) ----
    // A causal launch whose workgroups are all resident at once ends when its longest item does: q-block nqb - 1 sweeps every key tile,
    // most of the time alone on its CU (config 2: 2.5 us + 16 tiles x 1.2 us + 1.3 us of a 24-us launch whose mean item is 9 tiles).
    // Here the items of a head are paired (qi, qj = nqb - 1 - qi) and every pair is dealt to TWO workgroups of equal length:
    //   part A  tiles [0, a) of the long q-block qj, a = ceil((n_i + n_j) / 2) - cbal_delta (the fold's price, in tiles);
    //   part B  tiles [a, n_j) of qj -- published to the pair's slot of part_buf the moment they are done -- then q-block qi whole,
    //           in ONE sweep: the LDS ring never drains, only Q and the accumulators change at the switch.
    // Part A folds B's (O^T, m, l) into its own registers and stores.  Parts B occupy the lower half of the grid, so they are dispatched
    // first and wait for nothing: a part A's wait always ends.  Fence-free like the split-KV fold below: write-through (sc1) stores,
    // vmcnt(0), barrier, a relaxed agent-scope flag.  Slot: [wave 4][chunk 4 NDB + 1][lane 64] x 16 bytes.
    constexpr int CB_CH = 4 * NDB + 1;
    auto cb_rsrc = [&]() {
        return __builtin_amdgcn_make_buffer_rsrc((void*)((char*)p.part_buf + (size_t)cb_pair * (4 * CB_CH * 1024)), 0, 4 * CB_CH * 1024, 0x00020000);
    };
    auto cb_switch = [&](uint32_t t) {
        // part B, top of step t = cb_sw: the step's tile is in slot t % NS (landed and published by the barrier that ended step t - 1),
        // the other slot is free (its request follows)
        volatile uint32_t* const red = (volatile uint32_t*)(Kbuf + ((t + 1) % NS) * TILE_BYTES);
        int e2 = 0;
        if constexpr (VCONV && !RESWEEP) {
            e2 = v_range_check(red);
            __syncthreads();
            if (tid < 4) red[tid] = 0;  // (the tile areas start from zeros: rows past Skv are never written)
        }
        if (e2) return e2 & 0xffff;  // (nothing published yet: the second sweep publishes)
        v_shift_back();
        const float lsum = (l4[0] + l4[1]) + (l4[2] + l4[3]);
        const float lrow = lsum + xor32(lsum);
        const auto prs = cb_rsrc();
        const int base = (wave * CB_CH) * 1024 + lane * 16;
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4{acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]}),
                                                       prs, base + (4 * i + g) * 1024, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4{lrow > 0.0f ? m : -INFINITY, lrow, 0.0f, 0.0f}), prs, base + (4 * NDB) * 1024, 0, 16);
#ifdef UMFA_CB_LAB_IMMEDIATE_FLAG
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(p.part_cnt + cb_pair, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        // (the flag follows at the END of this step, behind stage_write's vmcnt(0) and the step's barrier: nothing waits for the stores here)
        // the second q-block: its Q fragments were requested during the step before (behind that step's Q K^T)
        qb = cb_qb2;
        q_row = qb * BM + rw * 32 + ql;
        wave_q0 = qb * BM + rw * 32;
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
        m = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j) l4[j] = 0.0f;
        return 0;
    };
    if constexpr (!PIPE) {
#pragma unroll
        for (int i = 0; i < NS - 1; ++i) stage_load(t_begin + i);  // (tiles past the end: all zeros, same instruction count)
        stage_write(t_begin % NS, true);
        __syncthreads();
    }
#ifdef UMFA_LAB_STAMPS
    stamp[1] = __builtin_amdgcn_s_memrealtime();
#endif

#ifdef UMFA_LAB_LOOP_STAMPS
    // cycles of this wave: tile request issue | compute issue | wait for the next tile (+ V conversion) | barrier | of compute: until the
    // last Q K^T MFMA is issued | from there until the first P V MFMA
    uint32_t lp[6] = {0, 0, 0, 0, 0, 0};
#ifdef UMFA_LAB_LOOP_STAMPS_FINE
#define UMFA_LP_STAMP(x) __builtin_amdgcn_sched_barrier(0); const uint32_t x = (uint32_t)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0)
#else
#define UMFA_LP_STAMP(x) const uint32_t x = (uint32_t)__builtin_amdgcn_s_memtime()
#endif
#else
#define UMFA_LP_STAMP(x)
#endif
    if constexpr (PIPE) {
        // ---------------- software-pipelined sweep (see the template's comment) ----------------
        const uint32_t ntiles_all = (p.Skv + BN - 1) / BN;
        auto act = [&](uint32_t t) { return !CAUSAL || t * BN <= wave_q0 + 31; };
        auto clampt = [&](uint32_t t) { return t < ntiles_all ? t : ntiles_all; };  // (a tile past the end reads as zeros whatever its index)
        auto k_frags = [&](const char* Kt, V8 (&kf)[NKB][NKS]) {
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) kf[kb][ks] = *(const V8*)(Kt + k_off<DP>(32 * kb + ql, 2 * ks + hi));
        };
        auto qk = [&](const V8 (&kf)[NKB][NKS], f32x16 (&s)[NKB]) {
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] = 0.0f;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) s[kb] = M::mma(kf[kb][ks], qf[ks], s[kb]);
            }
        };
        // softmax of one tile's scores + P V; NEXT: the next tile's Q K^T (its K fragments in kf) is issued between the exponentials
        auto tile = [&](f32x16 (&s)[NKB], const char* Vt, uint32_t key_base, bool edge, auto has_next, auto&& next) {
            constexpr bool NEXT = decltype(has_next)::value;
            PV8 va[NST];
#pragma unroll
            for (int st = 0; st < NST; ++st) va[st] = v_frag(Vt, 0, st);
            if (edge) {
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const uint32_t key = key_base + 32 * kb + acc_row(r, hi);
                        if (key >= p.Skv || (CAUSAL && key > q_row)) s[kb][r] = -INFINITY;
                    }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
            mx = max_xor32(mx * c2);
            constexpr float TAU16 = 6.0f;  // the deferred reference of the ordinary loop below
            const bool move = mx > m + TAU16;
            float m_use = m;
            if (__any(move)) {
                m_use = move ? mx : m;
                const float alpha = __builtin_amdgcn_exp2f(m - m_use);
#pragma unroll
                for (int j = 0; j < 4; ++j) l4[j] *= alpha;
#pragma unroll
                for (int i = 0; i < NDB; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] *= alpha;
                m = m_use;
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NEXT) next();
            PV8 pf[NST];
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][r], c2, -m_use));
                    l4[r & 3] += e;
                    pf[2 * kb + (r >> 3)][r & 7] = (PT)e;
                }
            if constexpr (NEXT) {
                // one MFMA, then its share of the 3 vector instructions per score hipcc emits here (fma, exp, half a cvt_pk, half a packed add)
                constexpr int NM = NKB * NKS, PER = (NKB * 16 * 3) / NM;
#pragma unroll
                for (int i = 0; i < NM; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, PER, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NDB; ++i) {
                PV8 vb[NST];
                if (i + 1 < NDB) {
#pragma unroll
                    for (int st = 0; st < NST; ++st) vb[st] = v_frag(Vt, i + 1, st);
                }
#pragma unroll
                for (int st = 0; st < NST; ++st) acc[i] = MP::mma(va[st], pf[st], acc[i]);
                if (i + 1 < NDB) {
#pragma unroll
                    for (int st = 0; st < NST; ++st) va[st] = vb[st];
                }
            }
        };
        // prologue: K(t_begin), V(t_begin) requested first (then the barrier), K(t_begin + 1) behind them
        stage_load(t_begin, 3);
        stage_write(t_begin % NS, true);
        __syncthreads();
        stage_load(t_begin, 1, 1);
        f32x16 sA[NKB], sB[NKB];
        if (act(t_begin)) {
            V8 kf[NKB][NKS];
            k_frags(Kbuf + (t_begin % NS) * TILE_BYTES, kf);
            qk(kf, sA);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        auto step = [&](uint32_t t, f32x16 (&s_in)[NKB], f32x16 (&s_out)[NKB]) {
            const char* Kn = Kbuf + ((t + 1) % NS) * TILE_BYTES;  // K(t + 1): landed before the barrier that ended step t - 1
            const char* Vt = Vbuf + (t % NS) * TILE_BYTES;
            const uint32_t key_base = t * BN;
            // K(t + 2) into K(t)'s slot (S(t) exists), V(t + 1) into V(t - 1)'s
            {
                const uint32_t tv = clampt(t + 1), tk = clampt(t + 2);
                const int ktile = (int)(tk * BN) * k_stride_b, vtile = (int)(tv * BN) * v_stride_b;
                const unsigned kdst = lds_wave + (t % NS) * TILE_BYTES, vdst = lds_wave + ((t + 1) % NS) * TILE_BYTES + NS * TILE_BYTES;
#pragma unroll
                for (int j = 0; j < IPW; ++j) {
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                                 ::"s"(kdst + j * 1024), "v"(kdma[j] + ktile), "s"(k_srd) : "memory");
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                                 ::"s"(vdst + j * 1024), "v"(vdma[j] + vtile), "s"(v_srd) : "memory");
                }
            }
            const bool a_cur = act(t), a_next = t + 1 < t_end && act(t + 1);
            const bool edge = (key_base + BN > p.Skv) || (CAUSAL && key_base + BN - 1 > wave_q0);
            if (a_cur && a_next && !edge) {
                V8 kf[NKB][NKS];
                k_frags(Kn, kf);
                tile(s_in, Vt, key_base, false, std::true_type{}, [&]() { qk(kf, s_out); });
            } else {
                if (a_next) {
                    V8 kf[NKB][NKS];
                    k_frags(Kn, kf);
                    qk(kf, s_out);
                }
                if (a_cur) tile(s_in, Vt, key_base, edge, std::false_type{}, []() {});
            }
            stage_write((t + 1) % NS);
            __syncthreads();
        };
        for (uint32_t t = t_begin; t < t_end; t += 2) {
            step(t, sA, sB);
            if (t + 1 < t_end) step(t + 1, sB, sA);
        }
    } else {
    for (uint32_t t = t_begin; t < t_end; ++t) {
        UMFA_LP_STAMP(lp_t0);
        if constexpr (CBAL) {
            // part A: a first look at the pair's flag, a step ahead of the fold (the load's round trip hides under the last tile)
#ifndef UMFA_CB_LAB_NO_EARLY
            if (cb_role == 1 && t + 1 == t_end && tid == 0) cb_early = __hip_atomic_load(p.part_cnt + cb_pair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
            if (t == cb_sw) {  // part B: the long q-block's tail is done -- publish it, go on with the short q-block
                UMFA_CB_STAMP(0);
                if (const int e2 = cb_switch(t)) return e2;
                UMFA_CB_STAMP(1);
            }
        }
        const int cur = t % NS;
        // (KS = 2: this wave's half of the tile -- both swizzles depend on row bits below 32 / 4 only, so a whole number of
        // 32-key blocks is a pure byte offset)
        const char* Kt = Kbuf + cur * TILE_BYTES + kh * (32 * NKBW) * (2 * DP);
        const char* Vt = Vbuf + cur * TILE_BYTES + kh * (32 * NKBW) * (2 * DP);
        const uint32_t key_base = tile_of(t) * BN + kh * (32 * NKBW);  // first key of this wave's part of the tile
        constexpr uint32_t BNW = BN / KS;                     // keys of it
        // wave-uniform: is any part of this tile visible to this wave's rows?
        bool active = !CAUSAL || key_base <= wave_q0 + 31;
        if (KS >= 2) active = active && key_base < p.Skv;  // (the second half of a ragged last tile may hold no key at all)
        int mflag = 0;  // 1: every element of this wave's tile is masked (skip), 2: none is (no mask reads)
        if (HAS_MASK && mf_row) {
            if (t == t_begin || (t & 63) == 0) {
                const uint32_t t64 = t & ~63u;
                mf_reg = t64 + lane < p.mf_ntiles ? (int)mf_row[t64 + lane] : 0;
            }
            mflag = __builtin_amdgcn_readlane(mf_reg, (int)(t & 63));
            active = active && mflag != 1;
        } else if (HAS_MASK && p.mask_kind == MK_WINDOW) {
            // sliding window: the flags of this wave's 32 rows x this tile's keys are arithmetic
            const uint32_t k1 = key_base + BNW - 1, r1 = wave_q0 + 31;
            if (k1 + p.win_left < wave_q0 || key_base > r1 + p.win_right) mflag = 1;
            else if (key_base + p.win_left >= r1 && k1 <= wave_q0 + p.win_right) mflag = 2;
            active = active && mflag != 1;
        }
#ifndef UMFA_ABL_NO_LOAD
        // next tile in flight under this tile's MFMAs (T14); past the end: all zeros.  SPLIT_DMA: only the K half
        // here, the V half behind the QK^T MFMAs of an active tile
        stage_load(t + NS - 1, (SPLIT_DMA && active) ? 1 : 3);
#endif
        UMFA_LP_STAMP(lp_t1);

        if (active) {
            // ---------------- S^T = K Q^T ----------------
            f32x16 s[NKBW];
#pragma unroll
            for (int kb = 0; kb < NKBW; ++kb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] = 0.0f;
#pragma unroll
#ifdef UMFA_ABL_NO_QK
                for (int r = 0; r < 16; ++r) s[kb][r] = __builtin_bit_cast(float, (int)(Kt[r * 4 + lane] & 1) + 0x3f800000);
#else
                for (int ks = 0; ks < NKS; ++ks) {
                    const V8 a = *(const V8*)(Kt + k_off<DP>(32 * kb + ql, 2 * ks + hi));
                    s[kb] = M::mma(a, qf[ks], s[kb]);
                }
#endif
            }
#ifndef UMFA_ABL_NO_LOAD
            if constexpr (SPLIT_DMA) {
                __builtin_amdgcn_sched_barrier(0);
                stage_load(t + NS - 1, 2);
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
#ifdef UMFA_LAB_LOOP_STAMPS_FINE
            UMFA_LP_STAMP(lp_tq);
            lp[4] += lp_tq - lp_t1;
#endif
            if constexpr (CBAL) {
                // part B, last step of the long q-block: its Q fragments are dead from here on -- request the short q-block's under this tile's softmax
                // and P V (stage_write's vmcnt(0) at the end of the step covers them)
                if (t + 1 == cb_sw) {
                    __builtin_amdgcn_sched_barrier(0);
                    load_q(cb_qb2 * BM + rw * 32 + ql);
                }
            }
            // first V^T fragments requested before the softmax so their LDS latency hides under it
            PV8 va[NSTW];
#pragma unroll
            for (int st = 0; st < NSTW; ++st) va[st] = v_frag(Vt, 0, st);

            // ---------------- online softmax (log2 domain) ----------------
            const bool edge = (key_base + BNW > p.Skv) || (CAUSAL && key_base + BNW - 1 > wave_q0);
            float mx = -INFINITY;
            if (HAS_MASK && mvec && mflag != 2) {
                // registers 4g .. 4g+3 of a 32-key block are keys 8g + 4hi + 0..3: one aligned dword of the mask row.
                // Two copies (round 5): interior tiles -- every key below Skv, nothing above a causal diagonal -- skip the two compares and the select per
                // SCORE that only edge tiles need (an additive mask makes this body vector-bound: FLUX with an fp16 bias 0.56 ms against 0.22 unmasked)
                auto mixed_tile = [&](auto EDGE_T) {
                    constexpr bool EDGE = decltype(EDGE_T)::value;
#pragma unroll
                    for (int kb = 0; kb < NKBW; ++kb)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const uint32_t key0 = key_base + 32 * kb + 8 * g + 4 * hi;
                            float term[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // keys past Skv / rows past Sq are masked below anyway
                            if ((!EDGE || key0 < p.Skv) && q_row < p.Sq) {
                                const int64_t at = mrow + key0;
                                if (p.mask_kind == MK_BOOL) {
                                    const uint32_t w = *(const uint32_t*)((const uint8_t*)p.mask + at);
#pragma unroll
                                    for (int e = 0; e < 4; ++e) term[e] = ((w >> (8 * e)) & 0xffu) ? 0.0f : -INFINITY;
                                } else if (p.mask_kind == MK_F32) {
                                    const f32x4 w = *(const f32x4*)((const float*)p.mask + at);
#pragma unroll
                                    for (int e = 0; e < 4; ++e) term[e] = w[e] * UMFA_LOG2E;
                                } else {
                                    typedef uint16_t u16x4_t __attribute__((ext_vector_type(4)));
                                    const u16x4_t w = *(const u16x4_t*)((const uint16_t*)p.mask + at);
#pragma unroll
                                    for (int e = 0; e < 4; ++e)
                                        term[e] = (p.mask_kind == MK_F16 ? (float)__builtin_bit_cast(_Float16, (uint16_t)w[e])
                                                                         : bf16_bits_to_float(w[e])) * UMFA_LOG2E;
                                }
                            }
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int r = 4 * g + e;
                                float tv = s[kb][r] * c2 + term[e];
                                if (EDGE && (key0 + e >= p.Skv || (CAUSAL && key0 + e > q_row))) tv = -INFINITY;
                                s[kb][r] = tv;
                                mx = fmaxf(mx, tv);
                            }
                        }
                };
                if (edge) mixed_tile(std::true_type{});
                else mixed_tile(std::false_type{});
            } else if (HAS_MASK && mflag == 2 && !edge) {
                // fully open interior tile: nothing to read, nothing to compare
#pragma unroll
                for (int kb = 0; kb < NKBW; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        s[kb][r] *= c2;
                        mx = fmaxf(mx, s[kb][r]);
                    }
            } else if (HAS_MASK) {
#pragma unroll
                for (int kb = 0; kb < NKBW; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const uint32_t key = key_base + 32 * kb + acc_row(r, hi);
                        float tv = s[kb][r] * c2;
                        if (mflag != 2 && key < p.Skv && q_row < p.Sq)
                            tv += p.mask_kind == MK_WINDOW ? window_term(q_row, key, p.win_left, p.win_right)
                                                           : mask_term(p.mask, mrow + (int64_t)key * p.ms[3], p.mask_kind);
                        if (key >= p.Skv || (CAUSAL && key > q_row)) tv = -INFINITY;
                        s[kb][r] = tv;
                        mx = fmaxf(mx, tv);
                    }
            } else {
                if (edge) {
#pragma unroll
                    for (int kb = 0; kb < NKBW; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const uint32_t key = key_base + 32 * kb + acc_row(r, hi);
                            if (key >= p.Skv || (CAUSAL && key > q_row)) s[kb][r] = -INFINITY;
                        }
                }
#pragma unroll
                for (int kb = 0; kb < NKBW; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
                mx *= c2;  // scale > 0 on this path
            }
            mx = max_xor32(mx);
            // Deferred reference (round 4; the one-wave-per-SIMD kernels' "deferred" policy, tau = 6): a row's reference moves
            // only when this tile's max exceeds it by more than 2^tau -- P <= 64 fits fp16 and bf16 alike, the row sums are
            // fp32 -- so after a row's first tile the 36 rescale multiplies (O^T, l) run almost never instead of on nearly every
            // tile of a short causal sweep.  This kernel is vector-bound (config 2: 267 vector instructions per 16 MFMAs).
            // A row that has seen no key yet has m = -inf: its first finite max always moves it.
            constexpr float TAU16 = 6.0f;
            const bool move = mx > m + TAU16;
            // (rows that have seen no key: under a mask tensor, and -- KS = 2 -- the rows of a causal diagonal block's second key half)
            constexpr bool EMPTY_ROWS = HAS_MASK || KS >= 2 || CBAL;  // (CBAL: a part B starts above tile 0)
            float m_use = (EMPTY_ROWS && m == -INFINITY) ? 0.0f : m;
            if (__any(move)) {
                const float m_new = move ? mx : m;
                m_use = (EMPTY_ROWS && m_new == -INFINITY) ? 0.0f : m_new;
                const float alpha = __builtin_amdgcn_exp2f(m - m_use);  // exactly 1 for the rows that stay
#pragma unroll
                for (int j = 0; j < 4; ++j) l4[j] *= alpha;
#pragma unroll
                for (int i = 0; i < NDB; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] *= alpha;
                m = m_new;
            }
            PV8 pf[NSTW];
#pragma unroll
            for (int kb = 0; kb < NKBW; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
#ifdef UMFA_ABL_NO_EXP
                    const float e = __builtin_fmaf(s[kb][r], c2, -m_use);
#else
                    const float e = HAS_MASK ? __builtin_amdgcn_exp2f(s[kb][r] - m_use)
                                             : __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][r], c2, -m_use));
#endif
                    l4[r & 3] += e;
                    pf[2 * kb + (r >> 3)][r & 7] = (PT)e;
                }

#ifdef UMFA_LAB_LOOP_STAMPS_FINE
            UMFA_LP_STAMP(lp_ts);
            lp[5] += lp_ts - lp_tq;
#endif
            // ---------------- O^T += V^T P^T (fragments of block i+1 requested before block i's MFMAs) ----
#pragma unroll
            for (int i = 0; i < NDB; ++i) {
                PV8 vb[NSTW];
                if (i + 1 < NDB) {
#pragma unroll
                    for (int st = 0; st < NSTW; ++st) vb[st] = v_frag(Vt, i + 1, st);
                }
#ifdef UMFA_ABL_NO_PV
#pragma unroll
                for (int st = 0; st < NSTW; ++st) acc[i][st] += (float)va[st][0] * (float)pf[st][0];
#else
#pragma unroll
                for (int st = 0; st < NSTW; ++st) acc[i] = MP::mma(va[st], pf[st], acc[i]);
#endif
                if (i + 1 < NDB) {
#pragma unroll
                    for (int st = 0; st < NSTW; ++st) va[st] = vb[st];
                }
            }
        }

        if constexpr (CBAL) {
            if (!active && t + 1 == cb_sw) load_q(cb_qb2 * BM + rw * 32 + ql);  // (a wave that sat this tile out)
        }
        UMFA_LP_STAMP(lp_t2);
#ifndef UMFA_ABL_NO_LOAD
        stage_write((t + 1) % NS);
#endif
        UMFA_LP_STAMP(lp_t3);
#ifndef UMFA_ABL_NO_BARRIER
        __syncthreads();
#endif
        if constexpr (CBAL) {
            // part B, the step that began with the switch: every wave's write-through stores of the pair's slot are complete (vmcnt(0) in
            // stage_write, then the barrier) -- raise the pair's flag
#ifndef UMFA_CB_LAB_IMMEDIATE_FLAG
            if (t == cb_sw && tid == 0) __hip_atomic_store(p.part_cnt + cb_pair, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        }
#ifdef UMFA_LAB_LOOP_STAMPS
        {
            UMFA_LP_STAMP(lp_t4);
            lp[0] += lp_t1 - lp_t0; lp[1] += lp_t2 - lp_t1; lp[2] += lp_t3 - lp_t2; lp[3] += lp_t4 - lp_t3;
        }
#endif
    }
    }  // !PIPE

    if constexpr (VCONV && !RESWEEP) {
        if constexpr (NS > 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the ring's youngest requests still land in the tile area)
        if (const int e2 = v_range_check((volatile uint32_t*)smem)) return (e2 & 0xffff) | (CBAL && cb_role == 2 ? 0x10000 : 0);
    }
    v_shift_back();

#ifdef UMFA_LAB_STAMPS
    stamp[2] = __builtin_amdgcn_s_memrealtime();
    stamp[5] = __builtin_amdgcn_s_memtime();
#endif
    // ---------------- epilogue ----------------
    float l = (l4[0] + l4[1]) + (l4[2] + l4[3]);
    float lt = l + xor32(l);
    bool lead = true;  // this wave holds rows to store (KS = 4: key quarter 0, once the others are folded in)
    if constexpr (KS >= 2) {
        // the key halves (KS = 4: quarters) of a row-wave meet in LDS: the others publish their un-normalised (O^T, m, l), part 0 folds them in
        // (the split-KV fold's arithmetic) and stores.  The tile area is free: every LDS-DMA write has landed (the loop's last
        // stage_write waited vmcnt(0)) and the loop's last barrier is behind every wave's last tile read.
        constexpr int NREG = 16 * NDB + 2;
        constexpr int NOTH = KS == 4 ? 3 : 1;  // publishing parts per row-wave
        float* const ex0 = (float*)smem + (KS == 4 ? 0 : rw * (NREG * 64)) + lane;  // (KS = 4: areas 0 .. 2 for quarters 1 .. 3)
        if constexpr (NS > 2) {  // the ring's youngest requests (tiles past the end) are still on their way into the tile area
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (kh != 0) {
            float* const ex = ex0 + (KS == 4 ? (kh - 1) * (NREG * 64) : 0);
#pragma unroll
            for (int i = 0; i < NDB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) ex[(16 * i + r) * 64] = acc[i][r];
            ex[(16 * NDB) * 64] = m;
            ex[(16 * NDB + 1) * 64] = lt;
        }
        __syncthreads();
        if (kh != 0) {
            if constexpr (KS == 2) return 0;
            lead = false;  // (KS = 4: stays for the barriers of a split-KV fold, stores nothing)
        } else {
#pragma unroll
            for (int o = 0; o < NOTH; ++o) {
                const float* const ex = ex0 + o * (NREG * 64);
                const float mo = ex[(16 * NDB) * 64], lo = ex[(16 * NDB + 1) * 64];
                const float mn = fmaxf(m, mo);
                const float mu = mn == -INFINITY ? 0.0f : mn;
                const float a0 = __builtin_amdgcn_exp2f(m - mu), a1 = __builtin_amdgcn_exp2f(mo - mu);
#pragma unroll
                for (int i = 0; i < NDB; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] = acc[i][r] * a0 + ex[(16 * i + r) * 64] * a1;
                lt = lt * a0 + lo * a1;
                m = mn;
            }
        }
    }
    if constexpr (CBAL) {
        if (cb_role == 1) {
            // part A: fold the pair's part B in (it was published tiles ago: B's share of the long q-block is the shorter one).  The wait is
            // bounded all the same -- two seconds, then the rows come out NaN rather than the queue hanging
            // (the tile area is free: every LDS-DMA write has landed, every wave is behind the loop's last barrier; word 16: the range check's four words may still be read)
            volatile uint32_t& flag_s = *((volatile uint32_t*)smem + 16);
            if (tid == 0) {
                const uint64_t t_in = __builtin_amdgcn_s_memrealtime();
                uint32_t f = cb_early;
                while (f == 0 && __builtin_amdgcn_s_memrealtime() - t_in < 200000000ull) {
                    __builtin_amdgcn_s_sleep(2);
                    f = __hip_atomic_load(p.part_cnt + cb_pair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (f != 0) __hip_atomic_store(p.part_cnt + cb_pair, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // zero between launches
                flag_s = f;
            }
            __syncthreads();
            UMFA_CB_STAMP(2);
            const bool got = flag_s != 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // compiler-only: the payload loads stay below the flag
            const auto prs = cb_rsrc();
            const int base = (wave * CB_CH) * 1024 + lane * 16;
            typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
            f32x4 x[CB_CH];
#pragma unroll
            for (int c = 0; c < CB_CH; ++c) x[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, base + c * 1024, 0, 16));
            const float mo = x[4 * NDB][0], lo = x[4 * NDB][1];
            const float mn = fmaxf(m, mo);
            const float mu = mn == -INFINITY ? 0.0f : mn;
            const float a0 = got ? __builtin_amdgcn_exp2f(m - mu) : __builtin_nanf(""), a1 = __builtin_amdgcn_exp2f(mo - mu);
#pragma unroll
            for (int i = 0; i < NDB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = acc[i][r] * a0 + x[4 * i + (r >> 2)][r & 3] * a1;
            lt = lt * a0 + lo * a1;
            m = mn;
            UMFA_CB_STAMP(3);
        }
    }
    if ((KS == 1 || KS == 4) && nparts > 1) {
        // Split-KV combine (cdna_hip_programming.md Guideline 16, "every load sc1" form -- the fold protocol of fa_fwd16_w64): every
        // part publishes its un-normalised (O^T, m, l) with WRITE-THROUGH (sc1) stores -> vmcnt(0) -> barrier -> relaxed ticket; the part
        // that draws the last ticket folds the others with sc1 loads.  No release / acquire fence anywhere: an agent-scope release is
        // a write-back of the XCD's whole L2, and a launch of a few hundred parts paid for a few hundred of them (round 3: the causal
        // half-split 2 x SLOWER than no split, profiles/r3/cfg2_causal_split_ab.json; round 4 with this protocol: lab notes section 6).
        // Slot layout (round 6): [part][wave 4][chunk 4 NDB + 1][lane 64] x 16 bytes -- chunk 4 i + g = O^T registers 16 i + 4 g .. + 3 of block i,
        // the last chunk {m, l, 0, 0}: 16-byte stores and loads (a part is 4 NDB + 1 memory instructions per wave instead of 16 NDB + 2).  A wave whose 32
        // rows lie past Sq (decode-like calls: one query row, one wave) publishes and folds nothing.  The fold reads part o + 1 while it folds part o
        // (two register buffers): a launch of 8 heads x 32 parts spent 64 of its 202 us reading parts one after the other, a dependent round trip of
        // write-through memory each (profiles/r6/decode_fold.txt).  Same values in the same order as before: bitwise the same results.
        constexpr int NCHK = 4 * NDB + 1;
        // (fwd_16_split_plan sizes the buffer: 16 NDB + 4 words per lane, wave and part)
        // every static __shared__ object would shift the dynamic LDS base (Guideline 17): reuse the tile area
        volatile uint32_t& ticket_s = *((volatile uint32_t*)smem + 16);
        const uint32_t sidx = item - p.n_full;
        const size_t item_bytes = (size_t)nparts * 4 * (size_t)(NCHK * 1024);
        const auto prs = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)p.part_buf + (size_t)sidx * item_bytes), 0, (int)item_bytes, 0x00020000);
        auto slot = [&](uint32_t part_, int chunk) -> int { return (int)(((part_ * 4 + (uint32_t)wave) * NCHK + (uint32_t)chunk) * 1024u) + lane * 16; };
        constexpr int SC1 = 16;  // cache policy bit of the buffer builtins: system-coherent level 1 = write-through / read-around the XCD's L2
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        if constexpr (KS == 4) __syncthreads();  // (the exchange areas above sit in the tile area the ticket word below is in)
        const bool wave_rows = wave_q0 < p.Sq && lead;  // (wave-uniform)
        if (wave_rows) {
#pragma unroll
            for (int i = 0; i < NDB; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4{acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]}),
                                                           prs, slot(part, 4 * i + g), 0, SC1);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4{m, lt, 0.0f, 0.0f}), prs, slot(part, 4 * NDB), 0, SC1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // also: every wave is done with the K/V tiles, so the tile area is free
        if (tid == 0) ticket_s = __hip_atomic_fetch_add(p.part_cnt + sidx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const uint32_t ticket = ticket_s;
        if (ticket != nparts - 1) return 0;  // not the last part of this item
        // every part has drawn: the word is free again -- leave it zero for the next launch (no memset per launch, and no
        // memset node in a captured graph)
        if (tid == 0) __hip_atomic_store(p.part_cnt + sidx, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // compiler-only: the payload loads stay below the ticket
        if (!wave_rows) return 0;  // (behind the workgroup's last barrier)
        // fold the parts in index order (this workgroup's own part from its slot like the others: the same values): the result does not
        // depend on which part arrived last, so two launches are bitwise identical
        auto ldp = [&](uint32_t o, f32x4 (&x)[NCHK]) {
#pragma unroll
            for (int c = 0; c < NCHK; ++c) x[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, slot(o, c), 0, SC1));
        };
        auto foldp = [&](const f32x4 (&x)[NCHK], const bool first) {
            const float mo = x[4 * NDB][0], lo = x[4 * NDB][1];
            if (first) {
#pragma unroll
                for (int i = 0; i < NDB; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] = x[4 * i + (r >> 2)][r & 3];
                m = mo;
                lt = lo;
                return;
            }
            const float mn = fmaxf(m, mo);
            const float mu = mn == -INFINITY ? 0.0f : mn;
            const float a0 = __builtin_amdgcn_exp2f(m - mu), a1 = __builtin_amdgcn_exp2f(mo - mu);
#pragma unroll
            for (int i = 0; i < NDB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = acc[i][r] * a0 + x[4 * i + (r >> 2)][r & 3] * a1;
            lt = lt * a0 + lo * a1;
            m = mn;
        };
        f32x4 xa[NCHK], xb[NCHK];
        ldp(0, xa);
        for (uint32_t o = 0; o < nparts; o += 2) {
            if (o + 1 < nparts) ldp(o + 1, xb);
            foldp(xa, o == 0);
            if (o + 1 < nparts) {
                if (o + 2 < nparts) ldp(o + 2, xa);
                foldp(xb, false);
            }
        }
    }
    const float inv = lt > 0.0f ? 1.0f / lt : 0.0f;
    if (q_row < p.Sq && lead) {
        OUT* __restrict__ op = (OUT*)p.o + ((int64_t)bh * p.Sq + q_row) * D;
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = 32 * i + 8 * g + 4 * hi;
                if (d0 < D) {  // D % 8 == 0 on this path, so a group of 4 is all-in or all-out
                    if constexpr (sizeof(OUT) == 4) {
                        f32x4 val = {acc[i][4 * g] * inv, acc[i][4 * g + 1] * inv, acc[i][4 * g + 2] * inv,
                                     acc[i][4 * g + 3] * inv};
                        *(f32x4*)(op + d0) = val;
                    } else {
                        typedef OUT O4 __attribute__((ext_vector_type(4)));
                        O4 val = {(OUT)(acc[i][4 * g] * inv), (OUT)(acc[i][4 * g + 1] * inv),
                                  (OUT)(acc[i][4 * g + 2] * inv), (OUT)(acc[i][4 * g + 3] * inv)};
                        *(O4*)(op + d0) = val;
                    }
                }
            }
        if (p.lse && hi == 0)
            p.lse[(int64_t)bh * p.Sq + q_row] = lt > 0.0f ? (m + log2f(lt)) * UMFA_LN2 : -INFINITY;
    }
#ifdef UMFA_LAB_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp[3] = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) {  // a debug buffer of its own (lab only): [block][6]
        // (CBAL: behind the pairs' slots)
        unsigned long long* const dbg0 = (unsigned long long*)p.part_buf + (CBAL ? (size_t)(gridDim.x / 2) * (4 * CB_CH * 1024 / 8) : 0);
        unsigned long long* dbg = dbg0 + (size_t)bid_in * 8;
        for (int i = 0; i < 6; ++i) dbg[i] = stamp[i];
        if (CBAL) for (int i = 0; i < 4; ++i) dbg0[(size_t)gridDim.x * 16 + (size_t)bid_in * 4 + i] = cbs[i];
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        dbg[6] = xcc;
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        dbg[7] = hwid;
#ifdef UMFA_LAB_LOOP_STAMPS
        dbg = dbg0 + (size_t)gridDim.x * 8 + (size_t)bid_in * 6;
        for (int i = 0; i < 6; ++i) dbg[i] = lp[i];
#endif
    }
#endif
    return 0;
}

}  // namespace umfa
