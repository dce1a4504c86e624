// fa_common.h -- shared device/host definitions for the gfx950 attention kernels.
//
// Thread <-> matrix mapping used by every forward kernel here ("swapped QK^T",
// cdna_hip_programming.md T12): a wave owns 32 query rows and computes
// S^T = K Q^T with the 32x32 MFMA, so lane l holds query q = l & 31 and, in
// accumulator register r of a 32-key block, key = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5).
// Row max / row sum are therefore lane-local plus one lane <-> lane^32 exchange,
// and O^T = V^T P^T takes the S^T accumulator as its B operand without any
// cross-lane movement.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace umfa {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

enum Prec : int { P_FP16 = 0, P_BF16 = 1, P_FP32 = 2, P_INT8 = 3, P_INT4 = 4 };
// mask kinds after host-side normalisation (mfa_mask_type_t x mfa_mask_scalar_t)
enum MaskKind : int { MK_NONE = 0, MK_BOOL = 1, MK_F32 = 2, MK_F16 = 3, MK_BF16 = 4,
                      MK_WINDOW = 5 /* no tensor: key attends iff row - win_left <= key <= row + win_right */ };

#define UMFA_LOG2E 1.4426950408889634f
#define UMFA_LN2 0.6931471805599453f

// One forward launch.  Strides are in ELEMENTS; head_dim is contiguous (stride 1).
struct FwdParams {
    const void* q;
    const void* k;
    const void* v;
    void* o;            // dense [B,H,Sq,D], element type out_prec
    float* lse;         // optional, [B*H*Sq], natural log
    const void* mask;   // optional
    int64_t qs[4], ks[4], vs[4];  // batch, head, seq, head_dim strides ([3] != 1 only on the exact path)
    int64_t os[2];                // output seq / head_dim strides inside a (b,h) slab (dense: D, 1)
    int64_t ms[4];                // mask strides for (b, h, q, k); 0 = broadcast
    uint32_t B, H, Sq, Skv, D;
    float scale;
    int causal;
    int mask_kind;
    int in_prec;   // P_FP16 / P_BF16 / P_FP32
    int out_prec;  // P_FP16 / P_BF16 / P_FP32
    // split-KV tail (fa_fwd_16): the last `n_items - n_full` items are cut into `nsplit` key ranges, one
    // workgroup each; partial (O, m, l) go through part_buf and the last arriver (part_cnt) combines.
    uint32_t n_full;     // items handled whole (== all items when nsplit <= 1)
    uint32_t nsplit;     // parts per split item (0/1 = no split)
    float* part_buf;     // [split item][part][wave 4][reg 16*NDB+2][lane 64] fp32
    uint32_t* part_cnt;  // [split item] arrival tickets: zero on entry, left zero by the folding workgroup
    // mask tile flags (fa_fwd_16, optional): one byte per (mask batch, mask head, 32-row block, 64-key tile) from
    // mask_flags_kernel -- 0 mixed, 1 every in-range element masked (the tile is skipped), 2 every in-range element
    // attends with a zero term (the tile runs without reading the mask)
    const uint8_t* mask_flags;
    uint32_t mf_bs, mf_hs;         // flag-row strides of batch / head in units of (mf_nrb * mf_ntiles); 0 = broadcast
    uint32_t mf_nrb, mf_ntiles;    // 32-row blocks, 64-key tiles
    uint32_t win_left, win_right;  // MK_WINDOW: sliding window (in-stream entry, mask type 3); tile flags are arithmetic
    // bool mask re-packed for fa_fwd16_w64 (fa_aux.hip mask_pack_kernel): per-lane bit words, the visited-tile list of every 256-row
    // block and its length; slab of (b, h) = b * mk_bs + h * mk_hs (0 = broadcast)
    const uint32_t* mk_bits;
    const uint32_t* mk_list;
    const uint32_t* mk_cnt;
    uint32_t mk_bs, mk_hs, mk_nrb64, mk_T;
    const uint32_t* mk_prefix;  // running sums of the list lengths of the blocks the one-wave-per-SIMD mask kernel shares between workgroups
    // fused rotary embedding of Q (umfa_rope_attention_forward_stream): fp32 tables [Sq, D] (or [B, Sq, D] with
    // rope_tb = Sq * D), pair-duplicated, applied to the Q fragments right after their load; K arrives already rotated
    const float* rope_cos;
    const float* rope_sin;
    int64_t rope_tb;
    // bf16 operands with the P V product in fp16 (option pv_fp16, the default): P is rounded to fp16 -- 11 bits instead of 8 -- and V
    // becomes fp16; the bf16-input forward then meets the 1e-3 bound.  fp16 has five exponent bits where bf16 has eight, so V goes in
    // as V * 2^-e with ONE power of two e per (batch, head) slab and 2^e comes back in the epilogue (exact both ways):
    //   pv16 = 2  `v` points at the fp16 image the runtime's cast pre-pass wrote (fa_aux.hip cast_rows_bf16_f16_kernel: e from the slab's
    //             largest |v|); `vsc` = that pass's header: 2^e of slab (b, h) at vsc[128 * (b * vsc_bs + h * vsc_hs) + 65] (0 strides: broadcast K / V heads)
    //   pv16 = 1  the 128-row kernel converts bf16 -> fp16 on V's way into LDS with e = 0; a workgroup whose outputs show that this was
    //             not enough (non-finite, or all below 2^-11) takes the slab's amax itself and sweeps its keys again with the right e
    // No status word, nothing for the host to read back, the same result under hipGraph replay.
    int pv16;
    const float* vsc;
    uint32_t vsc_bs, vsc_hs;
    // balanced causal pairs (fa_fwd_16_kernel.h CBAL; fwd_16_split_plan decides): a head's q-blocks (i, nqb - 1 - i) are dealt to two
    // workgroups of equal length; part_buf holds one slot per pair, part_cnt one flag word per pair (zero between launches)
    uint32_t decode_form; // 1 = the launch runs the decode form (fa_fwd_16_kernel.h KS = 4: <= 32 query rows, four key quarters per 128-key tile); the plan decides
    uint32_t cbal;        // 1 = the launch runs the CBAL instantiation
    uint32_t cbal_delta;  // key tiles by which a pair's part A is shorter than half (it pays the fold)
    // fp32 additive masks on the one-wave-per-SIMD structure (end of round 6): whether fp16 holds every mask value is known on the DEVICE only
    // (fa_aux.hip mask_classify_f32_kernel, folded by mask_list_kernel), so the call enqueues both routes and `guard` points at the verdict word
    // (0 = every value exact in fp16: the bias kernel on the fp16 copy; 1 = not: the 128-row kernel on the caller's fp32 tensor).  A guarded
    // kernel -- MASKA instantiations of fa_fwd16_w64, HAS_MASK instantiations of fa_fwd16 -- leaves at once unless
    // *guard == guard_want.  NULL: not guarded.  No host read-back, the same under hipGraph replay whatever the mask holds then.
    const uint32_t* guard;
    uint32_t guard_want;
    uint32_t mask_padded;  // 1: `mask` is the classification pass's fp16 copy, padded to whole 64 x 64 tiles with -inf (fa_aux.hip launch_mask_classify) -- a ragged Sq / Skv is fine
};

// Interleaved-pair rotary rotation of 8 consecutive elements (4 pairs) given the 8 table entries of their columns
// (only the even ones are read, MFABridge.swift:264-266).  ONE definition for the stand-alone rotate kernel (fa_aux.hip)
// and for the fused Q load of the forward kernels: both then round identically (the fused path is checked bit-for-bit
// against rotate-then-attend).  Products and sums are pinned (no contraction choice left to the compiler).
__device__ __forceinline__ void rope_rotate8(const float (&x)[8], const f32x4& c0, const f32x4& c1, const f32x4& s0,
                                             const f32x4& s1, bool negate_sin, float (&y)[8]) {
    const float cs[4] = {c0[0], c0[2], c1[0], c1[2]};
    float sn[4] = {s0[0], s0[2], s1[0], s1[2]};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (negate_sin) sn[k] = -sn[k];
        y[2 * k] = __builtin_fmaf(x[2 * k], cs[k], -__fmul_rn(x[2 * k + 1], sn[k]));
        y[2 * k + 1] = __builtin_fmaf(x[2 * k + 1], cs[k], __fmul_rn(x[2 * k], sn[k]));
    }
    // the fp32 results are final here: without this the fp16 instantiations fold the last FMA and the conversion into
    // v_fma_mix{lo,hi}_f16 in one caller and not in the other (seen: ~1e-4 of the elements one fp16 ulp apart)
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(y[j]));
}

// Q^T fragment (8 consecutive head-dim elements of one query row, 16-bit type T) rotated in registers
template <typename T, typename V8>
__device__ __forceinline__ V8 rope_fragment(V8 v, const float* __restrict__ cos_t, const float* __restrict__ sin_t, int64_t at) {
    float x[8], y[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = (float)v[j];
    const f32x4 c0 = *(const f32x4*)(cos_t + at), c1 = *(const f32x4*)(cos_t + at + 4);
    const f32x4 s0 = *(const f32x4*)(sin_t + at), s1 = *(const f32x4*)(sin_t + at + 4);
    rope_rotate8(x, c0, c1, s0, s1, false, y);
    V8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (T)y[j];
    return o;
}

struct BwdParams {
    const void* dout;
    const void* q;
    const void* k;
    const void* v;
    const float* o;
    const float* lse;
    float* dq;
    float* dk;
    float* dv;
    float* dvec;
    float* rowc;    // bwd16 only, internal scratch [2][B*H*Sq]: -LSE * log2(e), then -D (written by bwd16_dq, read by bwd16_dkdv)
    const float* mask;  // optional fp32 additive [B,H,Sq,Skv] (quantised backward only)
    uint32_t B, H, Sq, Skv, D;
    float scale;
    int causal;
    int in_prec;    // q, k, v
    int dout_prec;  // dO
    int grad_in_type;  // 0: dQ, dK, dV are fp32 (the ABI contract); 1: they are written in the input type (in-stream entry)
    int o_in_type;     // 0: O is fp32 (the ABI contract); 1: `o` points at O in the input type (in-stream entry)
    int phases;     // 0 = everything; else bit 0 = D vector, bit 1 = dQ, bit 2 = dK/dV (the pre-quantised ABI's two calls)
    uint32_t Hkv;   // bwd16 only: K / V hold Hkv heads (grouped-query attention, H % Hkv == 0; 0 = H): query head h reads K / V head
                    // h / (H / Hkv) in place (no expanded copies); dK / dV still come out per QUERY head, the caller sums the groups
    int dkdv_fp32;  // bwd16 only: dK / dV in fp32 even when grad_in_type asks for operand-type dQ (they are summed over a group next)
    void* ds;       // bwd16 "dS-store" form (option bwd_ds_store, head_dim 128, non-causal): scratch [B*H][Sq][Skv] in the operand
                    // type -- bwd16_dkdv writes dS = P (dP - D) there, bwd16_dq_gemm computes dQ = scale dS K from it: 5 products
                    // instead of 7 for B H Sq Skv 2 bytes of HBM (805 MB at the FLUX shape); NULL = the two recomputing kernels
    int ds_lab;     // lab (env UMFA_LAB_DS, timing only): bit 0 = every dS store goes to tile 0 of the slab (no HBM write stream)
    const float* units;   // bwd16 only, device, NULL = none: EVERY operand arrives as a power-of-two multiple (mfa_quantized_backward: fp16 images
                          // Q 2^-eq, K 2^-ek, V 2^-ev of the de-quantised operands, dO 2^-edo; fa_aux.hip bwd_units_kernel) --
                          // [0] 2^(eq+ek) onto the softmax scale, [1] 2^(edo+ev+ek) onto dQ, [2] 2^(edo+ev+eq) onto dK, [3] 2^edo onto dV,
                          // [4] 2^-ev onto the D the kernels subtract from dP, [5] 2^edo onto the D vector that leaves, [6] 2^-(edo+ev) onto a
                          // D vector that comes in (bwd16_rowc_kernel).  dS = P (dP - D), rounded to fp16, is then bounded by 8 head_dim.
};

__device__ __forceinline__ float unit_c(const BwdParams& p) { return p.units ? p.units[0] : 1.0f; }
__device__ __forceinline__ float unit_dq(const BwdParams& p) { return p.units ? p.units[1] : 1.0f; }
__device__ __forceinline__ float unit_dk(const BwdParams& p) { return p.units ? p.units[2] : 1.0f; }
__device__ __forceinline__ float unit_dv(const BwdParams& p) { return p.units ? p.units[3] : 1.0f; }
__device__ __forceinline__ float unit_rowd(const BwdParams& p) { return p.units ? p.units[4] : 1.0f; }
__device__ __forceinline__ float unit_dvec(const BwdParams& p) { return p.units ? p.units[5] : 1.0f; }
// e with amax * 2^-e in [1, 2) (amax as fp32 bits); 0 for an all-zero or non-finite tensor
__device__ __forceinline__ int unit_exponent(unsigned amax_bits) {
    if (amax_bits == 0 || amax_bits >= 0x7f800000u) return 0;
    const int E = (int)(amax_bits >> 23);
    const int e = (E ? E : 1) - 127;
    return e < -126 ? -126 : e > 126 ? 126 : e;
}

__device__ __forceinline__ float bf16_bits_to_float(uint16_t b) {
    return __uint_as_float(((uint32_t)b) << 16);
}

__device__ __forceinline__ float load_as_float(const void* p, int64_t idx, int prec) {
    if (prec == P_FP32) return ((const float*)p)[idx];
    if (prec == P_FP16) return (float)((const _Float16*)p)[idx];
    return bf16_bits_to_float(((const uint16_t*)p)[idx]);
}

// Additive mask term in the log2 domain (already multiplied by log2 e); -inf = masked.
// Semantics: MFABridge.swift:193-238 (bool: nonzero attends; bf16: bits << 16).
__device__ __forceinline__ float mask_term(const void* mask, int64_t idx, int kind) {
    switch (kind) {
    case MK_BOOL: return ((const uint8_t*)mask)[idx] != 0 ? 0.0f : -INFINITY;
    case MK_F32: return ((const float*)mask)[idx] * UMFA_LOG2E;
    case MK_F16: return (float)((const _Float16*)mask)[idx] * UMFA_LOG2E;
    case MK_BF16: return bf16_bits_to_float(((const uint16_t*)mask)[idx]) * UMFA_LOG2E;
    default: return 0.0f;
    }
}

// MK_WINDOW term of (row, key)
__device__ __forceinline__ float window_term(uint32_t row, uint32_t key, uint32_t left, uint32_t right) {
    return (key + left >= row && key <= row + right) ? 0.0f : -INFINITY;
}

// key index inside a 32-key block held by accumulator register r of lane-half hi
__device__ __forceinline__ constexpr int acc_row(int r, int hi) {
    return (r & 3) + 8 * (r >> 2) + 4 * hi;
}

__device__ __forceinline__ float xor32(float x) {
    // exchange with lane ^ 32 (the other half of the wave)
    return __shfl_xor(x, 32, 64);
}

// XCD-aware, bijective remap of a linear workgroup id (cdna_hip_programming.md T1):
// blocks with equal id % 8 share an XCD, so give each XCD a contiguous slice of the
// (batch*head, q-block) space and K/V of a head stay in that XCD's L2.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t id, uint32_t n) {
    const uint32_t q = n >> 3, r = n & 7, x = id & 7;
    const uint32_t base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (id >> 3);
}

}  // namespace umfa
