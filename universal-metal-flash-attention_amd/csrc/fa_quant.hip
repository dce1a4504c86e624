// fa_quant.hip -- runtime-quantised attention for gfx950 (SageAttention-style: Q, K AND V quantised).
//
// Replaces, for mfa_quantized_forward_with_lse (MFABridge+Quantized.swift:227-358):
//   * createQuantizedTensorFromBufferPublic x3 (three command-buffer commits in the reference,
//     README.md:127-131) by ONE fused quantiser launch for Q, K and V (block-wise mode), and
//   * QuantizedAttention.forwardMultiHead (INT storage, dequantise-on-load, FP32 math -- AGENTS.md:143-152)
//     by a flash forward whose QK^T runs on the int8 MFMA (v_mfma_i32_32x32x32_i8, 2x the bf16 rate):
//         S = (sum_d q8 k8) * sq[q-block] * sk[k-block]        exact int32 accumulation
//     and whose PV runs on the fp16 MFMA with V de-quantised ONCE by the pre-pass to fp16
//     (q_v * s_v has <= 7 significant bits times a scale: fp16's 11 bits hold it to 2^-12) -- as q_v * s_v * 2^-e, one
//     power of two e per (batch, head) slab found on the device (fp16 has five exponent bits, the caller's V eight:
//     QuantParams::vhdr), with 2^e handed back in the kernels' epilogues.
// Quantiser = the reference's symmetric formula (Tests/QuantizationTests/QuantizationTests.swift:72-128):
//   scale = absmax / 127 (INT8) or / 7 (INT4), q = clamp(round_half_away(x / scale)), zero point 0.
// Block-wise mode (quant_mode 2): one scale per (batch, head, 64 consecutive rows) -- the block edge is
// the kernel's 64-key tile, so a tile has ONE K scale and a wave ONE Q scale.  Tensor-wise mode
// (quant_mode 0): one scale per tensor (the same kernels, every block scale equal).
// INT4 uses the same kernels with values in [-8, 7] (unpacked in the internal workspace; nibble
// packing only matters at rest, which this ABI never exposes).
#include <climits>
#include <cstring>

#include <type_traits>

#include "fa_common.h"
#include "kernels.h"

namespace umfa {

#define LDS_AS __attribute__((address_space(3)))
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));

constexpr int QBLK = 64;  // rows per quantisation block

struct QuantParams {
    const void* src[3];   // q, k, v (input precision)
    int8_t* q8;           // [B*H*Sq][DPQ]
    int8_t* k8;           // [B*H*Skv][DPQ]
    _Float16* v16;        // [B*H*Skv][DPQ] (rows zero-padded like Q/K)
    uint8_t* v8;          // fp8 P V mode (quant_mode 3, D = 128): V as e4m3 [B*H][tile][8192] in MFMA operand order, else NULL
    uint32_t* v_e8;       // ... and the tile's power-of-two scale as an E8M0 byte replicated four times
    float* f32[3];        // optional fake-quantised fp32 copies (backward), [rows][D]
    _Float16* f16[3];     // optional fake-quantised fp16 copies (MFMA backward), [rows][D]
    uint32_t* overflow;   // with f16: set when q * scale does not fit fp16
    float* scale[3];      // per (bh, block)
    uint32_t rows[3];     // Sq, Skv, Skv
    uint32_t nblk[3];     // blocks per (b,h)
    uint32_t BH, D, DPQ;
    int in_prec;
    float qmax;           // 127 or 7
    int qlo, qhi;         // clamp range
    int t_first, t_end;   // the tensors (0 Q, 1 K, 2 V) this launch covers: [t_first, t_end)
    // The fp16 V image holds q * s * 2^-e, one power of two e per (batch, head) slab from the slab's largest |v| (fp16 has five exponent
    // bits, the caller's V has eight: as plain q * s the image was inf from |v| ~ 1e5 and subnormal below ~ 1e-4).  vhdr: the slab headers
    // of the bf16 forward's cast pass (kernels.h VSC_HDR_*: 128 words per slab, zero between launches except the scale word, which is left
    // holding 2^e for the attention kernels' epilogues); NULL = plain q * s (no forward consumes the image: the backward entries).
    // v_exchange 1: quantize_wave_kernel's V workgroups find the slab's amax among themselves (the cast pass's flag-word exchange, four
    // blocks per workgroup); 0: word VSC_HDR_AMAX holds it (vimage_amax_kernel ran before, vimage_finish_kernel runs behind).
    uint32_t* vhdr;
    const uint32_t* famax;  // f16 copies (the backward's operands) as q * s * 2^-e_t, e_t from the tensor's largest magnitude (fp32 bits) in famax[t]; NULL: e_t = 0
    int v_exchange;
    uint32_t wait_ticks;  // bound of the exchange's wait (100 MHz ticks), then the workgroup reads the slab itself
};

// e of the image: the slab's largest |v| (top 16 bits of its fp32 pattern) lands in [2^13, 2^14) -- two binades under fp16's last: q * s can
// exceed the block's amax by a rounding, and in tensor-wise mode (ONE s for the tensor) a slab whose amax is just above s / 2 quantises to
// q = +-1, i.e. q * s up to twice the slab's amax.  inf / NaN count as the largest finite exponent.
__device__ __forceinline__ int vimage_exponent(unsigned amax_b16) {
    if (amax_b16 == 0) return 0;
    int E = (int)(amax_b16 >> 7);
    E = E > 254 ? 254 : E;
    const int e = (E ? E - 127 : -126) - 13;
    return e < -100 ? -100 : e;
}
__device__ __forceinline__ float exp2i(int e) { return __uint_as_float((unsigned)(127 + e) << 23); }

// mode 0: write block absmax only; mode 1: quantise with scales already in scale[]; mode 2: both (fused).
// One workgroup per (tensor, batch*head, 64-row block).  The block (<= 64 x 256 elements) is read ONCE with
// 16-byte loads into registers (8 elements per chunk, <= 8 chunks per thread), reduced to its absmax through
// wave shuffles + 4 LDS words, then quantised from the registers and written with 8-byte (int8 Q/K) or
// 16-byte (fp16 V) stores -- HBM-bound: every input byte is read once (twice in tensor-wise mode).
// fp8 V image (fa_fwd_w64_i8f8, tools/gen_w64_body.py "fp8 variant"): byte offset inside the 8-KiB image of a 64-key tile of
// element (key kk < 64, d < 128).  A-operand order of v_mfma_scale_f32_32x32x64_f8f6f4 matched to the packed P^T:
// k-slot 32 h + j of lane (d % 32, h) is key (j & 3) + 8 ((j >> 2) & 3) + 4 h + 32 (j >> 4); the two 16-byte halves of a
// lane's 32 bytes are stored 1 KiB apart so that each ds_read_b128 of a wave is one contiguous KiB.
__device__ __forceinline__ int v8_off(int kk, int d) {
    const int k32 = kk & 31, h = (k32 >> 2) & 1, b = (k32 & 3) + 4 * (k32 >> 3);
    return (d >> 5) * 2048 + (kk >> 5) * 1024 + (32 * h + (d & 31)) * 16 + b;
}

// a / b for a divisor b whose refined reciprocal r1 = fma(fma(-b, rcp(b), 1), rcp(b), rcp(b)) the caller holds: the
// numerator half of the AMDGPU fp32 division (LLVM LowerFDIV32: div_scale, rcp, fma x2 | mul, fma x3, div_fmas, div_fixup)
// with the scale steps left out.  Those steps are the identity when b is in [2^-60, 2^60] and |a| >= 2^-103 (then no
// operand or quotient is near the denormal range and the exponents are < 96 apart: v_div_scale passes a and b through,
// v_div_fmas is a plain fma, v_div_fixup returns its first operand), so the quotient is bit-identical to `a / b`.  Smaller
// |a| give |a / b| < 2^-43 in either form: the same integer 0 after rounding, which is all the quantiser keeps.
__device__ __forceinline__ float div_by_block_scale(float a, float b, float r1) {
    const float q0 = a * r1;
    const float e1 = __builtin_fmaf(-b, q0, a);
    const float q1 = __builtin_fmaf(e1, r1, q0);
    const float e2 = __builtin_fmaf(-b, q1, a);
    return __builtin_fmaf(e2, r1, q1);
}

// The same five operations on two elements at once (v_pk_mul_f32 / v_pk_fma_f32: the packed forms round each half like the scalar
// ones) followed by the reference's round-half-away and clamp: q = clamp(trunc(y + copysign(nextbelow(0.5), y))).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void quant_pair_fast(float a0, float a1, float b, float r1, int qlo, int qhi, int& q0, int& q1) {
    const f32x2_t a = {a0, a1}, bb = {b, b}, rr = {r1, r1};
    const f32x2_t y0 = a * rr;
    const f32x2_t e1 = __builtin_elementwise_fma(-bb, y0, a);
    const f32x2_t y1 = __builtin_elementwise_fma(e1, rr, y0);
    const f32x2_t e2 = __builtin_elementwise_fma(-bb, y1, a);
    const f32x2_t y = __builtin_elementwise_fma(e2, rr, y1);
    const f32x2_t h = {__builtin_copysignf(0x1.fffffep-2f, y[0]), __builtin_copysignf(0x1.fffffep-2f, y[1])};
    const f32x2_t t = y + h;
    q0 = max(min((int)t[0], qhi), qlo);
    q1 = max(min((int)t[1], qhi), qlo);
}

// IN16: 16-bit inputs stay PACKED in registers (4 per 8-element chunk instead of 8 floats) and are decoded where they are used
// (twice: absmax, then quantise).  The pass is latency-bound, not vector-bound -- round 4 took a quarter of its vector
// instructions out (rounding / clamp / pack) and it stayed at 40 us, round 3 the same with the division -- so registers are
// cheap: 74 instead of 104 registers, 6 instead of 4 resident workgroups per CU.  (It did not move the pass either -- 40.1 us, also
// with 8 workgroups per CU forced at the price of 11 spills: 126 MB at 3.15 TB/s against 4.6 TB/s for the one-phase V cast pass.
// What is left is the two-phase shape itself: load a block, reduce its absmax across the workgroup, only then convert and store.)
template <int MODE, bool IN16>
__global__ __launch_bounds__(256) void quantize_kernel(QuantParams p) {
    __shared__ float red[4];
    __shared__ __attribute__((aligned(16))) unsigned char v8img[8192];
    constexpr int MAXC = 8;  // chunks per thread: 64 rows * (256 / 8) chunks / 256 threads
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    uint32_t id = blockIdx.x;
    int t = p.t_first;
    while (t < p.t_end - 1 && id >= p.BH * p.nblk[t]) { id -= p.BH * p.nblk[t]; ++t; }
    const uint32_t bh = id / p.nblk[t], blk = id % p.nblk[t];
    const uint32_t row0 = blk * QBLK;
    const uint32_t nrows = min((uint32_t)QBLK, p.rows[t] - row0);
    const int64_t base = ((int64_t)bh * p.rows[t] + row0) * p.D;
    const uint32_t cpr = p.D / 8;           // chunks per row (D % 8 == 0)
    const uint32_t nchunks = nrows * cpr;   // <= 2048
    const int tid = threadIdx.x;
    const bool is_bf16 = p.in_prec == P_BF16;

    typename std::conditional<IN16, u32x4, f32x4>::type xr[MAXC][IN16 ? 1 : 2];
    auto X = [&](int c, int j) -> float {  // element j of chunk c
        if constexpr (IN16) {
            const unsigned w = xr[c][0][j >> 1];
            if (is_bf16) return __uint_as_float((j & 1) ? (w & 0xffff0000u) : (w << 16));
            return (float)__builtin_bit_cast(_Float16, (uint16_t)((j & 1) ? (w >> 16) : (w & 0xffffu)));
        } else {
            return xr[c][j >> 2][j & 3];
        }
    };
    float amax = 0.0f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const uint32_t ch = tid + 256 * c;
        if (ch < nchunks) {
            const int64_t e0 = base + (int64_t)ch * 8;
            if constexpr (IN16) {
                xr[c][0] = *(const u32x4*)((const uint16_t*)p.src[t] + e0);
            } else {
                xr[c][0] = *(const f32x4*)((const float*)p.src[t] + e0);
                xr[c][1] = *(const f32x4*)((const float*)p.src[t] + e0 + 4);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const uint32_t ch = tid + 256 * c;
        if (ch < nchunks) {
#pragma unroll
            for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(X(c, j)));
        }
    }
    float sc;
    if (MODE != 1) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
        if ((tid & 63) == 0) red[tid >> 6] = amax;
        __syncthreads();
        amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (MODE == 0) {
            if (tid == 0) p.scale[t][bh * p.nblk[t] + blk] = amax;  // absmax for now; reduced later
            return;
        }
        sc = amax > 0.0f ? amax / p.qmax : 1.0f;
        if (tid == 0) p.scale[t][bh * p.nblk[t] + blk] = sc;
    } else {
        sc = p.scale[t][bh * p.nblk[t] + blk];
    }
    if (MODE == 2 && t == 2 && p.v8) {
        // fp8 P V mode: V / 2^e as e4m3 (round to nearest even), 2^e the smallest power of two with absmax / 2^e <= 448
        // -- a power-of-two scale costs a floating-point format nothing and rides in the MFMA's E8M0 scale operand
        int e = 0;
        if (amax > 0.0f) {
            (void)frexpf(amax * (1.0f / 448.0f), &e);   // amax / 448 = f * 2^e, f in [0.5, 1)  =>  2^e >= amax / 448
            e = e < -126 ? -126 : e;
        }
        const float inv = __builtin_amdgcn_ldexpf(1.0f, -e);
        if (tid == 0) {
            const uint32_t byte = (uint32_t)(e + 127) & 0xffu;
            p.v_e8[bh * p.nblk[2] + blk] = byte * 0x01010101u;
            p.scale[2][bh * p.nblk[2] + blk] = __builtin_amdgcn_ldexpf(1.0f, e);
        }
        *(i32x4*)(v8img + tid * 32) = i32x4{0, 0, 0, 0};        // rows past Skv stay fp8 zero: 0 * P, never NaN * 0
        *(i32x4*)(v8img + tid * 32 + 16) = i32x4{0, 0, 0, 0};
        __syncthreads();
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const uint32_t ch = tid + 256 * c;
            if (ch < nchunks) {
                const int r = (int)(ch / cpr), d0 = (int)(ch % cpr) * 8;
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const unsigned w = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(X(c, j) * inv, X(c, j + 1) * inv, 0, false);
                    v8img[v8_off(r, d0 + j)] = (unsigned char)(w & 0xff);
                    v8img[v8_off(r, d0 + j + 1)] = (unsigned char)((w >> 8) & 0xff);
                }
            }
        }
        __syncthreads();
        uint8_t* dst = p.v8 + ((int64_t)bh * p.nblk[2] + blk) * 8192;
        *(i32x4*)(dst + tid * 16) = *(const i32x4*)(v8img + tid * 16);
        *(i32x4*)(dst + 4096 + tid * 16) = *(const i32x4*)(v8img + 4096 + tid * 16);
        return;
    }
    const int64_t orow0 = (int64_t)bh * p.rows[t] + row0;
    // x / sc with ONE divisor for the whole block: the divisor's half of the IEEE division (reciprocal + one Newton step)
    // is taken once, each element pays the numerator's half only (div_by_block_scale): the same operations in the same
    // order as hipcc's correctly rounded fp32 divide, so the integers stay bit-exact with the oracle
    // (tests/test_gpu_quantized.py checks them).  Outside the exponent range where that divide would not rescale its
    // operands the plain divide runs (uniform per block).
    const float rcp0 = __builtin_amdgcn_rcpf(sc);
    const float rcp1 = __builtin_fmaf(__builtin_fmaf(-sc, rcp0, 1.0f), rcp0, rcp0);
    const bool fast_div = sc >= 0x1p-60f && sc <= 0x1p60f;
    // (V image: q * s * 2^-e, e from the slab's amax that vimage_amax_kernel left in the slab's header)
    const float scv = (t == 2 && p.vhdr) ? sc * exp2i(-vimage_exponent(p.vhdr[VSC_HDR_WORDS * (size_t)bh + VSC_HDR_AMAX])) : sc;
    const float scf = p.famax ? sc * exp2i(-unit_exponent(p.famax[t])) : sc;  // (the backward's fp16 operands: amax 2^-e in [1, 2))
    bool f16_ovf = false;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const uint32_t ch = tid + 256 * c;
        if (ch < nchunks) {
            const uint32_t r = ch / cpr, d0 = (ch % cpr) * 8;
            int q[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xv = X(c, j);
                const float y = fast_div ? div_by_block_scale(xv, sc, rcp1) : xv / sc;  // (quantize_wave_kernel: the packed form of the same five operations)
                // roundf (half away from zero, the reference's .rounded()) as trunc(y + copysign(nextbelow(0.5), y)): one add and
                // the truncating convert instead of trunc / sub / compare / select / add -- bit-identical for |y| < 2^22 (checked
                // exhaustively over every float in [2^-3, 2^9); below that both give 0; |y| <= 127.x here by construction)
                const float half = __builtin_copysignf(0x1.fffffep-2f, y);
                const int qv = (int)(y + half);
                q[j] = max(min(qv, p.qhi), p.qlo);  // v_med3_i32
            }
            if (t < 2) {
                // four int8 per word: two byte permutes + one or (the values are inside int8 already)
                const uint32_t a0 = __builtin_amdgcn_perm((uint32_t)q[1], (uint32_t)q[0], 0x0c0c0400u);  // {q0.b0, q1.b0, 0, 0}
                const uint32_t a1 = __builtin_amdgcn_perm((uint32_t)q[3], (uint32_t)q[2], 0x04000c0cu);  // {0, 0, q2.b0, q3.b0}
                const uint32_t b0 = __builtin_amdgcn_perm((uint32_t)q[5], (uint32_t)q[4], 0x0c0c0400u);
                const uint32_t b1 = __builtin_amdgcn_perm((uint32_t)q[7], (uint32_t)q[6], 0x04000c0cu);
                const uint32_t w0 = a0 | a1, w1 = b0 | b1;
                int8_t* dst = (t == 0 ? p.q8 : p.k8) + (orow0 + r) * p.DPQ + d0;
                *(uint2*)dst = make_uint2(w0, w1);
            } else {
                f16x8 hv;
#pragma unroll
                for (int j = 0; j < 8; ++j) hv[j] = (_Float16)((float)q[j] * scv);
                *(f16x8*)(p.v16 + (orow0 + r) * p.DPQ + d0) = hv;
            }
            if (p.f16[t]) {
                f16x8 hv;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float y = (float)q[j] * scf;
                    f16_ovf |= !(fabsf(y) <= 65504.0f);
                    hv[j] = (_Float16)y;
                }
                *(f16x8*)(p.f16[t] + (orow0 + r) * p.D + d0) = hv;
            }
            if (p.f32[t]) {
                float* f = p.f32[t] + (orow0 + r) * p.D + d0;
                *(f32x4*)f = f32x4{(float)q[0] * sc, (float)q[1] * sc, (float)q[2] * sc, (float)q[3] * sc};
                *(f32x4*)(f + 4) = f32x4{(float)q[4] * sc, (float)q[5] * sc, (float)q[6] * sc, (float)q[7] * sc};
            }
        }
    }
    if (f16_ovf && p.overflow) atomicOr(p.overflow, 1u);
    // zero the row padding of the images (head_dim below the padded 64 / 128 / 256)
    if (p.DPQ > p.D) {
        const uint32_t padc = (p.DPQ - p.D) / 8;
        for (uint32_t e = tid; e < nrows * padc; e += 256) {
            const uint32_t r = e / padc, d0 = p.D + (e % padc) * 8;
            if (t < 2) *(uint2*)((t == 0 ? p.q8 : p.k8) + (orow0 + r) * p.DPQ + d0) = make_uint2(0u, 0u);
            else *(i32x4*)(p.v16 + (orow0 + r) * p.DPQ + d0) = i32x4{0, 0, 0, 0};
        }
    }
}

// The block-wise quantiser of the hot configuration -- 16-bit operands, head_dim 64 / 128, no copies for a backward -- with ONE WAVE
// per 64-row block (round 5).  Same arithmetic, same bits (tests/test_gpu_quantized.py holds both kernels to the oracle's integers);
// what changes is the shape.  quantize_kernel gives a block to a workgroup: 4 loads per thread in flight, then a workgroup barrier
// between the absmax and the conversion -- 126 MB at 3.15 TB/s whatever its instruction or register count was (rounds 3, 4).  The V
// cast pass of the bf16 forward showed what that shape costs (same 16 KB per workgroup: 3.1 TB/s; 64 KB per workgroup: 4.1 TB/s).
// Here a lane holds its block's 16 (head_dim 128) chunks in registers -- every load of the block is in flight before the first use --
// the absmax is six wave shuffles, and nothing in the kernel waits for another wave.
typedef unsigned u32x4_q __attribute__((ext_vector_type(4)));
template <bool BF16> __device__ __forceinline__ float q16_elem(const u32x4_q& x, int j) {
    const unsigned w = x[j >> 1];
    if constexpr (BF16) return __uint_as_float((j & 1) ? (w & 0xffff0000u) : (w << 16));
    return (float)__builtin_bit_cast(_Float16, (uint16_t)((j & 1) ? (w >> 16) : (w & 0xffffu)));
}
// one lane's CPL chunks -> int8 rows (Q / K) or de-quantised fp16 rows (V); FAST: the block's divisor allows the packed division
template <int CPL, bool BF16, bool FAST, bool ISV>
__device__ __forceinline__ void quantize_wave_convert(const u32x4_q (&xr)[CPL], int lane, uint32_t nchunks, float sc, float rcp1, int qlo, int qhi,
                                                      int8_t* __restrict__ dst8, _Float16* __restrict__ dst16, float scv) {
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const uint32_t ch = (uint32_t)lane + 64u * c;
        if (ch < nchunks) {
            int q[8];
            if constexpr (FAST) {
#pragma unroll
                for (int j = 0; j < 8; j += 2) quant_pair_fast(q16_elem<BF16>(xr[c], j), q16_elem<BF16>(xr[c], j + 1), sc, rcp1, qlo, qhi, q[j], q[j + 1]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float y = q16_elem<BF16>(xr[c], j) / sc;
                    const float half = __builtin_copysignf(0x1.fffffep-2f, y);
                    q[j] = max(min((int)(y + half), qhi), qlo);
                }
            }
            if constexpr (!ISV) {
                const uint32_t a0 = __builtin_amdgcn_perm((uint32_t)q[1], (uint32_t)q[0], 0x0c0c0400u);
                const uint32_t a1 = __builtin_amdgcn_perm((uint32_t)q[3], (uint32_t)q[2], 0x04000c0cu);
                const uint32_t b0 = __builtin_amdgcn_perm((uint32_t)q[5], (uint32_t)q[4], 0x0c0c0400u);
                const uint32_t b1 = __builtin_amdgcn_perm((uint32_t)q[7], (uint32_t)q[6], 0x04000c0cu);
                *(uint2*)(dst8 + (int64_t)ch * 8) = make_uint2(a0 | a1, b0 | b1);  // (rows are D = 8 CPL bytes: chunk ch sits at byte 8 ch of the block)
            } else {
                f16x8 hv;
#pragma unroll
                for (int j = 0; j < 8; ++j) hv[j] = (_Float16)((float)q[j] * scv);
                *(f16x8*)(dst16 + (int64_t)ch * 8) = hv;
            }
        }
    }
}

template <int CPL, bool BF16>  // 16-byte chunks per lane: 64 rows * (D / 8) / 64 lanes = D / 8
__global__ __launch_bounds__(256) void quantize_wave_kernel(QuantParams p) {
    const int lane = threadIdx.x & 63;
    uint32_t id = blockIdx.x * 4 + (threadIdx.x >> 6);
    int t = p.t_first;
    while (t < p.t_end - 1 && id >= p.BH * p.nblk[t]) { id -= p.BH * p.nblk[t]; ++t; }
    if (id >= p.BH * p.nblk[t]) return;  // (the last workgroup's spare waves)
    const uint32_t bh = id / p.nblk[t], blk = id % p.nblk[t];
    const uint32_t row0 = blk * QBLK;
    const uint32_t nrows = min((uint32_t)QBLK, p.rows[t] - row0);
    constexpr uint32_t cpr = CPL;           // chunks per row = D / 8 = chunks per lane
    const uint32_t nchunks = nrows * cpr;
    const int64_t base = ((int64_t)bh * p.rows[t] + row0) * (int64_t)(8 * cpr);  // elements: the block's rows are contiguous (dense [rows][D])
    const uint16_t* __restrict__ src = (const uint16_t*)p.src[t] + base;
    u32x4_q xr[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const uint32_t ch = (uint32_t)lane + 64u * c;
        xr[c] = ch < nchunks ? __builtin_nontemporal_load((const u32x4_q*)(src + (int64_t)ch * 8)) : u32x4_q{0, 0, 0, 0};
    }
    float amax = 0.0f;
#pragma unroll
    for (int c = 0; c < CPL; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(q16_elem<BF16>(xr[c], j)));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
    // (the packed inputs are decoded AGAIN for the conversion: without this the decoded floats of the absmax phase stay live across the
    // shuffles -- 128 registers per lane at head_dim 128, two waves per SIMD instead of four)
#pragma unroll
    for (int c = 0; c < CPL; ++c) asm volatile("" : "+v"(xr[c]));
    int vexp = 0;  // V image: q * s * 2^-vexp, one exponent per (batch, head) slab
    if (t == 2 && p.vhdr) {
        uint32_t* const hw = p.vhdr + VSC_HDR_WORDS * (size_t)bh;
        if (p.v_exchange) {
            // The four waves of a workgroup hold four consecutive blocks of ONE slab (the launcher checked the alignment), the slab has
            // <= 64 such workgroups: the V cast pass's exchange (fa_aux.hip cast_rows_body) -- a flag word per workgroup, wave 0 polls the
            // slab's words with one load, a bounded wait, then it reads the slab for the amax itself; the last to leave cleans up.
            __shared__ unsigned wg_amax[4], slab_amax;
            const int wave = threadIdx.x >> 6;
            if (lane == 0) wg_amax[wave] = __float_as_uint(amax) >> 16;
            __syncthreads();
            const uint32_t chunk = blk >> 2, chunks = p.nblk[2] >> 2;
            if (wave == 0) {
                unsigned m = wg_amax[0];
                for (int w = 1; w < 4; ++w) m = wg_amax[w] > m ? wg_amax[w] : m;
                if (lane == 0) __hip_atomic_store(hw + chunk, 0x80000000u | m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint64_t t_in = __builtin_amdgcn_s_memrealtime();
                unsigned f;
                bool served = true;
                for (;;) {
                    f = (uint32_t)lane < chunks ? __hip_atomic_load(hw + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0x80000000u;
                    if (__builtin_amdgcn_ballot_w64((f & 0x80000000u) == 0) == 0) break;
                    if (__builtin_amdgcn_s_memrealtime() - t_in >= p.wait_ticks) { served = false; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                f &= 0x7fffffffu;
                if (!served) {  // nobody promises that the slab's other workgroups are resident: help yourself
                    const uint16_t* __restrict__ s0 = (const uint16_t*)p.src[2] + (int64_t)bh * p.rows[2] * (int64_t)(8 * cpr);
                    const uint32_t total = p.rows[2] * cpr;
                    f = 0;
                    for (uint32_t i = (uint32_t)lane; i < total; i += 64) {
                        const u32x4_q x = *(const u32x4_q*)(s0 + (int64_t)i * 8);
                        float a = 0.0f;
#pragma unroll
                        for (int j = 0; j < 8; ++j) a = fmaxf(a, fabsf(q16_elem<BF16>(x, j)));
                        const unsigned b = __float_as_uint(a) >> 16;
                        f = b > f ? b : f;
                    }
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const unsigned o = (unsigned)__shfl_xor((int)f, off, 64);
                    f = o > f ? o : f;
                }
                if (lane == 0) slab_amax = f;
            }
            __syncthreads();
            vexp = vimage_exponent(slab_amax);
            if (wave == 0) {  // leave: the last workgroup of the slab writes 2^e and zeroes the exchange words (every workgroup has read them by now)
                unsigned d = 0;
                if (lane == 0) d = __hip_atomic_fetch_add(hw + VSC_HDR_DEPART, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                d = (unsigned)__builtin_amdgcn_readfirstlane((int)d);
                if (d == chunks - 1) {
                    if ((uint32_t)lane < chunks) __hip_atomic_store(hw + lane, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (lane == 0) {
                        hw[VSC_HDR_SCALE] = (unsigned)(127 + vexp) << 23;
                        __hip_atomic_store(hw + VSC_HDR_DEPART, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
        } else {
            vexp = vimage_exponent(hw[VSC_HDR_AMAX]);
        }
    }
    const float sc = amax > 0.0f ? amax / p.qmax : 1.0f;
    if (lane == 0) p.scale[t][bh * p.nblk[t] + blk] = sc;
    const float rcp0 = __builtin_amdgcn_rcpf(sc);
    const float rcp1 = __builtin_fmaf(__builtin_fmaf(-sc, rcp0, 1.0f), rcp0, rcp0);
    // wave-uniform choices are scalar branches around whole loops (as per-element selects hipcc emitted both divides and a branch per element)
    const bool fast_div = __builtin_amdgcn_readfirstlane((int)(sc >= 0x1p-60f && sc <= 0x1p60f)) != 0;
    int8_t* const d8 = (t == 0 ? p.q8 : p.k8) + base;
    _Float16* const d16 = p.v16 + base;
    if (t == 2) {
        const float scv = sc * exp2i(-vexp);
        if (fast_div) quantize_wave_convert<CPL, BF16, true, true>(xr, lane, nchunks, sc, rcp1, p.qlo, p.qhi, d8, d16, scv);
        else quantize_wave_convert<CPL, BF16, false, true>(xr, lane, nchunks, sc, rcp1, p.qlo, p.qhi, d8, d16, scv);
    } else {
        if (fast_div) quantize_wave_convert<CPL, BF16, true, false>(xr, lane, nchunks, sc, rcp1, p.qlo, p.qhi, d8, d16, sc);
        else quantize_wave_convert<CPL, BF16, false, false>(xr, lane, nchunks, sc, rcp1, p.qlo, p.qhi, d8, d16, sc);
    }
}

// tensor-wise: reduce block absmax -> one scale, broadcast to every block entry
__global__ __launch_bounds__(256) void tensor_scale_kernel(float* s0, uint32_t n0, float* s1, uint32_t n1, float* s2,
                                                           uint32_t n2, float qmax) {
    __shared__ float red[4];
    float* s = blockIdx.x == 0 ? s0 : blockIdx.x == 1 ? s1 : s2;
    const uint32_t n = blockIdx.x == 0 ? n0 : blockIdx.x == 1 ? n1 : n2;
    float amax = 0.0f;
    for (uint32_t i = threadIdx.x; i < n; i += 256) amax = fmaxf(amax, s[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float sc = amax > 0.0f ? amax / qmax : 1.0f;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 256) s[i] = sc;
}

// Head dims 257 ... 1024 (the reference's callers admit them: metal_sdpa_backend.cpp:1078-1086; its quantised entry has no limit of its own,
// MFABridge+Quantized.swift:227-358): the quantiser's arithmetic per 64-row block -- absmax, s = absmax / qmax, q = clamp(round-half-away(x / s)) --
// as two sweeps over the block in memory (a 64 x 1024 block does not fit a workgroup's registers), leaving q * s as fp32 for the wide fp32
// forward / backward (fa_fwd_wide.hip, fa_bwd_wide.hip).  Correct to the oracle's integers like the register kernels; not tuned.
// MODE as quantize_kernel: 0 block absmax only, 1 scales given, 2 both.
template <int MODE>
__global__ __launch_bounds__(256) void quantize_wide_kernel(QuantParams p) {
    __shared__ float red[4];
    uint32_t id = blockIdx.x;
    int t = p.t_first;
    while (t < p.t_end - 1 && id >= p.BH * p.nblk[t]) { id -= p.BH * p.nblk[t]; ++t; }
    const uint32_t bh = id / p.nblk[t], blk = id % p.nblk[t];
    const uint32_t row0 = blk * QBLK;
    const uint32_t nrows = min((uint32_t)QBLK, p.rows[t] - row0);
    const int64_t base = ((int64_t)bh * p.rows[t] + row0) * p.D;
    const uint32_t n = nrows * p.D;
    float sc;
    if (MODE != 1) {
        float amax = 0.0f;
        for (uint32_t i = threadIdx.x; i < n; i += 256) amax = fmaxf(amax, fabsf(load_as_float(p.src[t], base + i, p.in_prec)));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
        __syncthreads();
        amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (MODE == 0) {
            if (threadIdx.x == 0) p.scale[t][bh * p.nblk[t] + blk] = amax;
            return;
        }
        sc = amax > 0.0f ? amax / p.qmax : 1.0f;
        if (threadIdx.x == 0) p.scale[t][bh * p.nblk[t] + blk] = sc;
    } else {
        sc = p.scale[t][bh * p.nblk[t] + blk];
    }
    float* const dst = p.f32[t] + base;
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const float y = load_as_float(p.src[t], base + i, p.in_prec) / sc;  // (the correctly rounded divide the register kernels reproduce)
        const int qv = max(min((int)(y + __builtin_copysignf(0x1.fffffep-2f, y)), p.qhi), p.qlo);
        dst[i] = (float)qv * sc;
    }
}

// ------------------------------------------------------------------ int8 QK^T forward
template <int DP> __device__ __forceinline__ constexpr int k8_off(int row, int ch) {  // rows of DP bytes, 16-B chunks
    int sw = DP >= 256 ? (ch ^ (row & 15)) : DP == 128 ? (ch ^ ((row >> 1) & 7)) : (ch ^ ((row >> 2) & 3));
    return row * DP + 16 * sw;
}
template <int DP> __device__ __forceinline__ constexpr int v16_off(int row, int ch) {  // rows of 2*DP bytes
    int sw = DP >= 128 ? (ch ^ ((row & 3) << 2)) : (ch ^ (((row >> 1) & 1) << 2));
    return row * (2 * DP) + 16 * sw;
}

struct I8FwdParams {
    const int8_t* q8;
    const int8_t* k8;
    const _Float16* v16;
    const float* q_scale;
    const float* k_scale;
    float* o;
    float* lse;
    const void* mask;   // NULL, or a mask of kind mask_kind with element strides ms[] over (batch, head, row, key) -- 0 = broadcast.  The reference ABI's
                        // form is MK_F32 dense [B,H,Sq,Skv]; umfa_quantized_forward_masked_stream hands over what the caller has (a bool [1,1,Sq,Skv]
                        // stays 1 byte per element instead of becoming 4 B H bytes)
    int mask_kind;
    int64_t ms[4];
    uint32_t mf_bs, mf_hs;  // tile-flag slab strides of batch / head (0 = broadcast), as FwdParams
    // tile flags of the mask (fa_aux.hip mask_flags_kernel, as for fa_fwd16): one byte per (b, h, 32-row block, 64-key tile) -- 1: every element
    // masked (the wave skips the tile), 2: every element attends with a zero term (the tile runs without reading the mask), 0: mixed
    const uint8_t* mask_flags;
    uint32_t mf_nrb, mf_ntiles;
    uint32_t B, H, Sq, Skv, D;
    uint32_t nqblk, nkblk;
    float scale;
    const float* vsc;  // slab headers (kernels.h VSC_HDR_*): the V image is q * s * 2^-e, word VSC_HDR_SCALE of slab bh holds 2^e; NULL: plain q * s
};

template <int DP, bool CAUSAL, bool HAS_MASK, int BN>
__global__ __launch_bounds__(256, (DP > 128 ? 1 : (BN == 32 ? 3 : 2))) void fa_fwd_i8_kernel(I8FwdParams p) {
    constexpr int BM = 128;
    constexpr int NKB = BN / 32, NST = BN / 16;
    constexpr int NKS = DP / 32;             // int8 k-steps (32 per MFMA)
    constexpr int NDB = DP / 32;             // 32-row blocks of O^T
    constexpr int KT_BYTES = BN * DP;        // int8 K tile
    constexpr int VT_BYTES = BN * DP * 2;    // fp16 V tile
    constexpr int KCH = DP / 16, VCH = DP / 8;
    constexpr int NPK = KT_BYTES / 1024, NPV = VT_BYTES / 1024;  // 1-KiB LDS-DMA pieces per tile

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Kbuf = smem;                  // 2 x KT_BYTES
    char* const Vbuf = smem + 2 * KT_BYTES;   // 2 x VT_BYTES

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, ql = lane & 31, hi = lane >> 5;
    const uint32_t nqb = (p.Sq + BM - 1) / BM;
    const uint32_t vid = xcd_remap(blockIdx.x, nqb * p.B * p.H);
    uint32_t bh = vid / nqb;
    uint32_t qb = vid % nqb;
    if (CAUSAL) {
        qb = nqb - 1 - qb;
        // short causal launches: one long and one short item of a mirrored pair per CU (fa_fwd_16_kernel.h, where
        // the dispatcher's co-location rule -- XCD-local workgroups 32 apart -- was measured)
        if ((nqb & 1) == 0 && DP <= 128 && (uint64_t)nqb * nqb * p.B * p.H <= 32768) {
            const uint32_t pi = vid >> 1, h2 = nqb >> 1, j = pi % h2;
            bh = pi / h2;
            qb = (((vid & 1) ^ (vid >> 5)) & 1) ? j : nqb - 1 - j;
        }
    }
    const uint32_t q_row = qb * BM + wave * 32 + ql;
    const uint32_t wave_q0 = qb * BM + wave * 32;
    const int D = (int)p.D;

    const int8_t* __restrict__ qp = p.q8 + (int64_t)bh * p.Sq * DP;
    const int8_t* kp = p.k8 + (int64_t)bh * p.Skv * DP;
    const _Float16* vp = p.v16 + (int64_t)bh * p.Skv * DP;

    i32x4 qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        if (q_row < p.Sq) qf[ks] = *(const i32x4*)(qp + (int64_t)q_row * DP + 32 * ks + 16 * hi);
        else qf[ks] = i32x4{0, 0, 0, 0};
    }
    const uint32_t wq_blk = wave_q0 / QBLK;
    const float sq = (wave_q0 < p.Sq ? p.q_scale[bh * p.nqblk + wq_blk] : 0.0f) * p.scale * UMFA_LOG2E;

    // ---- LDS-DMA staging (the quantiser's images have rows of exactly DP / 2 DP bytes, zero-padded): piece n
    // of a tile image is 1 KiB = rows n*RPI...; lane l lands in row l / CH, chunk slot l % CH and fetches the
    // source chunk slot ^ swz(row).  Inline asm: see fa_fwd_16_kernel.h (hipcc would drain vmcnt before LDS reads).
    const int uw = __builtin_amdgcn_readfirstlane(wave);
    auto make_srd = [&](const void* base, uint32_t bytes) {
        const unsigned long long a = (unsigned long long)base;
        i32x4 d;
        d[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
        d[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
        d[2] = __builtin_amdgcn_readfirstlane((int)bytes);
        d[3] = 0x00020000;
        return d;
    };
    const i32x4 k_srd = make_srd(kp, p.Skv * (uint32_t)DP);
    const i32x4 v_srd = make_srd(vp, p.Skv * (uint32_t)DP * 2);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((LDS_AS char*)smem));
    constexpr int KRPI = 1024 / DP, VRPI = 1024 / (2 * DP);  // rows per piece
    const int k_r = lane / KCH, k_c = lane % KCH, v_r = lane / VCH, v_c = lane % VCH;
#pragma unroll
    for (int i = 0; i < (2 * KT_BYTES + 2 * VT_BYTES) / 4096; ++i) *(i32x4*)(smem + i * 4096 + tid * 16) = i32x4{0, 0, 0, 0};
    __syncthreads();
    auto stage_load = [&](uint32_t t) {
        const int buf = t & 1;
#pragma unroll
        for (int n0 = 0; n0 < NPK; n0 += 4) {
            const int n = n0 + uw;
            if (n < NPK) {
                const int row = n * KRPI + k_r;
                const int voff = (int)(t * BN + row) * DP + (k8_off<DP>(row, k_c) - row * DP);
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                             ::"s"(lds0 + buf * KT_BYTES + n * 1024), "v"(voff), "s"(k_srd) : "memory");
            }
        }
#pragma unroll
        for (int n0 = 0; n0 < NPV; n0 += 4) {
            const int n = n0 + uw;
            if (n < NPV) {
                const int row = n * VRPI + v_r;
                const int voff = (int)(t * BN + row) * (2 * DP) + (v16_off<DP>(row, v_c) - row * (2 * DP));
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                             ::"s"(lds0 + 2 * KT_BYTES + buf * VT_BYTES + n * 1024), "v"(voff), "s"(v_srd) : "memory");
            }
        }
    };
    // (the builtin repeats the wait for hipcc's scoreboard: see fa_fwd_16_kernel.h stage_write)
    auto stage_write = [&](int) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_waitcnt(0x0F70); };

    uint32_t ntiles = (p.Skv + BN - 1) / BN;
    if (CAUSAL) {
        const uint32_t lim = (qb * BM + BM + BN - 1) / BN;
        ntiles = ntiles < lim ? ntiles : lim;
    }
    f32x16 acc[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    float m = -INFINITY, l = 0.0f;
    const int tr_qq = (lane >> 2) & 3, tr_pp = lane & 3, tr_g1 = (lane >> 4) & 1;
    const int64_t mrow = HAS_MASK ? (int64_t)(bh / p.H) * p.ms[0] + (int64_t)(bh % p.H) * p.ms[1] + (int64_t)q_row * p.ms[2] : 0;
    // fp32 rows, contiguous and 16-byte aligned: four keys per load
    const bool mvec = HAS_MASK && p.mask_kind == MK_F32 && p.ms[3] == 1 && (p.Skv & 3u) == 0 && ((uintptr_t)p.mask & 15) == 0 &&
                      ((p.ms[0] | p.ms[1] | p.ms[2]) & 3) == 0;
    // The reference ABI hands the quantised entry a DENSE fp32 [B,H,Sq,Skv] mask (MFABridge+Quantized.swift:227-358): 4.3 GB at config 4.
    // Read per score it came in at 0.76-0.96 TB/s (each lane its own row: a quarter of every sector used) and the masked call took 10-14 x
    // the unmasked one.  The tile-flag pre-pass reads it ONCE at streaming rate; a 0 / -inf mask then costs this kernel no mask read at all.
    const uint8_t* mf_row = nullptr;
    int mf_reg = 0;
    if (HAS_MASK && p.mask_flags && BN == 64 && wave_q0 / 32 < p.mf_nrb)
        mf_row = p.mask_flags + (((uint64_t)(bh / p.H) * p.mf_bs + (uint64_t)(bh % p.H) * p.mf_hs) * p.mf_nrb + wave_q0 / 32) * p.mf_ntiles;

    stage_load(0);
    stage_write(0);
    __syncthreads();

    for (uint32_t t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        const bool more = t + 1 < ntiles;
        if (more) stage_load(t + 1);
        const char* Kt = Kbuf + cur * KT_BYTES;
        const char* Vt = Vbuf + cur * VT_BYTES;
        const uint32_t key_base = t * BN;
        bool active = !CAUSAL || key_base <= wave_q0 + 31;
        int mflag = 0;
        if (HAS_MASK && mf_row) {
            if (t == 0 || (t & 63) == 0) {
                const uint32_t t64 = t & ~63u;
                mf_reg = t64 + lane < p.mf_ntiles ? (int)mf_row[t64 + lane] : 0;
            }
            mflag = __builtin_amdgcn_readlane(mf_reg, (int)(t & 63));
            active = active && mflag != 1;
        }
        if (active) {
            const float ct = sq * p.k_scale[bh * p.nkblk + key_base / QBLK];  // dequant * softmax scale * log2e, >= 0
            i32x16 s[NKB];
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] = 0;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const i32x4 a = *(const i32x4*)(Kt + k8_off<DP>(32 * kb + ql, 2 * ks + hi));
                    s[kb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, qf[ks], s[kb], 0, 0, 0);
                }
            }
            const bool edge = (key_base + BN > p.Skv) || (CAUSAL && key_base + BN - 1 > wave_q0);
            float m_use, m_new;
            float rs = 0.0f;
            f16x8 pf[NST];
            // (tiles the flags call fully open run the unmasked body below: with an all-true mask the masked instantiation took 0.84 ms where the unmasked
            // kernel takes 0.59 at config 4 -- per-score float maxima and the log2-domain detour for tiles that have no mask term)
            if (HAS_MASK && mflag != 2) {
                // additive mask: scores go to the log2 domain first
                float tv[NKB][16];
                float mx = -INFINITY;
                if (mvec) {
                    // registers 4g .. 4g+3 of a 32-key block are keys 8g + 4hi + 0..3: ONE 16-byte load of the mask row (round 5; as sixteen
                    // 4-byte loads per block, each touching 64 different cache lines, the reference ABI's dense fp32 mask -- 4.3 GB at
                    // config 4 -- came in at 0.76 TB/s and the masked call took 14 x the unmasked one)
#pragma unroll
                    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const uint32_t key0 = key_base + 32 * kb + 8 * g + 4 * hi;
                            f32x4 w = {0.0f, 0.0f, 0.0f, 0.0f};
                            if (mflag != 2 && key0 < p.Skv && q_row < p.Sq) w = __builtin_nontemporal_load((const f32x4*)((const float*)p.mask + mrow + key0));
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int r = 4 * g + e;
                                float x = __builtin_fmaf(w[e], UMFA_LOG2E, (float)s[kb][r] * ct);
                                if (edge && (key0 + e >= p.Skv || (CAUSAL && key0 + e > q_row))) x = -INFINITY;
                                tv[kb][r] = x;
                                mx = fmaxf(mx, x);
                            }
                        }
                } else {
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const uint32_t key = key_base + 32 * kb + acc_row(r, hi);
                        float x = (float)s[kb][r] * ct;
                        if (mflag != 2 && key < p.Skv && q_row < p.Sq) x += mask_term(p.mask, mrow + (int64_t)key * p.ms[3], p.mask_kind);
                        if (edge && (key >= p.Skv || (CAUSAL && key > q_row))) x = -INFINITY;
                        tv[kb][r] = x;
                        mx = fmaxf(mx, x);
                    }
                }
                mx = fmaxf(mx, xor32(mx));
                m_new = fmaxf(m, mx);
                m_use = m_new == -INFINITY ? 0.0f : m_new;
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float e = __builtin_amdgcn_exp2f(tv[kb][r] - m_use);
                        rs += e;
                        pf[2 * kb + (r >> 3)][r & 7] = (_Float16)e;
                    }
            } else {
                // integer row max, then ONE conversion + fma per score: p = exp2(float(s) * ct - m)   (ct > 0)
                if (edge) {
#pragma unroll
                    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const uint32_t key = key_base + 32 * kb + acc_row(r, hi);
                            if (key >= p.Skv || (CAUSAL && key > q_row)) s[kb][r] = INT_MIN / 2;  // exp2 -> 0, never the max
                        }
                }
                int mxi = s[0][0];
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mxi = max(mxi, s[kb][r]);
                float mx = (float)mxi * ct;
                mx = fmaxf(mx, xor32(mx));
                m_new = fmaxf(m, mx);
                m_use = m_new;
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float e = __builtin_amdgcn_exp2f(__builtin_fmaf((float)s[kb][r], ct, -m_use));
                        rs += e;
                        pf[2 * kb + (r >> 3)][r & 7] = (_Float16)e;
                    }
            }
            if (!__all(m_new == m)) {
                const float alpha = __builtin_amdgcn_exp2f(m - m_use);
                l *= alpha;
#pragma unroll
                for (int i = 0; i < NDB; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] *= alpha;
                m = m_new;
            }
            l += rs;
#pragma unroll
            for (int i = 0; i < NDB; ++i)
#pragma unroll
                for (int st = 0; st < NST; ++st) {
                    const int row0 = 16 * st + 4 * hi + tr_qq;
                    const int ch = 4 * i + 2 * tr_g1 + (tr_pp >> 1);
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 LDS_AS*)(Vt + v16_off<DP>(row0, ch) + 8 * (tr_pp & 1)));
                    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 LDS_AS*)(Vt + v16_off<DP>(row0 + 8, ch) + 8 * (tr_pp & 1)));
                    const f16x8 a = __builtin_shufflevector(__builtin_bit_cast(f16x4_t, lo), __builtin_bit_cast(f16x4_t, hi4), 0, 1, 2, 3, 4, 5, 6, 7);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pf[st], acc[i], 0, 0, 0);
                }
        }
        if (more) stage_write(cur ^ 1);
        __syncthreads();
    }

    const float lt = l + xor32(l);
    const float inv = lt > 0.0f ? (1.0f / lt) * (p.vsc ? p.vsc[VSC_HDR_WORDS * (size_t)bh + VSC_HDR_SCALE] : 1.0f) : 0.0f;  // (2^e of the V image comes back here)
    if (q_row < p.Sq) {
        float* __restrict__ op = p.o + ((int64_t)bh * p.Sq + q_row) * D;
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = 32 * i + 8 * g + 4 * hi;
                if (d0 < D) {
                    f32x4 val = {acc[i][4 * g] * inv, acc[i][4 * g + 1] * inv, acc[i][4 * g + 2] * inv, acc[i][4 * g + 3] * inv};
                    *(f32x4*)(op + d0) = val;
                }
            }
        if (p.lse && hi == 0) p.lse[(int64_t)bh * p.Sq + q_row] = lt > 0.0f ? (m + log2f(lt)) * UMFA_LN2 : -INFINITY;
    }
}

// ---- V image exponent when the quantiser's own workgroups cannot exchange it (tensor-wise mode, fp32 operands, head dims without the
// wave kernel, slabs of more than 64 x 4 blocks): the slab's amax first, into word VSC_HDR_AMAX of its header ...
template <int PREC>
__global__ __launch_bounds__(256) void vimage_amax_kernel(const void* __restrict__ src, int64_t slab8, uint32_t chunks, uint32_t* __restrict__ hdr) {
    __shared__ unsigned wmax[4];
    const uint32_t bh = blockIdx.x / chunks, chunk = blockIdx.x - bh * chunks;
    constexpr int U = 8;
    unsigned amax = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t i = ((int64_t)chunk * U + u) * 256 + threadIdx.x;
        if (i < slab8) {
            float a = 0.0f;
            if constexpr (PREC == P_FP32) {
                const f32x4 lo = ((const f32x4*)src)[2 * ((int64_t)bh * slab8 + i)], hi = ((const f32x4*)src)[2 * ((int64_t)bh * slab8 + i) + 1];
#pragma unroll
                for (int j = 0; j < 4; ++j) a = fmaxf(a, fmaxf(fabsf(lo[j]), fabsf(hi[j])));
            } else {
                const u32x4_q x = ((const u32x4_q*)src)[(int64_t)bh * slab8 + i];
#pragma unroll
                for (int j = 0; j < 8; ++j) a = fmaxf(a, fabsf(PREC == P_BF16 ? q16_elem<true>(x, j) : q16_elem<false>(x, j)));
            }
            const unsigned b = __float_as_uint(a) >> 16;
            amax = b > amax ? b : amax;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)amax, off, 64);
        amax = o > amax ? o : amax;
    }
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) amax = wmax[w] > amax ? wmax[w] : amax;
        if (amax) (void)__hip_atomic_fetch_max(hdr + VSC_HDR_WORDS * (size_t)bh + VSC_HDR_AMAX, amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// ... and behind the quantiser: 2^e for the attention kernel's epilogue, the amax word back to zero
__global__ __launch_bounds__(256) void vimage_finish_kernel(uint32_t* __restrict__ hdr, uint32_t slabs) {
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s >= slabs) return;
    uint32_t* const hw = hdr + VSC_HDR_WORDS * (size_t)s;
    hw[VSC_HDR_SCALE] = (unsigned)(127 + vimage_exponent(hw[VSC_HDR_AMAX])) << 23;
    hw[VSC_HDR_AMAX] = 0;
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
static inline uint32_t dp_of(uint32_t D) { return D <= 64 ? 64 : D <= 128 ? 128 : 256; }

struct WsLayout {
    size_t q8, k8, v16, sq, sk, sv, f32q, f32k, f32v, mflags, total;
};
static WsLayout ws_layout(uint32_t B, uint32_t H, uint32_t Sq, uint32_t Skv, uint32_t D, bool want_f32) {
    const size_t BH = (size_t)B * H, DP = dp_of(D);
    const size_t nqb = (Sq + QBLK - 1) / QBLK, nkb = (Skv + QBLK - 1) / QBLK;
    WsLayout w;
    size_t off = 0;
    if (D > 256) {  // the wide path (quantize_wide_kernel): block scales and the fp32 images q * s only
        w.q8 = w.k8 = w.v16 = 0;
        w.sq = off; off = align256(off + BH * nqb * 4);
        w.sk = off; off = align256(off + BH * nkb * 4);
        w.sv = off; off = align256(off + BH * nkb * 4);
        w.f32q = off; off = align256(off + BH * Sq * D * 4);
        w.f32k = off; off = align256(off + BH * Skv * D * 4);
        w.f32v = off; off = align256(off + BH * Skv * D * 4);
        w.mflags = off;
        w.total = off;
        return w;
    }
    w.q8 = off; off = align256(off + BH * Sq * DP);
    w.k8 = off; off = align256(off + BH * Skv * DP);
    w.v16 = off; off = align256(off + BH * Skv * DP * 2);
    w.sq = off; off = align256(off + BH * nqb * 4);
    w.sk = off; off = align256(off + BH * nkb * 4);
    w.sv = off; off = align256(off + BH * nkb * 4);
    off = align256(off + BH * nkb * 4);  // fp8 P V mode: the V tiles' E8M0 scale words behind the fp32 scales
    w.f32q = w.f32k = w.f32v = 0;
    if (want_f32) {
        w.f32q = off; off = align256(off + BH * Sq * D * 4);
        w.f32k = off; off = align256(off + BH * Skv * D * 4);
        w.f32v = off; off = align256(off + BH * Skv * D * 4);
    }
    w.mflags = off; off = align256(off + BH * ((Sq + 31) / 32) * ((Skv + 63) / 64));  // tile flags of a mask (one byte per 32 x 64 tile)
    w.total = off;
    return w;
}

size_t quant_workspace_bytes(uint32_t B, uint32_t H, uint32_t Sq, uint32_t Skv, uint32_t D, bool want_f32) {
    return ws_layout(B, H, Sq, Skv, D, want_f32).total;
}

bool quantized_supported(uint32_t D) { return D >= 8 && D % 8 == 0 && D <= 1024; }  // and softmax_scale > 0  (257 ... 1024: the wide fp32 path)

// head dims 257 ... 1024: q * s of the three operands as fp32 images in the workspace (views->qf / kf / vf), per-block scales beside them
static hipError_t launch_quantize_wide(const void* q, const void* k, const void* v, int in_prec, uint32_t B, uint32_t H, uint32_t Sq, uint32_t Skv,
                                       uint32_t D, int bits, int quant_mode, void* workspace, QuantViews* views, hipStream_t stream) {
    const WsLayout w = ws_layout(B, H, Sq, Skv, D, true);
    char* ws = (char*)workspace;
    QuantParams qp;
    memset(&qp, 0, sizeof(qp));
    qp.src[0] = q; qp.src[1] = k; qp.src[2] = v;
    qp.scale[0] = (float*)(ws + w.sq); qp.scale[1] = (float*)(ws + w.sk); qp.scale[2] = (float*)(ws + w.sv);
    qp.f32[0] = (float*)(ws + w.f32q); qp.f32[1] = (float*)(ws + w.f32k); qp.f32[2] = (float*)(ws + w.f32v);
    qp.rows[0] = Sq; qp.rows[1] = Skv; qp.rows[2] = Skv;
    for (int t = 0; t < 3; ++t) qp.nblk[t] = (qp.rows[t] + QBLK - 1) / QBLK;
    qp.BH = B * H; qp.D = D; qp.DPQ = D;
    qp.in_prec = in_prec;
    qp.qmax = bits == 4 ? 7.0f : 127.0f;
    qp.qlo = bits == 4 ? -8 : -128;
    qp.qhi = bits == 4 ? 7 : 127;
    qp.t_first = 0; qp.t_end = 3;
    const uint32_t grid = qp.BH * (qp.nblk[0] + qp.nblk[1] + qp.nblk[2]);
    if (quant_mode == 2 || quant_mode == 3) {
        hipLaunchKernelGGL(quantize_wide_kernel<2>, dim3(grid), dim3(256), 0, stream, qp);
    } else {
        hipLaunchKernelGGL(quantize_wide_kernel<0>, dim3(grid), dim3(256), 0, stream, qp);
        hipLaunchKernelGGL(tensor_scale_kernel, dim3(3), dim3(256), 0, stream, qp.scale[0], qp.BH * qp.nblk[0], qp.scale[1], qp.BH * qp.nblk[1],
                           qp.scale[2], qp.BH * qp.nblk[2], qp.qmax);
        hipLaunchKernelGGL(quantize_wide_kernel<1>, dim3(grid), dim3(256), 0, stream, qp);
    }
    if (views) {
        memset(views, 0, sizeof(*views));
        views->q_scale = qp.scale[0]; views->k_scale = qp.scale[1]; views->v_scale = qp.scale[2];
        views->qf = qp.f32[0]; views->kf = qp.f32[1]; views->vf = qp.f32[2];
        views->nqblk = qp.nblk[0]; views->nkblk = qp.nblk[1]; views->dpq = D;
    }
    return hipGetLastError();
}

hipError_t launch_quantize(const void* q, const void* k, const void* v, int in_prec, uint32_t B, uint32_t H,
                           uint32_t Sq, uint32_t Skv, uint32_t D, int bits, int quant_mode, void* workspace,
                           int copies, QuantViews* views, hipStream_t stream, uint32_t* overflow, uint32_t* vhdr, const uint32_t* famax) {
    if (D > 256) {  // head dims 257 ... 1024: fp32 images only (the fp32 engines); the fp16-copy form belongs to the 16-bit backward (head_dim <= 256)
        if (copies == 2) return hipErrorInvalidValue;
        return launch_quantize_wide(q, k, v, in_prec, B, H, Sq, Skv, D, bits, quant_mode, workspace, views, stream);
    }
    const bool want_f32 = copies != 0;  // the fp16 copies live in the (twice as large) fp32 regions
    const WsLayout w = ws_layout(B, H, Sq, Skv, D, want_f32);
    char* ws = (char*)workspace;
    QuantParams qp;
    memset(&qp, 0, sizeof(qp));
    qp.src[0] = q; qp.src[1] = k; qp.src[2] = v;
    qp.q8 = (int8_t*)(ws + w.q8);
    qp.k8 = (int8_t*)(ws + w.k8);
    qp.v16 = (_Float16*)(ws + w.v16);
    qp.scale[0] = (float*)(ws + w.sq);
    qp.scale[1] = (float*)(ws + w.sk);
    qp.scale[2] = (float*)(ws + w.sv);
    if (copies == 1) {
        qp.f32[0] = (float*)(ws + w.f32q);
        qp.f32[1] = (float*)(ws + w.f32k);
        qp.f32[2] = (float*)(ws + w.f32v);
    } else if (copies == 2) {
        qp.f16[0] = (_Float16*)(ws + w.f32q);
        qp.f16[1] = (_Float16*)(ws + w.f32k);
        qp.f16[2] = (_Float16*)(ws + w.f32v);
        qp.overflow = overflow;
        qp.famax = famax;
    }
    qp.rows[0] = Sq; qp.rows[1] = Skv; qp.rows[2] = Skv;
    for (int t = 0; t < 3; ++t) qp.nblk[t] = (qp.rows[t] + QBLK - 1) / QBLK;
    qp.BH = B * H; qp.D = D; qp.DPQ = dp_of(D);
    qp.in_prec = in_prec;
    qp.qmax = bits == 4 ? 7.0f : 127.0f;
    qp.qlo = bits == 4 ? -8 : -128;
    qp.qhi = bits == 4 ? 7 : 127;
    const uint32_t grid = qp.BH * (qp.nblk[0] + qp.nblk[1] + qp.nblk[2]);
    const bool f8v = quant_mode == 3 && D == 128 && bits == 8 && !want_f32;  // fp8 P V mode: block-wise Q / K + the fp8 V image
    if (f8v) {
        qp.v8 = (uint8_t*)(ws + w.v16);       // the fp16 V region holds the (half as large) fp8 image instead
        qp.v_e8 = (uint32_t*)(ws + w.sv + (size_t)align256(qp.BH * qp.nblk[2] * 4));
    }
    if (quant_mode == 3) quant_mode = 2;
    const bool in16 = in_prec != P_FP32;
    qp.t_first = 0; qp.t_end = 3;
    const bool wave_form = quant_mode == 2 && in16 && !want_f32 && !f8v && (D == 128 || D == 64) && !tuning().quant_block_wg.load(std::memory_order_relaxed) &&
                           ((uintptr_t)q & 15) == 0 && ((uintptr_t)k & 15) == 0 && ((uintptr_t)v & 15) == 0;
    // the fp16 V image as q * s * 2^-e (QuantParams::vhdr): the wave kernel's V workgroups exchange the slab's amax among themselves when
    // they are whole (four blocks of one slab each) and <= 64 per slab; everything else gets the amax from a pass of its own
    bool v_prepass = false;
    if (vhdr && !f8v) {
        qp.vhdr = vhdr;
        qp.wait_ticks = (uint32_t)std::min<int64_t>(std::max(tuning().cast_wait_us.load(std::memory_order_relaxed), 0), 1000000) * 100u;
        qp.v_exchange = wave_form && qp.nblk[2] % 4 == 0 && qp.nblk[2] / 4 <= 64 && ((uint64_t)qp.BH * (qp.nblk[0] + qp.nblk[1])) % 4 == 0 &&
                        !tuning().cast_two_pass.load(std::memory_order_relaxed);
        if (!qp.v_exchange) {
            v_prepass = true;
            const int64_t slab8 = (int64_t)Skv * D / 8;
            const uint32_t chunks = (uint32_t)((slab8 + 8 * 256 - 1) / (8 * 256));
            const dim3 g(qp.BH * chunks);
            if (in_prec == P_FP32) hipLaunchKernelGGL(vimage_amax_kernel<P_FP32>, g, dim3(256), 0, stream, v, slab8, chunks, vhdr);
            else if (in_prec == P_BF16) hipLaunchKernelGGL(vimage_amax_kernel<P_BF16>, g, dim3(256), 0, stream, v, slab8, chunks, vhdr);
            else hipLaunchKernelGGL(vimage_amax_kernel<P_FP16>, g, dim3(256), 0, stream, v, slab8, chunks, vhdr);
        }
    }
    if (wave_form) {
        // the hot configuration: one wave per block (quantize_wave_kernel).  (The fp8 P V mode keeps the workgroup form for all three
        // tensors: its V image needs the workgroup's LDS transpose, and a second launch for Q / K alone cost 4 us of a 165-us call.)
        QuantParams qw = qp;
        uint32_t nb = 0;
        for (int t = 0; t < qw.t_end; ++t) nb += qp.BH * qp.nblk[t];
        const dim3 gw((nb + 3) / 4);
        if (D == 128 && in_prec == P_BF16) hipLaunchKernelGGL((quantize_wave_kernel<16, true>), gw, dim3(256), 0, stream, qw);
        else if (D == 128) hipLaunchKernelGGL((quantize_wave_kernel<16, false>), gw, dim3(256), 0, stream, qw);
        else if (in_prec == P_BF16) hipLaunchKernelGGL((quantize_wave_kernel<8, true>), gw, dim3(256), 0, stream, qw);
        else hipLaunchKernelGGL((quantize_wave_kernel<8, false>), gw, dim3(256), 0, stream, qw);
    } else if (quant_mode == 2) {
        if (in16) hipLaunchKernelGGL((quantize_kernel<2, true>), dim3(grid), dim3(256), 0, stream, qp);
        else hipLaunchKernelGGL((quantize_kernel<2, false>), dim3(grid), dim3(256), 0, stream, qp);
    } else {
        if (in16) hipLaunchKernelGGL((quantize_kernel<0, true>), dim3(grid), dim3(256), 0, stream, qp);
        else hipLaunchKernelGGL((quantize_kernel<0, false>), dim3(grid), dim3(256), 0, stream, qp);
        hipLaunchKernelGGL(tensor_scale_kernel, dim3(3), dim3(256), 0, stream, qp.scale[0], qp.BH * qp.nblk[0],
                           qp.scale[1], qp.BH * qp.nblk[1], qp.scale[2], qp.BH * qp.nblk[2], qp.qmax);
        if (in16) hipLaunchKernelGGL((quantize_kernel<1, true>), dim3(grid), dim3(256), 0, stream, qp);
        else hipLaunchKernelGGL((quantize_kernel<1, false>), dim3(grid), dim3(256), 0, stream, qp);
    }
    if (v_prepass) hipLaunchKernelGGL(vimage_finish_kernel, dim3((qp.BH + 255) / 256), dim3(256), 0, stream, vhdr, qp.BH);
    if (views) {
        views->q8 = qp.q8; views->k8 = qp.k8; views->v16 = qp.v16;
        views->v8 = qp.v8; views->v_e8 = qp.v_e8;
        views->q_scale = qp.scale[0]; views->k_scale = qp.scale[1]; views->v_scale = qp.scale[2];
        views->qf = qp.f32[0]; views->kf = qp.f32[1]; views->vf = qp.f32[2];
        views->qh = qp.f16[0]; views->kh = qp.f16[1]; views->vh = qp.f16[2];
        views->nqblk = qp.nblk[0]; views->nkblk = qp.nblk[1]; views->dpq = qp.DPQ;
    }
    return hipGetLastError();
}

template <int DP, bool CAUSAL, bool HAS_MASK>
static hipError_t launch_i8_one(const I8FwdParams& p, hipStream_t stream) {
    // 64-key tiles everywhere.  (Until the end of round 5 the unmasked non-causal head_dim-128 launch ran 32-key tiles, three workgroups per CU, like the bf16
    // kernel: measured again after the masked instantiation turned out FASTER on an all-true mask than the unmasked one -- 64-key tiles win 3-12 % on the
    // shapes this kernel still serves unmasked, B2 H16 S1000 0.0519 -> 0.0456 ms, config 4 0.587 -> 0.539, profiles/r5/ab_i8_bn64.jsonl)
    constexpr int BN = 64;
    const uint32_t nqb = (p.Sq + 127) / 128;
    const size_t lds = 2 * BN * DP + 2 * BN * DP * 2;
    auto kfn = fa_fwd_i8_kernel<DP, CAUSAL, HAS_MASK, BN>;
    if (hipError_t e = ensure_dynamic_lds((const void*)kfn, lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kfn, dim3(nqb * p.B * p.H), dim3(256), lds, stream, p);
    return hipGetLastError();
}

template <int DP>
static hipError_t launch_i8_flags(const I8FwdParams& p, bool causal, hipStream_t stream) {
    const bool mk = p.mask != nullptr;
    if (causal) return mk ? launch_i8_one<DP, true, true>(p, stream) : launch_i8_one<DP, true, false>(p, stream);
    return mk ? launch_i8_one<DP, false, true>(p, stream) : launch_i8_one<DP, false, false>(p, stream);
}

hipError_t launch_quantized_fwd(const FwdParams& fp, int bits, int quant_mode, void* workspace, hipStream_t stream,
                                const char** name) {
    if (!quantized_supported(fp.D)) return hipErrorInvalidValue;
    if (fp.D > 256) {
        // head dims 257 ... 1024: quantise -> q * s as fp32 -> the wide fp32 forward (both products as fp32 fma chains on exactly the values the
        // int8 kernels multiply: the oracle's quantised restatement to fp32 rounding; P is not rounded to 16 bits here)
        QuantViews v;
        if (hipError_t e = launch_quantize_wide(fp.q, fp.k, fp.v, fp.in_prec, fp.B, fp.H, fp.Sq, fp.Skv, fp.D, bits, quant_mode, workspace, &v, stream); e != hipSuccess) return e;
        FwdParams pw = fp;
        pw.q = v.qf; pw.k = v.kf; pw.v = v.vf;
        pw.in_prec = P_FP32; pw.out_prec = P_FP32;
        const int64_t D = fp.D;
        pw.qs[0] = (int64_t)fp.H * fp.Sq * D; pw.qs[1] = (int64_t)fp.Sq * D; pw.qs[2] = D; pw.qs[3] = 1;
        pw.ks[0] = pw.vs[0] = (int64_t)fp.H * fp.Skv * D; pw.ks[1] = pw.vs[1] = (int64_t)fp.Skv * D; pw.ks[2] = pw.vs[2] = D; pw.ks[3] = pw.vs[3] = 1;
        pw.os[0] = D; pw.os[1] = 1;
        pw.part_buf = nullptr; pw.part_cnt = nullptr; pw.vsc = nullptr; pw.pv16 = 0;
        if (fp.mask && fp.mask_kind == MK_NONE) {  // the ABI's dense fp32 additive [B, H, Sq, Skv]
            pw.mask_kind = MK_F32;
            pw.ms[0] = (int64_t)fp.H * fp.Sq * fp.Skv; pw.ms[1] = (int64_t)fp.Sq * fp.Skv; pw.ms[2] = fp.Skv; pw.ms[3] = 1;
        }
        const char* nm = "none";
        const hipError_t e = launch_fwd_wide(pw, stream, &nm);
        if (name) *name = bits == 4 ? "fa_fwd_wide<i4 images>" : "fa_fwd_wide<i8 images>";
        return e;
    }
    // quant_mode 3 (fp8 P V, opt-in): only the 64-rows-per-wave kernel implements it; every other case runs the
    // block-wise int8 path (mode 2), which is the more accurate of the two
    if (quant_mode == 3 && !(bits == 8 && fp.part_buf && fp.part_cnt && fwd_w64_i8_supported(fp) && !fp.mask)) quant_mode = 2;  // (no mask instantiation of the fp8 kernel)
    QuantViews v;
    // fp.vsc: the (device, stream) pool's slab headers (runtime: StreamScratch::ensure_v16) -- the fp16 V image goes in as q * s * 2^-e
    hipError_t e = launch_quantize(fp.q, fp.k, fp.v, fp.in_prec, fp.B, fp.H, fp.Sq, fp.Skv, fp.D, bits, quant_mode,
                                   workspace, 0, &v, stream, nullptr, (uint32_t*)fp.vsc);
    if (e != hipSuccess) return e;
    I8FwdParams p;
    memset(&p, 0, sizeof(p));
    p.q8 = v.q8; p.k8 = v.k8; p.v16 = (const _Float16*)v.v16;
    p.q_scale = v.q_scale; p.k_scale = v.k_scale;
    p.vsc = v.v8 ? nullptr : fp.vsc;
    p.o = (float*)fp.o; p.lse = fp.lse;
    p.mask = fp.mask;
    const bool w64 = fp.part_buf && fp.part_cnt && fwd_w64_i8_supported(fp);  // (with a bool mask tensor: its MASKT instantiation, the runtime has packed the mask)
    if (fp.mask && !w64) {
        // fp.mask_kind == MK_NONE with a mask: the reference ABI's dense fp32 additive [B, H, Sq, Skv]; else what the caller normalised (runtime.hip normalise_mask)
        FwdParams mp = fp;
        if (fp.mask_kind == MK_NONE) {
            mp.mask_kind = MK_F32;
            mp.ms[0] = (int64_t)fp.H * fp.Sq * fp.Skv; mp.ms[1] = (int64_t)fp.Sq * fp.Skv; mp.ms[2] = fp.Skv; mp.ms[3] = 1;
        }
        if (mp.mask_kind == MK_WINDOW) return hipErrorInvalidValue;
        p.mask_kind = mp.mask_kind;
        for (int i = 0; i < 4; ++i) p.ms[i] = mp.ms[i];
        if (!tuning().no_mask_flags.load(std::memory_order_relaxed)) {
            // the mask's tile flags: one streaming pass over every DISTINCT mask byte, into the workspace
            uint8_t* fl = (uint8_t*)workspace + ws_layout(fp.B, fp.H, fp.Sq, fp.Skv, fp.D, false).mflags;
            if (launch_mask_flags(mp, fl, stream) == hipSuccess && mp.mask_flags) {
                p.mask_flags = mp.mask_flags; p.mf_nrb = mp.mf_nrb; p.mf_ntiles = mp.mf_ntiles;
                p.mf_bs = mp.mf_bs; p.mf_hs = mp.mf_hs;
            }
        }
    }
    p.B = fp.B; p.H = fp.H; p.Sq = fp.Sq; p.Skv = fp.Skv; p.D = fp.D;
    p.nqblk = v.nqblk; p.nkblk = v.nkblk;
    p.scale = fp.scale;
    const uint32_t dp = dp_of(fp.D);
    if (w64) {
        *name = v.v8 ? "fa_fwd_w64_i8f8<128>" : fp.mask ? (bits == 4 ? "fa_fwd_w64_i4<128,mask>" : "fa_fwd_w64_i8<128,mask>") : bits == 4 ? "fa_fwd_w64_i4<128>" : "fa_fwd_w64_i8<128>";
        return launch_fwd_w64_i8(fp, v, fp.part_buf, fp.part_cnt, stream);
    }
    if (dp == 64) { *name = bits == 4 ? "fa_fwd_i4<64>" : "fa_fwd_i8<64>"; return launch_i8_flags<64>(p, fp.causal, stream); }
    if (dp == 128) { *name = bits == 4 ? "fa_fwd_i4<128>" : "fa_fwd_i8<128>"; return launch_i8_flags<128>(p, fp.causal, stream); }
    *name = bits == 4 ? "fa_fwd_i4<256>" : "fa_fwd_i8<256>";
    return launch_i8_flags<256>(p, fp.causal, stream);
}

}  // namespace umfa
