/*
 * oracle/sdpa_ref.c -- CPU restatement of the reference's SDPA hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under universal-metal-flash-attention_amd/
 * may link, load or call this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg do, and only as the checker.
 *
 * What it restates (reference = bghira/universal-metal-flash-attention):
 *   - O = softmax(scale * Q K^T [+causal] [+mask]) V, fp32 O            mfa_ffi.h:245-300
 *   - contiguous / strided BHSD operands                                 metal_sdpa_backend.cpp:188-193,1042-1058
 *   - mask broadcast rules (right-aligned, size-1 dims broadcast,
 *     element strides, bool nonzero = attend, additive fp32/fp16/bf16)   MFABridge.swift:157-242
 *   - causal = lower triangular, top-left aligned (torch is_causal)      MFABridge.swift:2205
 *   - LSE [B*H*Sq] fp32                                                  metal_sdpa_backend.cpp:2711-2712
 *   - backward: D = rowsum(dO o O); dQ, dK, dV fp32                      MFABridge.swift:3171-3282
 *   - symmetric quantiser: scale = absmax/127 (/7), q = clamp(round(x/scale)),
 *     zero point 0, int4 nibble packing (even index low nibble, +8 bias) Tests/QuantizationTests/QuantizationTests.swift:72-128
 *   - quantised forward = dequantise-on-load, then the same fp math      AGENTS.md:143-152
 *
 * The arithmetic of the reference lives in an un-vendored submodule
 * (bghira/metal-flash-attention-plus, commit not recorded); this oracle is
 * pinned instead against torch-CPU scaled_dot_product_attention, which the
 * reference's own tests declare as ground truth (tests/conftest.py:165-182,
 * test_scale_factor_fix.py:55-66), through tests/golden/ fixtures.
 * Tile-level / bit-level behaviour of the Metal kernels: parity unpinned.
 *
 * All accumulation is in double; outputs are rounded once to fp32.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { PREC_FP16 = 0, PREC_BF16 = 1, PREC_FP32 = 2 };
enum { MASK_NONE = 0, MASK_BOOL = 1, MASK_ADDITIVE = 2 };
enum { MSCALAR_BYTE = 0, MSCALAR_FP16 = 1, MSCALAR_BF16 = 2, MSCALAR_FP32 = 3 };

static float half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal: renormalise */
            int e = -1;
            do { man <<= 1; ++e; } while (!(man & 0x400u));
            man &= 0x3ffu;
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | (man << 13);
    } else {
        bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

static float bf16_to_float(uint16_t b) {
    uint32_t bits = (uint32_t)b << 16; /* MFABridge.swift:221-225 */
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

static double load_elem(const void* base, int prec, int64_t idx) {
    switch (prec) {
    case PREC_FP16: return half_to_float(((const uint16_t*)base)[idx]);
    case PREC_BF16: return bf16_to_float(((const uint16_t*)base)[idx]);
    default: return ((const float*)base)[idx];
    }
}

/* Additive mask value for (b,h,i,j); follows mfa_prepare_mask, MFABridge.swift:157-242. */
static double mask_value(const void* mask, int mask_type, int mask_scalar, const int64_t* shape,
                         const int64_t* strides, uint32_t ndim, uint32_t b, uint32_t h, uint32_t i,
                         uint32_t j) {
    if (mask_type == MASK_NONE || !mask || !shape || !strides) return 0.0;
    if (ndim > 4) return 0.0; /* sync path: no mask, MFABridge.swift:236-238 */
    uint32_t coords[4] = {b, h, i, j};
    int64_t lin = 0;
    for (uint32_t d = 0; d < ndim; ++d) {
        uint32_t c = coords[4 - ndim + d];
        if (shape[d] == 1) c = 0;
        lin += (int64_t)c * strides[d];
    }
    if (mask_type == MASK_BOOL) return ((const uint8_t*)mask)[lin] != 0 ? 0.0 : -INFINITY;
    if (mask_type == MASK_ADDITIVE) {
        switch (mask_scalar) {
        case MSCALAR_FP32: return ((const float*)mask)[lin];
        case MSCALAR_FP16: return half_to_float(((const uint16_t*)mask)[lin]);
        case MSCALAR_BF16: return bf16_to_float(((const uint16_t*)mask)[lin]);
        default: return 0.0;
        }
    }
    return 0.0;
}

/*
 * Forward.  q/k/v element strides are BHSD order (4 values each, last = 1 for
 * dense); NULL = contiguous [B,H,S,D].  out is dense fp32 [B,H,Sq,D]; lse (may
 * be NULL) is fp32 [B*H*Sq], natural log: lse = m + ln(sum exp(s - m)).
 * A row with every key masked yields O = 0 and lse = -inf (flash convention;
 * torch's math path would give NaN -- the reference never tests it).
 */
int ref_sdpa_forward(const void* q, const void* k, const void* v, float* out, float* lse,
                     uint32_t B, uint32_t H, uint32_t Sq, uint32_t Skv, uint32_t D, float scale,
                     int causal, int prec, const int64_t* qs, const int64_t* ks,
                     const int64_t* vs, const void* mask, const int64_t* mshape,
                     const int64_t* mstrides, uint32_t mndim, int mask_type, int mask_scalar) {
    int64_t dq_[4] = {(int64_t)H * Sq * D, (int64_t)Sq * D, D, 1};
    int64_t dk_[4] = {(int64_t)H * Skv * D, (int64_t)Skv * D, D, 1};
    if (!qs) qs = dq_;
    if (!ks) ks = dk_;
    if (!vs) vs = dk_;
    int failed = 0;
    /* (batch, head) slabs are independent, and so are the rows of a slab: one OpenMP task per (slab, chunk of
     * ROW_CHUNK rows), so that few-head cases (row-subset checks at the full sizes) still use every host core.
     * A task converts the slab's K and V to double itself (Skv*D loads against ROW_CHUNK*Skv*D multiply-adds). */
    enum { ROW_CHUNK = 32 };
    const int64_t nchunk = ((int64_t)Sq + ROW_CHUNK - 1) / ROW_CHUNK;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t task = 0; task < (int64_t)B * H * nchunk; ++task) {
        const int64_t bh = task / nchunk;
        const uint32_t i0 = (uint32_t)(task % nchunk) * ROW_CHUNK;
        const uint32_t i1 = i0 + ROW_CHUNK < Sq ? i0 + ROW_CHUNK : Sq;
        uint32_t b = (uint32_t)(bh / H), h = (uint32_t)(bh % H);
        size_t nkv = (size_t)Skv * D;
        double* kd = (double*)malloc(sizeof(double) * (2 * nkv + Skv + 2 * D + 1));
        if (!kd) { failed = 1; continue; }
        double *vd = kd + nkv, *s = vd + nkv, *acc = s + Skv, *qd = acc + D;
        for (uint32_t j = 0; j < Skv; ++j)
            for (uint32_t d = 0; d < D; ++d) {
                kd[(size_t)j * D + d] = load_elem(k, prec, b * ks[0] + h * ks[1] + j * ks[2] + d * ks[3]);
                vd[(size_t)j * D + d] = load_elem(v, prec, b * vs[0] + h * vs[1] + j * vs[2] + d * vs[3]);
            }
        for (uint32_t i = i0; i < i1; ++i) {
            for (uint32_t d = 0; d < D; ++d)
                qd[d] = load_elem(q, prec, b * qs[0] + h * qs[1] + i * qs[2] + d * qs[3]);
            double m = -INFINITY;
            for (uint32_t j = 0; j < Skv; ++j) {
                const double* kr = kd + (size_t)j * D;
                double dot = 0.0;
                for (uint32_t d = 0; d < D; ++d) dot += qd[d] * kr[d];
                double sv = dot * (double)scale;
                if (causal && j > i) sv = -INFINITY;
                if (mask_type != MASK_NONE)
                    sv += mask_value(mask, mask_type, mask_scalar, mshape, mstrides, mndim, b, h, i, j);
                s[j] = sv;
                if (sv > m) m = sv;
            }
            double l = 0.0;
            for (uint32_t d = 0; d < D; ++d) acc[d] = 0.0;
            if (m > -INFINITY) {
                for (uint32_t j = 0; j < Skv; ++j) {
                    double p = exp(s[j] - m);
                    l += p;
                    if (p != 0.0) {
                        const double* vr = vd + (size_t)j * D;
                        for (uint32_t d = 0; d < D; ++d) acc[d] += p * vr[d];
                    }
                }
            }
            size_t row = ((size_t)b * H + h) * Sq + i;
            for (uint32_t d = 0; d < D; ++d) out[row * D + d] = (float)(l > 0.0 ? acc[d] / l : 0.0);
            if (lse) lse[row] = (float)(l > 0.0 ? m + log(l) : -INFINITY);
        }
        free(kd);
    }
    return failed ? 2 : 0;
}

/*
 * Backward (dense, contiguous BHSD; MFABridge.swift:3171-3282 takes no mask).
 * dout/q/k/v in `prec`; out fp32; lse fp32 natural log; dq/dk/dv fp32; dvec
 * (may be NULL) receives D = rowsum(dO o O).
 */
int ref_sdpa_backward(const void* dout, const void* q, const void* k, const void* v,
                      const float* out, const float* lse, float* dq, float* dk, float* dv,
                      float* dvec, uint32_t B, uint32_t H, uint32_t Sq, uint32_t Skv, uint32_t D,
                      float scale, int causal, int prec) {
    size_t nkv = (size_t)Skv * D;
    double* dkacc = (double*)calloc(nkv ? nkv : 1, sizeof(double));
    double* dvacc = (double*)calloc(nkv ? nkv : 1, sizeof(double));
    double* dqacc = (double*)malloc(sizeof(double) * (D ? D : 1));
    if (!dkacc || !dvacc || !dqacc) { free(dkacc); free(dvacc); free(dqacc); return 2; }
    for (uint32_t b = 0; b < B; ++b)
        for (uint32_t h = 0; h < H; ++h) {
            size_t bh = (size_t)b * H + h;
            const size_t qoff = bh * Sq * D, koff = bh * Skv * D;
            memset(dkacc, 0, sizeof(double) * nkv);
            memset(dvacc, 0, sizeof(double) * nkv);
            for (uint32_t i = 0; i < Sq; ++i) {
                double delta = 0.0;
                for (uint32_t d = 0; d < D; ++d)
                    delta += load_elem(dout, prec, qoff + (size_t)i * D + d) * (double)out[qoff + (size_t)i * D + d];
                if (dvec) dvec[bh * Sq + i] = (float)delta;
                for (uint32_t d = 0; d < D; ++d) dqacc[d] = 0.0;
                double L = lse[bh * Sq + i];
                for (uint32_t j = 0; j < Skv; ++j) {
                    if (causal && j > i) break;
                    double dot = 0.0, dp = 0.0;
                    for (uint32_t d = 0; d < D; ++d) {
                        dot += load_elem(q, prec, qoff + (size_t)i * D + d) * load_elem(k, prec, koff + (size_t)j * D + d);
                        dp += load_elem(dout, prec, qoff + (size_t)i * D + d) * load_elem(v, prec, koff + (size_t)j * D + d);
                    }
                    double p = exp(dot * (double)scale - L);
                    double ds = p * (dp - delta) * (double)scale;
                    for (uint32_t d = 0; d < D; ++d) {
                        dvacc[(size_t)j * D + d] += p * load_elem(dout, prec, qoff + (size_t)i * D + d);
                        dkacc[(size_t)j * D + d] += ds * load_elem(q, prec, qoff + (size_t)i * D + d);
                        dqacc[d] += ds * load_elem(k, prec, koff + (size_t)j * D + d);
                    }
                }
                for (uint32_t d = 0; d < D; ++d) dq[qoff + (size_t)i * D + d] = (float)dqacc[d];
            }
            for (size_t e = 0; e < nkv; ++e) {
                dk[koff + e] = (float)dkacc[e];
                dv[koff + e] = (float)dvacc[e];
            }
        }
    free(dkacc);
    free(dvacc);
    free(dqacc);
    return 0;
}

/*
 * Symmetric quantiser, QuantizationTests.swift:72-128.  `bits` = 8 or 4.
 * x: n floats; groups of `group` consecutive elements share one scale
 * (group == n: per-tensor).  qout: int8 values (for bits==4 the values are in
 * [-8,7], one per byte, unpacked); scales: n/group floats.  Rounding is
 * round-half-away-from-zero (Swift `round`), done on the fp32 quotient.
 */
void ref_quantize_symmetric(const float* x, size_t n, size_t group, int bits, int8_t* qout,
                            float* scales) {
    const float qmax = bits == 4 ? 7.0f : 127.0f;
    const int lo = bits == 4 ? -8 : -128, hi = bits == 4 ? 7 : 127;
    for (size_t g0 = 0, gi = 0; g0 < n; g0 += group, ++gi) {
        size_t g1 = g0 + group < n ? g0 + group : n;
        float amax = 0.0f;
        for (size_t e = g0; e < g1; ++e) {
            float a = fabsf(x[e]);
            if (a > amax) amax = a;
        }
        float sc = amax > 0.0f ? amax / qmax : 1.0f;
        scales[gi] = sc;
        for (size_t e = g0; e < g1; ++e) {
            int v = (int)roundf(x[e] / sc);
            if (v < lo) v = lo;
            if (v > hi) v = hi;
            qout[e] = (int8_t)v;
        }
    }
}

/* int4 packing, QuantizationTests.swift:96-104: byte = ((q1+8)<<4) | (q0+8), even index low. */
void ref_pack_int4(const int8_t* q, size_t n, uint8_t* packed) {
    for (size_t i = 0; i < n; i += 2) {
        int a = q[i] + 8, b = (i + 1 < n ? q[i + 1] : 0) + 8;
        a = a < 0 ? 0 : (a > 15 ? 15 : a);
        b = b < 0 ? 0 : (b > 15 ? 15 : b);
        packed[i / 2] = (uint8_t)((b << 4) | a);
    }
}

void ref_unpack_int4(const uint8_t* packed, size_t n, int8_t* q) {
    for (size_t i = 0; i < n; ++i) {
        uint8_t byte = packed[i / 2];
        q[i] = (int8_t)((int)((i & 1) ? (byte >> 4) : (byte & 0xF)) - 8);
    }
}

/* (q - zp) * scale, QuantizationTests.swift:106-110 (zp == 0 on the symmetric path). */
void ref_dequantize(const int8_t* q, size_t n, size_t group, const float* scales, float* x) {
    for (size_t e = 0; e < n; ++e) x[e] = (float)q[e] * scales[e / group];
}

/*
 * Quantised forward, MFABridge+Quantized.swift:227-358: runtime-quantise Q, K
 * and V (per tensor when quant_mode == 0; per (batch, head, block of
 * `block_rows` rows) when quant_mode == 2), dequantise, then the fp forward.
 * Inputs contiguous BHSD in `prec`; additive fp32 mask [B,H,Sq,Skv] or NULL.
 */
int ref_quantized_forward(const void* q, const void* k, const void* v, float* out, float* lse,
                          const float* mask, uint32_t B, uint32_t H, uint32_t Sq, uint32_t Skv,
                          uint32_t D, float scale, int causal, int bits, int quant_mode,
                          uint32_t block_rows, int prec) {
    size_t nq = (size_t)B * H * Sq * D, nk = (size_t)B * H * Skv * D;
    float* fq = (float*)malloc(sizeof(float) * (nq + 2 * nk));
    int8_t* i8 = (int8_t*)malloc(nq > nk ? nq : nk);
    if (!fq || !i8) { free(fq); free(i8); return 2; }
    float *fk = fq + nq, *fv = fk + nk;
    for (size_t e = 0; e < nq; ++e) fq[e] = (float)load_elem(q, prec, (int64_t)e);
    for (size_t e = 0; e < nk; ++e) {
        fk[e] = (float)load_elem(k, prec, (int64_t)e);
        fv[e] = (float)load_elem(v, prec, (int64_t)e);
    }
    float* tensors[3] = {fq, fk, fv};
    size_t counts[3] = {nq, nk, nk};
    uint32_t seqs[3] = {Sq, Skv, Skv};
    for (int t = 0; t < 3; ++t) {
        if (quant_mode == 2) {
            /* blocks never straddle a (batch, head) slab: quantise slab by slab */
            size_t slab = (size_t)seqs[t] * D, group = (size_t)block_rows * D;
            size_t ngroups = (slab + group - 1) / group;
            float* sc = (float*)malloc(sizeof(float) * ngroups);
            for (size_t s0 = 0; s0 < counts[t]; s0 += slab) {
                ref_quantize_symmetric(tensors[t] + s0, slab, group, bits, i8, sc);
                ref_dequantize(i8, slab, group, sc, tensors[t] + s0);
            }
            free(sc);
        } else {
            float sc;
            ref_quantize_symmetric(tensors[t], counts[t], counts[t], bits, i8, &sc);
            ref_dequantize(i8, counts[t], counts[t], &sc, tensors[t]);
        }
    }
    int64_t mshape[4] = {B, H, Sq, Skv};
    int64_t mstr[4] = {(int64_t)H * Sq * Skv, (int64_t)Sq * Skv, Skv, 1};
    int rc = ref_sdpa_forward(fq, fk, fv, out, lse, B, H, Sq, Skv, D, scale, causal, PREC_FP32, NULL,
                              NULL, NULL, mask, mshape, mstr, 4, mask ? MASK_ADDITIVE : MASK_NONE,
                              MSCALAR_FP32);
    free(fq);
    free(i8);
    return rc;
}

/*
 * Rotary rotation, MFABridge.swift:269-319 (rope_rotate_*): interleaved pairs, fp32 tables [S,D] (or [B,S,D],
 * table_batch_stride = S*D) with pair-duplicated entries of which only the even one is read; strided BHSD source
 * (element strides, head_dim contiguous), dense BHSD destination in the same element type; negate_sin = inverse.
 * Math in double, rounded once to the element type's fp32 image (the caller rounds to fp16/bf16).
 */
void ref_rope_rotate(const void* src, float* dst, const float* cos_t, const float* sin_t, uint32_t B, uint32_t H,
                     uint32_t S, uint32_t D, int64_t sb, int64_t sh, int64_t ss, int64_t table_batch_stride,
                     int negate_sin, int prec) {
    for (uint32_t b = 0; b < B; ++b)
        for (uint32_t h = 0; h < H; ++h)
            for (uint32_t s = 0; s < S; ++s)
                for (uint32_t pr = 0; pr < D / 2; ++pr) {
                    int64_t si = b * sb + h * sh + s * ss + 2 * pr;
                    size_t di = (((size_t)b * H + h) * S + s) * D + 2 * pr;
                    size_t t = (size_t)b * table_batch_stride + (size_t)s * D + 2 * pr;
                    double c = cos_t[t], sn = negate_sin ? -(double)sin_t[t] : (double)sin_t[t];
                    double x0 = load_elem(src, prec, si), x1 = load_elem(src, prec, si + 1);
                    dst[di] = (float)(x0 * c - x1 * sn);
                    dst[di + 1] = (float)(x1 * c + x0 * sn);
                }
}

/*
 * Group-wise Walsh-Hadamard transform, normalised by 1/sqrt(N) (contract: MFABridge.swift:3433-3459,
 * AGENTS.md:161-170 "double application = identity").  Sylvester (natural) ordering; in place on fp64.
 */
void ref_hadamard(double* x, size_t block, size_t nblocks) {
    for (size_t bI = 0; bI < nblocks; ++bI) {
        double* v = x + bI * block;
        for (size_t h = 1; h < block; h <<= 1)
            for (size_t i = 0; i < block; i += 2 * h)
                for (size_t j = i; j < i + h; ++j) {
                    double a = v[j], b = v[j + h];
                    v[j] = a + b;
                    v[j + h] = a - b;
                }
        double nrm = 1.0 / sqrt((double)block);
        for (size_t i = 0; i < block; ++i) v[i] *= nrm;
    }
}
