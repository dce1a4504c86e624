// fa_fwd_16.hip -- bf16 / fp16 flash-attention forward for gfx950 (the headline path).
//
// Replaces the Metal `attention` forward kernel the reference generates in its absent submodule
// and launches from MultiHeadAttention.forward / encodeForward (MFABridge.swift:2240-2248,
// 2525-2541): O = softmax(scale QK^T [+causal] [+mask]) V with online softmax, never
// materialising S, and WITHOUT the reference's dense fp32 [B,H,Sq,Skv] mask pre-pass
// (MFABridge.swift:368-590) -- the strided mask is read in-tile.
//
// Structure (cdna_hip_programming.md "Fused attention prefill"):
//   workgroup = 4 waves = 128 query rows of one (batch, head); 2 workgroups per CU (<= 256 VGPR);
//   key/value tiles of 64 rows, double-buffered in LDS, staged through registers with the
//   issue-early / write-late split (T14); one barrier per tile;
//   S^T = K Q^T on v_mfma_f32_32x32x16 (swapped operands, T12): a lane owns one query row, so
//   row max / row sum are in-lane plus one exchange with lane^32;
//   P^T (the S^T accumulator, rounded to the input type) is the B operand of O^T += V^T P^T with
//   no cross-lane movement; V^T fragments come from ds_read_b64_tr_b16 (T10);
//   K rows are XOR-swizzled for conflict-free ds_read_b128 (T2), V rows for the transposed reads.
//   Softmax is computed in the log2 domain: p = exp2(s * scale*log2e - m).
#include "fa_common.h"
#include "kernels.h"

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

template <typename T, int DP, bool CAUSAL, bool HAS_MASK, typename OUT>
__global__ __launch_bounds__(256, (DP <= 128 ? 2 : 1)) void fa_fwd16_kernel(FwdParams p) {
    typedef Mma16<T> M;
    typedef typename M::V8 V8;
    typedef typename M::V4 V4;
    constexpr int BM = 128, BN = 64;
    constexpr int NCH = DP / 8;             // 16-byte chunks per row
    constexpr int NKS = DP / 16;            // k-steps of QK^T
    constexpr int NDB = DP / 32;            // 32-row blocks of O^T
    constexpr int TILE_BYTES = BN * DP * 2;
    constexpr int LPT = BN * NCH / 256;     // 16-byte loads per thread per tile
    static_assert(LPT >= 1, "tile too small for 256 threads");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [K buf0][K buf1][V buf0][V buf1]
    char* const Kbuf = smem;
    char* const Vbuf = smem + 2 * TILE_BYTES;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, ql = lane & 31, hi = lane >> 5;
    const uint32_t nqb = (p.Sq + BM - 1) / BM;
    const uint32_t vid = xcd_remap(blockIdx.x, nqb * p.B * p.H);
    const uint32_t bh = vid / nqb;
    uint32_t qb = vid % nqb;
    if (CAUSAL) qb = nqb - 1 - qb;
    const uint32_t b = bh / p.H, h = bh % p.H;
    const uint32_t q_row = qb * BM + wave * 32 + ql;
    const uint32_t wave_q0 = qb * BM + wave * 32;
    const int D = (int)p.D;

    const T* __restrict__ qp = (const T*)p.q + ((int64_t)b * p.qs[0] + (int64_t)h * p.qs[1]);
    const T* __restrict__ kp = (const T*)p.k + ((int64_t)b * p.ks[0] + (int64_t)h * p.ks[1]);
    const T* __restrict__ vp = (const T*)p.v + ((int64_t)b * p.vs[0] + (int64_t)h * p.vs[1]);

    // ---- Q^T fragments (B operand of S^T = K Q^T): lane (q, hi) holds Q[q][16 ks + 8 hi .. +7]
    V8 qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int d0 = 16 * ks + 8 * hi;
        if (q_row < p.Sq && d0 < D)
            qf[ks] = *(const V8*)(qp + (int64_t)q_row * p.qs[2] + d0);
        else
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[ks][j] = (T)0.0f;
    }

    // ---- tile staging: thread owns chunks c = tid + 256 i  ->  (row, ch)
    i32x4 kreg[LPT], vreg[LPT];
    auto stage_load = [&](uint32_t t) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int c = tid + 256 * i, row = c / NCH, ch = c % NCH;
            const uint32_t key = t * BN + row;
            if (key < p.Skv && ch * 8 < D) {
                kreg[i] = *(const i32x4*)(kp + (int64_t)key * p.ks[2] + ch * 8);
                vreg[i] = *(const i32x4*)(vp + (int64_t)key * p.vs[2] + ch * 8);
            } else {
                kreg[i] = i32x4{0, 0, 0, 0};
                vreg[i] = i32x4{0, 0, 0, 0};
            }
        }
    };
    auto stage_write = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int c = tid + 256 * i, row = c / NCH, ch = c % NCH;
            *(i32x4*)(Kbuf + buf * TILE_BYTES + k_off<DP>(row, ch)) = kreg[i];
            *(i32x4*)(Vbuf + buf * TILE_BYTES + v_off<DP>(row, ch)) = vreg[i];
        }
    };

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
    const float c2 = p.scale * UMFA_LOG2E;

    // per-lane LDS read addresses
    // K row read: row = 32 kb + ql, chunk = 2 ks + hi  (computed per use: XOR depends on ks)
    // V transposed read: lane = 16 g + 4 qq + pp supplies row key0 + qq, columns 32 db + 16 (g&1) + 4 pp
    const int tr_qq = (lane >> 2) & 3, tr_pp = lane & 3, tr_g1 = (lane >> 4) & 1;

    const int64_t mrow = HAS_MASK ? ((int64_t)b * p.ms[0] + (int64_t)h * p.ms[1] + (int64_t)q_row * p.ms[2]) : 0;

    stage_load(0);
    stage_write(0);
    __syncthreads();

    for (uint32_t t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        const bool more = t + 1 < ntiles;
        if (more) stage_load(t + 1);  // in flight under this tile's MFMAs (T14)

        const char* Kt = Kbuf + cur * TILE_BYTES;
        const char* Vt = Vbuf + cur * TILE_BYTES;
        const uint32_t key_base = t * BN;
        // wave-uniform: is any part of this tile visible to this wave's rows?
        const bool active = !CAUSAL || key_base <= wave_q0 + 31;

        if (active) {
            // ---------------- S^T = K Q^T ----------------
            f32x16 s[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] = 0.0f;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const V8 a = *(const V8*)(Kt + k_off<DP>(32 * kb + ql, 2 * ks + hi));
                    s[kb] = M::mma(a, qf[ks], s[kb]);
                }
            }

            // ---------------- online softmax (log2 domain) ----------------
            const bool edge = (key_base + BN > p.Skv) || (CAUSAL && key_base + BN - 1 > wave_q0);
            float mx = -INFINITY;
            if (HAS_MASK) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const uint32_t key = key_base + 32 * kb + acc_row(r, hi);
                        float tv = s[kb][r] * c2;
                        if (key < p.Skv && q_row < p.Sq)
                            tv += mask_term(p.mask, mrow + (int64_t)key * p.ms[3], p.mask_kind);
                        if (key >= p.Skv || (CAUSAL && key > q_row)) tv = -INFINITY;
                        s[kb][r] = tv;
                        mx = fmaxf(mx, tv);
                    }
            } else {
                if (edge) {
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const uint32_t key = key_base + 32 * kb + acc_row(r, hi);
                            if (key >= p.Skv || (CAUSAL && key > q_row)) s[kb][r] = -INFINITY;
                        }
                }
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
                mx *= c2;  // scale > 0 on this path
            }
            mx = fmaxf(mx, xor32(mx));
            const float m_new = fmaxf(m, mx);
            const float m_use = (HAS_MASK && m_new == -INFINITY) ? 0.0f : m_new;
            if (!__all(m_new == m)) {
                const float alpha = __builtin_amdgcn_exp2f(m - m_use);
                l *= alpha;
#pragma unroll
                for (int i = 0; i < NDB; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] *= alpha;
                m = m_new;
            }
            float rs = 0.0f;
            V8 pf[4];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = HAS_MASK ? __builtin_amdgcn_exp2f(s[kb][r] - m_use)
                                             : __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][r], c2, -m_use));
                    rs += e;
                    pf[2 * kb + (r >> 3)][r & 7] = (T)e;
                }
            l += rs;

            // ---------------- O^T += V^T P^T ----------------
#pragma unroll
            for (int i = 0; i < NDB; ++i) {
#pragma unroll
                for (int st = 0; st < 4; ++st) {  // 16-key step: keys 16 st + {4 hi + 0..3, 8 + 4 hi + 0..3}
                    const int row0 = 16 * st + 4 * hi + tr_qq;
                    const int ch = 4 * i + 2 * tr_g1 + (tr_pp >> 1);
                    const V4 lo = M::tr_read(Vt + v_off<DP>(row0, ch) + 8 * (tr_pp & 1));
                    const V4 hi4 = M::tr_read(Vt + v_off<DP>(row0 + 8, ch) + 8 * (tr_pp & 1));
                    const V8 a = __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
                    acc[i] = M::mma(a, pf[st], acc[i]);
                }
            }
        }

        if (more) stage_write(cur ^ 1);
        __syncthreads();
    }

    // ---------------- epilogue ----------------
    const float lt = l + xor32(l);
    const float inv = lt > 0.0f ? 1.0f / lt : 0.0f;
    if (q_row < p.Sq) {
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
}

bool fwd_16_supported(const FwdParams& p) {
    if (p.in_prec != P_FP16 && p.in_prec != P_BF16) return false;
    if (p.D == 0 || p.D % 8 != 0 || p.D > 256) return false;
    if (!(p.scale > 0.0f)) return false;
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    if (!al16(p.q) || !al16(p.k) || !al16(p.v) || !al16(p.o)) return false;
    for (int i = 0; i < 3; ++i)
        if (p.qs[i] % 8 || p.ks[i] % 8 || p.vs[i] % 8) return false;
    if (p.qs[3] != 1 || p.ks[3] != 1 || p.vs[3] != 1) return false;
    if (p.os[0] != (int64_t)p.D || p.os[1] != 1) return false;
    return true;
}

template <typename T, int DP, bool CAUSAL, bool HAS_MASK, typename OUT>
static hipError_t launch_one(const FwdParams& p, hipStream_t stream) {
    const uint32_t nqb = (p.Sq + 127) / 128;
    const size_t lds = 4 * 64 * DP * 2;
    auto kfn = fa_fwd16_kernel<T, DP, CAUSAL, HAS_MASK, OUT>;
    static bool attr_set = false;  // per instantiation
    if (lds > 48 * 1024 && !attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kfn, dim3(nqb * p.B * p.H), dim3(256), lds, stream, p);
    return hipGetLastError();
}

template <typename T, int DP, typename OUT>
static hipError_t launch_flags(const FwdParams& p, hipStream_t stream) {
    const bool mk = p.mask_kind != MK_NONE;
    if (p.causal) return mk ? launch_one<T, DP, true, true, OUT>(p, stream) : launch_one<T, DP, true, false, OUT>(p, stream);
    return mk ? launch_one<T, DP, false, true, OUT>(p, stream) : launch_one<T, DP, false, false, OUT>(p, stream);
}

template <typename T, int DP>
static hipError_t launch_out(const FwdParams& p, hipStream_t stream) {
    if (p.out_prec == P_FP32) return launch_flags<T, DP, float>(p, stream);
    // 16-bit epilogue only in the input type (the torch binding's cast-back, metal_sdpa_backend.cpp:1442-1444)
    if (p.out_prec == p.in_prec) return launch_flags<T, DP, T>(p, stream);
    return hipErrorNotSupported;
}

template <typename T>
static hipError_t launch_dp(const FwdParams& p, hipStream_t stream, const char** name) {
    constexpr bool bf = sizeof(T) == 2 && __is_same(T, __bf16);
    if (p.D <= 32) { *name = bf ? "fa_fwd16<bf16,32>" : "fa_fwd16<fp16,32>"; return launch_out<T, 32>(p, stream); }
    if (p.D <= 64) { *name = bf ? "fa_fwd16<bf16,64>" : "fa_fwd16<fp16,64>"; return launch_out<T, 64>(p, stream); }
    if (p.D <= 128) { *name = bf ? "fa_fwd16<bf16,128>" : "fa_fwd16<fp16,128>"; return launch_out<T, 128>(p, stream); }
    *name = bf ? "fa_fwd16<bf16,256>" : "fa_fwd16<fp16,256>";
    return launch_out<T, 256>(p, stream);
}

hipError_t launch_fwd_16(const FwdParams& p, hipStream_t stream, const char** name) {
    if (!fwd_16_supported(p)) return hipErrorNotSupported;
    return p.in_prec == P_BF16 ? launch_dp<__bf16>(p, stream, name) : launch_dp<_Float16>(p, stream, name);
}

}  // namespace umfa
