// fa_bwd.hip -- SDPA backward for gfx950 (dense, contiguous BHSD).  mfa_attention_backward takes no mask
// (maskBuffer: nil, MFABridge.swift:3265); mfa_quantized_backward may pass a dense fp32 additive one.
//
// Replaces the two Metal dispatches behind MultiHeadAttention.backward (MFABridge.swift:3253-3266:
// "backward query" then "backward key-value") plus its host-side zeroing of the D scratch:
//   bwd_delta   D[i]  = sum_d dO[i,d] O[i,d]
//   bwd_dq      dQ    = scale * sum_j dS[i,j] K[j]       (workgroup = 128 query rows, sweeps key tiles)
//   bwd_dkdv    dK    = scale * sum_i dS[i,j] Q[i],  dV = sum_i P[i,j] dO[i]
//                                                        (workgroup = 128 keys, sweeps query tiles)
// with P = exp(scale S - LSE) recomputed from the forward's log-sum-exp and dS = P o (dP - D).
// No atomics: dQ and dK/dV each have a single owner workgroup (bitwise reproducible).
//
// This round's backward is the fp32-exact one: operands of any input type are converted on load and
// all five products run on v_mfma_f32_32x32x2_f32 (an fp32 fma chain).  Operand orientations follow
// cdna_hip_programming.md "Attention backward": the accumulator of the first product of each chain is
// directly the B operand of the next (reduction index on the MFMA k-slot = lane half).
#include "fa_common.h"
#include "kernels.h"

namespace umfa {

__global__ __launch_bounds__(256) void bwd_delta_kernel(BwdParams p) {
    // one wave per row: D = rowsum(dO o O)
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int64_t rows = (int64_t)p.B * p.H * p.Sq;
    if (row >= rows) return;
    float s = 0.0f;
    for (uint32_t d = lane; d < p.D; d += 64)
        s += load_as_float(p.dout, row * p.D + d, p.dout_prec) * (p.o_in_type ? load_as_float(p.o, row * p.D + d, p.in_prec) : p.o[row * p.D + d]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) p.dvec[row] = s;
}

template <int DP>
__device__ __forceinline__ void load_tile_f32(float* dst, const void* src, int64_t base, uint32_t row0, uint32_t nrows,
                                              int D, int prec, int tid) {
    constexpr int LD = DP + 1;
    for (int idx = tid; idx < 32 * DP; idx += 256) {
        const int r = idx / DP, d = idx % DP;
        const uint32_t row = row0 + r;
        dst[r * LD + d] = (row < nrows && d < D) ? load_as_float(src, base + (int64_t)row * D + d, prec) : 0.0f;
    }
}

// ---------------------------------------------------------------- dQ
template <int DP>
__global__ __launch_bounds__(256) void bwd_dq_kernel(BwdParams p) {
    constexpr int LD = DP + 1, NDB = DP / 32;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float* Ks = smem_f;
    float* Vs = smem_f + 32 * LD;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, ql = lane & 31, hi = lane >> 5;
    const uint32_t nqb = (p.Sq + 127) / 128;
    const uint32_t bh = blockIdx.x / nqb, qb = blockIdx.x % nqb;
    const uint32_t q_row = qb * 128 + wave * 32 + ql;
    const uint32_t wave_qmax = qb * 128 + wave * 32 + 31;
    const int D = (int)p.D;
    const int64_t qbase = (int64_t)bh * p.Sq * D, kbase = (int64_t)bh * p.Skv * D;
    const bool qok = q_row < p.Sq;

    float qreg[DP / 2], doreg[DP / 2];
#pragma unroll
    for (int ks = 0; ks < DP / 2; ++ks) {
        const int d = 2 * ks + hi;
        const bool ok = qok && d < D;
        qreg[ks] = ok ? load_as_float(p.q, qbase + (int64_t)q_row * D + d, p.in_prec) : 0.0f;
        doreg[ks] = ok ? load_as_float(p.dout, qbase + (int64_t)q_row * D + d, p.dout_prec) : 0.0f;
    }
    const float c = p.scale * UMFA_LOG2E;
    // rows beyond Sq: L2 = +inf -> P = 0
    const float L2 = qok ? p.lse[(int64_t)bh * p.Sq + q_row] * UMFA_LOG2E : INFINITY;
    const float delta = qok ? p.dvec[(int64_t)bh * p.Sq + q_row] : 0.0f;

    f32x16 acc[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    uint32_t ntiles = (p.Skv + 31) / 32;
    if (p.causal) {
        const uint32_t lim = (qb * 128 + 128 + 31) / 32;
        ntiles = ntiles < lim ? ntiles : lim;
    }
    for (uint32_t t = 0; t < ntiles; ++t) {
        __syncthreads();
        load_tile_f32<DP>(Ks, p.k, kbase, t * 32, p.Skv, D, p.in_prec, tid);
        load_tile_f32<DP>(Vs, p.v, kbase, t * 32, p.Skv, D, p.in_prec, tid);
        __syncthreads();
        if (p.causal && t * 32 > wave_qmax) continue;
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.0f; dp[r] = 0.0f; }
#pragma unroll
        for (int ks = 0; ks < DP / 2; ++ks) {
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[ql * LD + 2 * ks + hi], qreg[ks], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[ql * LD + 2 * ks + hi], doreg[ks], dp, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t key = t * 32 + acc_row(r, hi);
            float arg = s[r] * c - L2;
            if (p.mask && qok && key < p.Skv) arg += p.mask[((int64_t)bh * p.Sq + q_row) * p.Skv + key] * UMFA_LOG2E;
            float pr = exp2f(arg);
            if (key >= p.Skv || (p.causal && key > q_row)) pr = 0.0f;
            s[r] = pr * (dp[r] - delta);  // dS^T (without the softmax scale)
        }
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[acc_row(r, hi) * LD + 32 * i + ql], s[r], acc[i], 0, 0, 0);
    }
    if (qok) {
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = 32 * i + acc_row(r, hi);
                if (d < D) p.dq[qbase + (int64_t)q_row * D + d] = acc[i][r] * p.scale;
            }
    }
}

// ---------------------------------------------------------------- dK, dV
template <int DP>
__global__ __launch_bounds__(256) void bwd_dkdv_kernel(BwdParams p) {
    constexpr int LD = DP + 1, NDB = DP / 32;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float* Qs = smem_f;
    float* dOs = smem_f + 32 * LD;
    float* Ls = smem_f + 64 * LD;  // [32] LSE * log2e (+inf beyond Sq)
    float* Ds = Ls + 32;           // [32] delta
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kl = lane & 31, hi = lane >> 5;
    const uint32_t nkb = (p.Skv + 127) / 128;
    const uint32_t bh = blockIdx.x / nkb, kb = blockIdx.x % nkb;
    const uint32_t key = kb * 128 + wave * 32 + kl;
    const uint32_t wave_kmin = kb * 128 + wave * 32;
    const int D = (int)p.D;
    const int64_t qbase = (int64_t)bh * p.Sq * D, kbase = (int64_t)bh * p.Skv * D;
    const bool kok = key < p.Skv;

    // K^T / V^T as B operands: lane (key, hi) holds K[key][2 ks + hi]
    float kreg[DP / 2], vreg[DP / 2];
#pragma unroll
    for (int ks = 0; ks < DP / 2; ++ks) {
        const int d = 2 * ks + hi;
        const bool ok = kok && d < D;
        kreg[ks] = ok ? load_as_float(p.k, kbase + (int64_t)key * D + d, p.in_prec) : 0.0f;
        vreg[ks] = ok ? load_as_float(p.v, kbase + (int64_t)key * D + d, p.in_prec) : 0.0f;
    }
    f32x16 dk[NDB], dv[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[i][r] = 0.0f; dv[i][r] = 0.0f; }
    const float c = p.scale * UMFA_LOG2E;

    const uint32_t ntiles = (p.Sq + 31) / 32;
    const uint32_t t0 = p.causal ? (kb * 128) / 32 : 0;  // query tiles entirely above this key block see nothing
    for (uint32_t t = t0; t < ntiles; ++t) {
        __syncthreads();
        load_tile_f32<DP>(Qs, p.q, qbase, t * 32, p.Sq, D, p.in_prec, tid);
        load_tile_f32<DP>(dOs, p.dout, qbase, t * 32, p.Sq, D, p.dout_prec, tid);
        if (tid < 32) {
            const uint32_t row = t * 32 + tid;
            Ls[tid] = row < p.Sq ? p.lse[(int64_t)bh * p.Sq + row] * UMFA_LOG2E : INFINITY;
            Ds[tid] = row < p.Sq ? p.dvec[(int64_t)bh * p.Sq + row] : 0.0f;
        }
        __syncthreads();
        if (p.causal && t * 32 + 31 < wave_kmin) continue;  // every query of the tile precedes this wave's keys

        // S[q][key], dP[q][key]: rows = queries (registers), columns = keys (lanes)
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.0f; dp[r] = 0.0f; }
#pragma unroll
        for (int ks = 0; ks < DP / 2; ++ks) {
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[kl * LD + 2 * ks + hi], kreg[ks], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(dOs[kl * LD + 2 * ks + hi], vreg[ks], dp, 0, 0, 0);
        }
        f32x16 ds;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qi = acc_row(r, hi);
            const uint32_t qrow = t * 32 + qi;
            float arg = s[r] * c - Ls[qi];
            if (p.mask && kok && qrow < p.Sq) arg += p.mask[((int64_t)bh * p.Sq + qrow) * p.Skv + key] * UMFA_LOG2E;
            float pr = exp2f(arg);
            if (p.causal && key > qrow) pr = 0.0f;
            s[r] = pr;
            ds[r] = pr * (dp[r] - Ds[qi]);
        }
        // dV^T[d][key] += dO^T[d][q] P[q][key];  dK^T[d][key] += Q^T[d][q] dS[q][key]
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int off = acc_row(r, hi) * LD + 32 * i + kl;
                dv[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(dOs[off], s[r], dv[i], 0, 0, 0);
                dk[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[off], ds[r], dk[i], 0, 0, 0);
            }
    }
    if (kok) {
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = 32 * i + acc_row(r, hi);
                if (d < D) {
                    p.dk[kbase + (int64_t)key * D + d] = dk[i][r] * p.scale;
                    p.dv[kbase + (int64_t)key * D + d] = dv[i][r];
                }
            }
    }
}

template <int DP>
static hipError_t launch_bwd_dp(const BwdParams& p, hipStream_t stream) {
    const int64_t rows = (int64_t)p.B * p.H * p.Sq;
    const int ph = p.phases ? p.phases : 7;  // the pre-quantised ABI computes dQ (+D) and dK/dV in separate calls
    if (ph & 1) hipLaunchKernelGGL(bwd_delta_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, p);
    const size_t lds_dq = 2 * 32 * (DP + 1) * sizeof(float);
    const size_t lds_kv = lds_dq + 64 * sizeof(float);
    hipError_t e;
    e = ensure_dynamic_lds((const void*)bwd_dq_kernel<DP>, lds_dq);
    if (e != hipSuccess) return e;
    e = ensure_dynamic_lds((const void*)bwd_dkdv_kernel<DP>, lds_kv);
    if (e != hipSuccess) return e;
    const uint32_t nqb = (p.Sq + 127) / 128, nkb = (p.Skv + 127) / 128;
    if (ph & 2) hipLaunchKernelGGL(bwd_dq_kernel<DP>, dim3(nqb * p.B * p.H), dim3(256), lds_dq, stream, p);
    if (ph & 4) hipLaunchKernelGGL(bwd_dkdv_kernel<DP>, dim3(nkb * p.B * p.H), dim3(256), lds_kv, stream, p);
    return hipGetLastError();
}

hipError_t launch_bwd(const BwdParams& p, hipStream_t stream, const char** name) {
    if (p.D <= 32) { *name = "fa_bwd_exact<32>"; return launch_bwd_dp<32>(p, stream); }
    if (p.D <= 64) { *name = "fa_bwd_exact<64>"; return launch_bwd_dp<64>(p, stream); }
    if (p.D <= 128) { *name = "fa_bwd_exact<128>"; return launch_bwd_dp<128>(p, stream); }
    if (p.D <= 256) { *name = "fa_bwd_exact<256>"; return launch_bwd_dp<256>(p, stream); }
    return launch_bwd_wide(p, stream, name);  // head dims 257 ... 1024 (fa_bwd_wide.hip); beyond: invalid
}

}  // namespace umfa
