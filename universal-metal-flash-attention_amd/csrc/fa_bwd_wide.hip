// fa_bwd_wide.hip -- SDPA backward for head dims 257 ... 1024, fp32 arithmetic, any operand type, dense contiguous BHSD, causal or not.
//
// The reference's callers admit head_dim <= 1024 (examples/pytorch-custom-op-ffi/src/metal_sdpa_backend.cpp:1078-1086) and its autograd
// functions hand such calls to mfa_attention_backward like every other (:2672-3397, MFABridge.swift:3171-3282).  Nothing in the reference's
// tests or models uses them: this is the domain-completing path behind fa_fwd_wide.hip, correct to the fp32-exact backward's standard
// (fa_bwd.hip: operands converted to fp32 on load, all five products on v_mfma_f32_32x32x2_f32), slow by design.
//
// Same decomposition as the exact backward -- D = rowsum(dO o O); dQ owned by a workgroup of query rows sweeping key tiles; dK / dV owned by
// a workgroup of keys sweeping query tiles; no atomics -- with fa_fwd_wide's answer to "32 rows x 1024 columns do not fit a wave": the four
// waves of a workgroup SHARE THE HEAD DIM.  Wave w owns columns [w DPW, (w + 1) DPW): its slices of the row operands in registers, its slice
// of the gradient as accumulators, its slice of every 32-row tile of the other side in LDS (private to the wave).  S and dP are sums over the
// head dim: each wave forms its partial, the four meet in LDS and are added in wave order (deterministic), every wave then holds the full
// 32 x 32 tile and continues alone.  A tile area holds one slice at a time (32 x 257 fp32 per wave = the LDS there is), so tiles are staged
// again where a product needs them again; the dK / dV kernel makes two sweeps (dV, then dK) so that one accumulator slice is live at a time.
#include <type_traits>

#include "fa_common.h"
#include "kernels.h"

namespace umfa {

namespace {

__global__ __launch_bounds__(256) void bwd_wide_delta_kernel(BwdParams p) {
    // one wave per row: D = rowsum(dO o O)
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= (int64_t)p.B * p.H * p.Sq) return;
    float s = 0.0f;
    for (uint32_t d = lane; d < p.D; d += 64)
        s += load_as_float(p.dout, row * p.D + d, p.dout_prec) * (p.o_in_type ? load_as_float(p.o, row * p.D + d, p.in_prec) : p.o[row * p.D + d]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) p.dvec[row] = s;
}

// rows [row0, row0 + 32) x columns [d0, d0 + DPW) of a dense [nrows][D] slab -> this wave's LDS rows (lanes run along the head dim)
template <int DPW>
__device__ __forceinline__ void stage_slice(float* Ts, const void* src, int64_t base, uint32_t row0, uint32_t nrows, int D, int d0, int prec, int lane) {
    constexpr int LDK = DPW + 1;
    for (int idx = lane; idx < 32 * DPW; idx += 64) {
        const int r = idx / DPW, dd = idx - r * DPW, d = d0 + dd;
        const uint32_t row = row0 + r;
        Ts[r * LDK + dd] = (row < nrows && d < D) ? load_as_float(src, base + (int64_t)row * D + d, prec) : 0.0f;
    }
    // the slice is private to this wave, but its lanes read what other lanes wrote: wave-level ordering, no workgroup barrier
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// the four waves' partial 32 x 32 tiles -> the full tile in every wave (added in wave order).  Two barriers: the partials are in place / they are read.
__device__ __forceinline__ void sum_over_waves(f32x16& s, float* Sx, int wave, int lane) {
#pragma unroll
    for (int r = 0; r < 16; ++r) Sx[(wave * 16 + r) * 64 + lane] = s[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r)
        s[r] = ((Sx[(0 * 16 + r) * 64 + lane] + Sx[(1 * 16 + r) * 64 + lane]) + Sx[(2 * 16 + r) * 64 + lane]) + Sx[(3 * 16 + r) * 64 + lane];
    __syncthreads();
}

}  // namespace

// ---------------------------------------------------------------- dQ: workgroup = 32 query rows
template <int DPW>
__global__ __launch_bounds__(256, 1) void bwd_wide_dq_kernel(BwdParams p) {
    constexpr int LDK = DPW + 1, NDB = DPW / 32;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, ql = lane & 31, hi = lane >> 5;
    float* const Ts = smem_f + wave * (32 * LDK);
    float* const Sx = smem_f + 4 * (32 * LDK);
    const uint32_t nqb = (p.Sq + 31) / 32;
    const uint32_t bh = blockIdx.x / nqb, qb = blockIdx.x % nqb;
    const uint32_t q_row = qb * 32 + ql;
    const int D = (int)p.D, d0 = wave * DPW;
    const int64_t qbase = (int64_t)bh * p.Sq * D, kbase = (int64_t)bh * p.Skv * D;
    const bool qok = q_row < p.Sq;

    // Q^T / dO^T slices as B operands: lane (q, hi) holds Q[q][d0 + 2 ks + hi]
    float qreg[DPW / 2], doreg[DPW / 2];
#pragma unroll
    for (int ks = 0; ks < DPW / 2; ++ks) {
        const int d = d0 + 2 * ks + hi;
        const bool ok = qok && d < D;
        qreg[ks] = ok ? load_as_float(p.q, qbase + (int64_t)q_row * D + d, p.in_prec) : 0.0f;
        doreg[ks] = ok ? load_as_float(p.dout, qbase + (int64_t)q_row * D + d, p.dout_prec) : 0.0f;
    }
    const float c = p.scale * UMFA_LOG2E;
    const float L2 = qok ? p.lse[(int64_t)bh * p.Sq + q_row] * UMFA_LOG2E : INFINITY;  // rows beyond Sq: P = 0
    const float delta = qok ? p.dvec[(int64_t)bh * p.Sq + q_row] : 0.0f;
    f32x16 acc[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    uint32_t ntiles = (p.Skv + 31) / 32;
    if (p.causal) ntiles = ntiles < qb + 1 ? ntiles : qb + 1;
    for (uint32_t t = 0; t < ntiles; ++t) {
        stage_slice<DPW>(Ts, p.k, kbase, t * 32, p.Skv, D, d0, p.in_prec, lane);
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.0f; dp[r] = 0.0f; }
#pragma unroll
        for (int ks = 0; ks < DPW / 2; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x2f32(Ts[ql * LDK + 2 * ks + hi], qreg[ks], s, 0, 0, 0);
        sum_over_waves(s, Sx, wave, lane);
        stage_slice<DPW>(Ts, p.v, kbase, t * 32, p.Skv, D, d0, p.in_prec, lane);
#pragma unroll
        for (int ks = 0; ks < DPW / 2; ++ks) dp = __builtin_amdgcn_mfma_f32_32x32x2f32(Ts[ql * LDK + 2 * ks + hi], doreg[ks], dp, 0, 0, 0);
        sum_over_waves(dp, Sx, wave, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t key = t * 32 + acc_row(r, hi);
            float pr = exp2f(s[r] * c - L2);
            if (key >= p.Skv || (p.causal && key > q_row)) pr = 0.0f;
            s[r] = pr * (dp[r] - delta);  // dS^T (without the softmax scale)
        }
        stage_slice<DPW>(Ts, p.k, kbase, t * 32, p.Skv, D, d0, p.in_prec, lane);  // (the V slice sat where K was)
        // dQ^T (this wave's columns) += K^T dS^T: k index of the MFMA = lane half = key acc_row(r, hi)
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ts[acc_row(r, hi) * LDK + 32 * i + ql], s[r], acc[i], 0, 0, 0);
    }
    if (qok) {
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = d0 + 32 * i + acc_row(r, hi);
                if (d < D) p.dq[qbase + (int64_t)q_row * D + d] = acc[i][r] * p.scale;
            }
    }
}

// ---------------------------------------------------------------- dK, dV: workgroup = 32 keys, two sweeps over the query tiles
template <int DPW>
__global__ __launch_bounds__(256, 1) void bwd_wide_dkdv_kernel(BwdParams p) {
    constexpr int LDK = DPW + 1, NDB = DPW / 32;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kl = lane & 31, hi = lane >> 5;
    float* const Ts = smem_f + wave * (32 * LDK);
    float* const Sx = smem_f + 4 * (32 * LDK);
    const uint32_t nkb = (p.Skv + 31) / 32;
    const uint32_t bh = blockIdx.x / nkb, kb = blockIdx.x % nkb;
    const uint32_t key = kb * 32 + kl;
    const int D = (int)p.D, d0 = wave * DPW;
    const int64_t qbase = (int64_t)bh * p.Sq * D, kbase = (int64_t)bh * p.Skv * D;
    const bool kok = key < p.Skv;

    // K^T slice as the B operand: lane (key, hi) holds K[key][d0 + 2 ks + hi]  (V^T likewise, in the sweep that needs it)
    float kreg[DPW / 2];
#pragma unroll
    for (int ks = 0; ks < DPW / 2; ++ks) {
        const int d = d0 + 2 * ks + hi;
        kreg[ks] = (kok && d < D) ? load_as_float(p.k, kbase + (int64_t)key * D + d, p.in_prec) : 0.0f;
    }
    const float c = p.scale * UMFA_LOG2E;
    const uint32_t ntiles = (p.Sq + 31) / 32;
    const uint32_t t0 = p.causal ? kb : 0;  // query tiles entirely above this key block see nothing

    auto run = [&](auto SW) {  // 0: dV = P^T dO;  1: dK = scale dS^T Q  (two instantiations: V's slice is live in the second only)
        constexpr int sweep = decltype(SW)::value;
        float vreg[sweep ? DPW / 2 : 1];
        if constexpr (sweep == 1) {
#pragma unroll
            for (int ks = 0; ks < DPW / 2; ++ks) {
                const int d = d0 + 2 * ks + hi;
                vreg[ks] = (kok && d < D) ? load_as_float(p.v, kbase + (int64_t)key * D + d, p.in_prec) : 0.0f;
            }
        }
        (void)vreg;
        f32x16 acc[NDB];
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
        for (uint32_t t = t0; t < ntiles; ++t) {
            // S[q][key]: rows = queries (registers), columns = keys (lanes)
            stage_slice<DPW>(Ts, p.q, qbase, t * 32, p.Sq, D, d0, p.in_prec, lane);
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < DPW / 2; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x2f32(Ts[kl * LDK + 2 * ks + hi], kreg[ks], s, 0, 0, 0);
            sum_over_waves(s, Sx, wave, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t qrow = t * 32 + acc_row(r, hi);
                float pr = qrow < p.Sq ? exp2f(s[r] * c - p.lse[(int64_t)bh * p.Sq + qrow] * UMFA_LOG2E) : 0.0f;
                if (p.causal && key > qrow) pr = 0.0f;
                s[r] = pr;
            }
            stage_slice<DPW>(Ts, p.dout, qbase, t * 32, p.Sq, D, d0, p.dout_prec, lane);
            if constexpr (sweep == 0) {
                // dV^T[d][key] += dO^T[d][q] P[q][key]
#pragma unroll
                for (int i = 0; i < NDB; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ts[acc_row(r, hi) * LDK + 32 * i + kl], s[r], acc[i], 0, 0, 0);
            } else {
                f32x16 dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) dp[r] = 0.0f;
#pragma unroll
                for (int ks = 0; ks < DPW / 2; ++ks) dp = __builtin_amdgcn_mfma_f32_32x32x2f32(Ts[kl * LDK + 2 * ks + hi], vreg[ks], dp, 0, 0, 0);
                sum_over_waves(dp, Sx, wave, lane);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t qrow = t * 32 + acc_row(r, hi);
                    s[r] = s[r] * (dp[r] - (qrow < p.Sq ? p.dvec[(int64_t)bh * p.Sq + qrow] : 0.0f));  // dS
                }
                stage_slice<DPW>(Ts, p.q, qbase, t * 32, p.Sq, D, d0, p.in_prec, lane);  // (the dO slice sat where Q was)
                // dK^T[d][key] += Q^T[d][q] dS[q][key]
#pragma unroll
                for (int i = 0; i < NDB; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ts[acc_row(r, hi) * LDK + 32 * i + kl], s[r], acc[i], 0, 0, 0);
            }
        }
        if (kok) {
            float* const dst = sweep == 0 ? p.dv : p.dk;
            const float mul = sweep == 0 ? 1.0f : p.scale;
#pragma unroll
            for (int i = 0; i < NDB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int d = d0 + 32 * i + acc_row(r, hi);
                    if (d < D) dst[kbase + (int64_t)key * D + d] = acc[i][r] * mul;
                }
        }
    };
    run(std::integral_constant<int, 0>{});
    run(std::integral_constant<int, 1>{});
}

template <int DPW>
static hipError_t launch_wide(const BwdParams& p, hipStream_t stream) {
    const int64_t rows = (int64_t)p.B * p.H * p.Sq;
    const int ph = p.phases ? p.phases : 7;
    if (ph & 1) hipLaunchKernelGGL(bwd_wide_delta_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, p);
    const size_t lds = (4 * 32 * (DPW + 1) + 4 * 16 * 64) * sizeof(float);
    if (hipError_t e = ensure_dynamic_lds((const void*)bwd_wide_dq_kernel<DPW>, lds); e != hipSuccess) return e;
    if (hipError_t e = ensure_dynamic_lds((const void*)bwd_wide_dkdv_kernel<DPW>, lds); e != hipSuccess) return e;
    const uint32_t nqb = (p.Sq + 31) / 32, nkb = (p.Skv + 31) / 32;
    if (ph & 2) hipLaunchKernelGGL(bwd_wide_dq_kernel<DPW>, dim3(nqb * p.B * p.H), dim3(256), lds, stream, p);
    if (ph & 4) hipLaunchKernelGGL(bwd_wide_dkdv_kernel<DPW>, dim3(nkb * p.B * p.H), dim3(256), lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_bwd_wide(const BwdParams& p, hipStream_t stream, const char** name) {
    *name = "none";
    if (p.D <= 256 || p.D > 1024 || p.mask || p.units) return hipErrorInvalidValue;
    if ((uint64_t)p.B * p.H * ((p.Sq + 31) / 32) > 0x7fffffffull || (uint64_t)p.B * p.H * ((p.Skv + 31) / 32) > 0x7fffffffull) return hipErrorInvalidValue;
    if (p.D <= 512) { *name = "fa_bwd_wide<512>"; return launch_wide<128>(p, stream); }
    *name = "fa_bwd_wide<1024>";
    return launch_wide<256>(p, stream);
}

}  // namespace umfa
