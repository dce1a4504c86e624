// fa_fwd_exact.hip -- fp32-exact SDPA forward for gfx950.
//
// Replaces the reference's FP32 attention kernel (launched at MFABridge.swift:1395-1408 /
// MultiHeadAttention.forward, :2240-2248) for every call whose arithmetic must be fp32:
// fp32 inputs (config 1, test_scale_factor_fix.py tolerance 1e-5) or any input type with
// intermediate_precision == FP32.  Operands are converted to fp32 on load and both GEMMs run on
// v_mfma_f32_32x32x2_f32, which is bit-for-bit an fp32 fma chain (MI355X_MICROARCH.md, Matrix
// cores) -- 1/16 of the bf16 rate, so this is the parity path, not the headline path.
//
// Layout: one workgroup = 4 waves = 128 query rows of one (batch, head); key tiles of 32 rows
// are staged as fp32 in LDS (rows padded by one dword: conflict-free column reads).
#include "fa_common.h"
#include "kernels.h"

namespace umfa {

template <int DP>
__global__ __launch_bounds__(256) void fa_fwd_exact_kernel(FwdParams p) {
    constexpr int BM = 128, BN = 32, LDK = DP + 1, NDB = DP / 32;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float* Ks = smem_f;
    float* Vs = smem_f + BN * LDK;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, ql = lane & 31, hi = lane >> 5;
    const uint32_t nqb = (p.Sq + BM - 1) / BM;
    const uint32_t vid = xcd_remap(blockIdx.x, nqb * p.B * p.H);
    const uint32_t bh = vid / nqb;
    uint32_t qb = vid % nqb;
    if (p.causal) qb = nqb - 1 - qb;  // heaviest query blocks first
    const uint32_t b = bh / p.H, h = bh % p.H;
    const uint32_t q_row = qb * BM + wave * 32 + ql;
    const uint32_t wave_qmax = qb * BM + wave * 32 + 31;
    const int D = (int)p.D;

    // Q^T as the B operand: lane (q, hi) holds Q[q][2*ks + hi]
    float qreg[DP / 2];
    {
        const int64_t qoff = (int64_t)b * p.qs[0] + (int64_t)h * p.qs[1] + (int64_t)q_row * p.qs[2];
#pragma unroll
        for (int ks = 0; ks < DP / 2; ++ks) {
            const int d = 2 * ks + hi;
            qreg[ks] = (q_row < p.Sq && d < D) ? load_as_float(p.q, qoff + (int64_t)d * p.qs[3], p.in_prec) : 0.0f;
        }
    }

    f32x16 acc[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    float m = -INFINITY, l = 0.0f;
    const float c = p.scale * UMFA_LOG2E;

    uint32_t ntiles = (p.Skv + BN - 1) / BN;
    if (p.causal) {
        const uint32_t lim = (qb * BM + BM + BN - 1) / BN;
        ntiles = ntiles < lim ? ntiles : lim;
    }
    const int64_t kbase = (int64_t)b * p.ks[0] + (int64_t)h * p.ks[1];
    const int64_t vbase = (int64_t)b * p.vs[0] + (int64_t)h * p.vs[1];
    const int64_t mbase = (int64_t)b * p.ms[0] + (int64_t)h * p.ms[1] + (int64_t)q_row * p.ms[2];

    for (uint32_t t = 0; t < ntiles; ++t) {
        __syncthreads();
        for (int idx = tid; idx < BN * DP; idx += 256) {
            const int row = idx / DP, d = idx % DP;
            const uint32_t key = t * BN + row;
            const bool ok = key < p.Skv && d < D;
            Ks[row * LDK + d] = ok ? load_as_float(p.k, kbase + (int64_t)key * p.ks[2] + (int64_t)d * p.ks[3], p.in_prec) : 0.0f;
            Vs[row * LDK + d] = ok ? load_as_float(p.v, vbase + (int64_t)key * p.vs[2] + (int64_t)d * p.vs[3], p.in_prec) : 0.0f;
        }
        __syncthreads();
        if (p.causal && t * BN > wave_qmax) continue;  // wave-uniform: tile above the diagonal

        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < DP / 2; ++ks)
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[ql * LDK + 2 * ks + hi], qreg[ks], s, 0, 0, 0);

        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t key = t * BN + acc_row(r, hi);
            float tv = s[r] * c;
            if (p.mask_kind != MK_NONE && key < p.Skv && q_row < p.Sq)
                tv += p.mask_kind == MK_WINDOW ? window_term(q_row, key, p.win_left, p.win_right)
                                               : mask_term(p.mask, mbase + (int64_t)key * p.ms[3], p.mask_kind);
            if (key >= p.Skv || (p.causal && key > q_row)) tv = -INFINITY;
            s[r] = tv;
            mx = fmaxf(mx, tv);
        }
        mx = fmaxf(mx, xor32(mx));
        const float m_new = fmaxf(m, mx);
        const float m_use = m_new == -INFINITY ? 0.0f : m_new;
        const float alpha = exp2f(m - m_use);
        float rs = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = exp2f(s[r] - m_use);
            rs += s[r];
        }
        l = l * alpha + rs;
        m = m_new;
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] *= alpha;

        // O^T += V^T P^T : k index of the 32x32x2 MFMA = lane half = key acc_row(r, hi)
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[acc_row(r, hi) * LDK + 32 * i + ql], s[r],
                                                              acc[i], 0, 0, 0);
    }

    const float lt = l + xor32(l);
    const float inv = lt > 0.0f ? 1.0f / lt : 0.0f;
    if (q_row < p.Sq) {
        const int64_t orow = (int64_t)bh * p.Sq * D + (int64_t)q_row * p.os[0];
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = 32 * i + acc_row(r, hi);
                if (d < D) {
                    const float val = acc[i][r] * inv;
                    const int64_t oi = orow + (int64_t)d * p.os[1];
                    if (p.out_prec == P_FP32) ((float*)p.o)[oi] = val;
                    else if (p.out_prec == P_FP16) ((_Float16*)p.o)[oi] = (_Float16)val;
                    else ((__bf16*)p.o)[oi] = (__bf16)val;
                }
            }
        if (p.lse && hi == 0)
            p.lse[(int64_t)bh * p.Sq + q_row] = lt > 0.0f ? (m + log2f(lt)) * UMFA_LN2 : -INFINITY;
    }
}

template <int DP>
static hipError_t launch_exact(const FwdParams& p, hipStream_t stream) {
    const uint32_t nqb = (p.Sq + 127) / 128;
    const size_t lds = 2 * 32 * (DP + 1) * sizeof(float);
    if (hipError_t e = ensure_dynamic_lds((const void*)fa_fwd_exact_kernel<DP>, lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(fa_fwd_exact_kernel<DP>, dim3(nqb * p.B * p.H), dim3(256), lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_fwd_exact(const FwdParams& p, hipStream_t stream, const char** name) {
    if (p.D <= 32) { *name = "fa_fwd_exact<32>"; return launch_exact<32>(p, stream); }
    if (p.D <= 64) { *name = "fa_fwd_exact<64>"; return launch_exact<64>(p, stream); }
    if (p.D <= 128) { *name = "fa_fwd_exact<128>"; return launch_exact<128>(p, stream); }
    if (p.D <= 256) { *name = "fa_fwd_exact<256>"; return launch_exact<256>(p, stream); }
    return hipErrorInvalidValue;
}

}  // namespace umfa
