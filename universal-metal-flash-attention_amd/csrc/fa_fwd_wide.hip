// fa_fwd_wide.hip -- SDPA forward for head dims 257 ... 1024, fp32 arithmetic, any operand type / strides / mask / causal / LSE.
//
// The reference's callers admit head_dim <= 1024 (examples/pytorch-custom-op-ffi/src/metal_sdpa_backend.cpp:1078-1086, :1382-1384) and
// hand such calls to the same `attention` forward dispatch as every other (MFABridge.swift:1395-1408).  Nothing in the reference's own
// tests or models uses them, so this is the domain-completing path, not a tuned one: correct to the fp32-exact kernel's standard
// (operands converted to fp32 on load, both products on v_mfma_f32_32x32x2_f32 = an fp32 fma chain), slow by design.
//
// Layout: workgroup = 4 waves = 32 query rows of one (batch, head).  The HEAD DIM is what the waves share: wave w owns columns
// [w DPW, (w + 1) DPW) -- its slice of Q^T in registers, its slice of O^T as accumulators (32 rows x 1024 columns of fp32 do not fit one
// wave's registers), and its slice of every 32-key tile of K, then V, in LDS (a slice is private to its wave: 32 x 256 fp32 = 32 KB).
// S = K Q^T is a sum over the head dim: each wave forms the partial product over its columns, the four partials meet in LDS and are
// added in wave order (deterministic), after which every wave holds the full 32 x 32 score tile and runs the same online softmax --
// redundantly, so m and l never need another exchange -- and multiplies P into its own V columns.
#include "fa_common.h"
#include "kernels.h"

namespace umfa {

template <int DPW>
__global__ __launch_bounds__(256, 1) void fa_fwd_wide_kernel(FwdParams p) {
    constexpr int BM = 32, BN = 32, LDK = DPW + 1, NDB = DPW / 32;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, ql = lane & 31, hi = lane >> 5;
    float* const Ts = smem_f + wave * (BN * LDK);  // this wave's column slice of the K tile, then of the V tile
    float* const Sx = smem_f + 4 * (BN * LDK);     // [wave 4][register 16][lane 64]: partial scores

    const uint32_t nqb = (p.Sq + BM - 1) / BM;
    const uint32_t vid = xcd_remap(blockIdx.x, nqb * p.B * p.H);
    const uint32_t bh = vid / nqb;
    uint32_t qb = vid % nqb;
    if (p.causal) qb = nqb - 1 - qb;  // heaviest query blocks first
    const uint32_t b = bh / p.H, h = bh % p.H;
    const uint32_t q_row = qb * BM + ql;
    const int D = (int)p.D, d0 = wave * DPW;

    // Q^T slice as the B operand: lane (q, hi) holds Q[q][d0 + 2 ks + hi]
    float qreg[DPW / 2];
    {
        const int64_t qoff = (int64_t)b * p.qs[0] + (int64_t)h * p.qs[1] + (int64_t)q_row * p.qs[2];
#pragma unroll
        for (int ks = 0; ks < DPW / 2; ++ks) {
            const int d = d0 + 2 * ks + hi;
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

    // one 32-key tile's column slice -> this wave's LDS rows (lanes run along the head dim: coalesced when it is contiguous)
    auto stage = [&](const void* src, int64_t base, const int64_t* st, uint32_t t) {
        for (int idx = lane; idx < BN * DPW; idx += 64) {
            const int row = idx / DPW, dd = idx - row * DPW, d = d0 + dd;
            const uint32_t key = t * BN + row;
            Ts[row * LDK + dd] = (key < p.Skv && d < D) ? load_as_float(src, base + (int64_t)key * st[2] + (int64_t)d * st[3], p.in_prec) : 0.0f;
        }
    };

    for (uint32_t t = 0; t < ntiles; ++t) {
        __syncthreads();  // the previous tile's V slice and partial scores are consumed
        stage(p.k, kbase, p.ks, t);
        __syncthreads();
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < DPW / 2; ++ks)
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(Ts[ql * LDK + 2 * ks + hi], qreg[ks], s, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) Sx[(wave * 16 + r) * 64 + lane] = s[r];
        __syncthreads();
        stage(p.v, vbase, p.vs, t);  // (this wave's K slice is read: its rows can take V)
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // the four column slices' partial dot products, in wave order
            const float full = ((Sx[(0 * 16 + r) * 64 + lane] + Sx[(1 * 16 + r) * 64 + lane]) + Sx[(2 * 16 + r) * 64 + lane]) + Sx[(3 * 16 + r) * 64 + lane];
            const uint32_t key = t * BN + acc_row(r, hi);
            float tv = full * c;
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
        __syncthreads();  // the V slice is staged
        // O^T (this wave's columns) += V^T P^T : k index of the 32x32x2 MFMA = lane half = key acc_row(r, hi)
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ts[acc_row(r, hi) * LDK + 32 * i + ql], s[r], acc[i], 0, 0, 0);
    }

    const float lt = l + xor32(l);
    const float inv = lt > 0.0f ? 1.0f / lt : 0.0f;
    if (q_row < p.Sq) {
        const int64_t orow = (int64_t)bh * p.Sq * D + (int64_t)q_row * p.os[0];
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = d0 + 32 * i + acc_row(r, hi);
                if (d < D) {
                    const float val = acc[i][r] * inv;
                    const int64_t oi = orow + (int64_t)d * p.os[1];
                    if (p.out_prec == P_FP32) ((float*)p.o)[oi] = val;
                    else if (p.out_prec == P_FP16) ((_Float16*)p.o)[oi] = (_Float16)val;
                    else ((__bf16*)p.o)[oi] = (__bf16)val;
                }
            }
        if (p.lse && wave == 0 && hi == 0)
            p.lse[(int64_t)bh * p.Sq + q_row] = lt > 0.0f ? (m + log2f(lt)) * UMFA_LN2 : -INFINITY;
    }
}

template <int DPW>
static hipError_t launch_wide(const FwdParams& p, hipStream_t stream) {
    const uint32_t nqb = (p.Sq + 31) / 32;
    const size_t lds = (4 * 32 * (DPW + 1) + 4 * 16 * 64) * sizeof(float);
    if (hipError_t e = ensure_dynamic_lds((const void*)fa_fwd_wide_kernel<DPW>, lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(fa_fwd_wide_kernel<DPW>, dim3(nqb * p.B * p.H), dim3(256), lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_fwd_wide(const FwdParams& p, hipStream_t stream, const char** name) {
    if (p.D <= 256 || p.D > 1024) return hipErrorInvalidValue;
    if (p.D <= 512) { *name = "fa_fwd_wide<512>"; return launch_wide<128>(p, stream); }
    *name = "fa_fwd_wide<1024>";
    return launch_wide<256>(p, stream);
}

}  // namespace umfa
