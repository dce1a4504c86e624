// fa_fwd_16_launch.h -- the one launcher of fa_fwd16_kernel instantiations, shared by fa_fwd_16.hip (bf16 / fp16 P V) and
// fa_fwd_16_pv.hip (bf16 operands with the P V product in fp16: its own translation unit so the two sets compile in parallel).
#pragma once
#include "fa_fwd_16_kernel.h"
#include "kernels.h"

namespace umfa {

template <typename T, int DP, bool CAUSAL, bool HAS_MASK, typename OUT, bool DMA, int BN, int PV16 = 0, int KS = 1, int PIPE = 0, bool CBAL = false>
static inline hipError_t launch_dma(const FwdParams& pin, hipStream_t stream) {
    FwdParams p = pin;
    constexpr uint32_t BM = KS == 4 ? 32 : 128;  // (KS = 4, the decode form: items of 32 query rows)
    const uint32_t nqb = (p.Sq + BM - 1) / BM;
    const uint32_t items = nqb * p.B * p.H;
    if ((KS != 1 && KS != 4) || PIPE || CBAL || p.nsplit < 2 || !p.part_buf || !p.part_cnt) { p.n_full = items; p.nsplit = 1; }
    if (CBAL && (!p.part_buf || !p.part_cnt || nqb < 2)) return hipErrorInvalidValue;  // (the plan's scratch: one slot and one flag per pair)
    const uint32_t grid = p.n_full + (items - p.n_full) * p.nsplit;
    // (p.part_cnt is zero on entry and on exit: the runtime zeroes a ticket block once, the folding workgroup resets its word)
    // 2 x ring depth (fa_fwd_16_kernel.h NS: 2, key-split form 4) tiles; the key halves' exchange (4 x 34 x 256 bytes at head_dim 64) fits inside
#ifdef UMFA_LAB_NS
    const size_t lds = 2 * ((CBAL || CAUSAL || HAS_MASK) ? 2 : UMFA_LAB_NS) * BN * DP * 2;  // lab: ring depth of the LDS-DMA staging (fa_fwd_16_kernel.h NS)
#else
    const size_t lds = (KS == 2 ? 8 : 4) * BN * DP * 2;
#endif
    static_assert(KS != 2 || 8 * BN * DP * 2 >= 4 * (16 * (DP / 32) + 2) * 256, "exchange area");
    static_assert(KS != 4 || 4 * BN * DP * 2 >= 3 * (16 * (DP / 32) + 2) * 256 + 256, "exchange area (decode form)");
    auto kfn = fa_fwd16_kernel<T, DP, CAUSAL, HAS_MASK, OUT, DMA, BN, PV16, KS, PIPE, CBAL>;
    if (hipError_t e = ensure_dynamic_lds((const void*)kfn, lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(fwd16_threads(KS)), lds, stream, p);
    return hipGetLastError();
}

// The two other forms of the head_dim-64 kernel (fa_fwd_16_kernel.h): calls whose K / V rows are exactly 64 elements (LDS-DMA
// staging), without a mask tensor / window, and without a split-KV plan.
//   1  software-pipelined loop (PIPE: the next tile's Q K^T inside this tile's softmax) -- bit-identical results, 0.80 ... 1.00 x the plain form's speed
//   2  key-split (KS = 2: eight waves per workgroup, option "ksplit") -- 0.94 ... 1.07 x
static inline int fwd16_d64_form(const FwdParams& p) {
#ifndef UMFA_D64_FORMS
    return 0;  // neither form is in the product build: both measured null or slower (profiles/r4/lab_notes.md section 2b); -DUMFA_D64_FORMS builds them in
#endif
    if (p.D != 64 || p.mask_kind != MK_NONE || tuning().no_dma.load(std::memory_order_relaxed) || !(p.nsplit < 2 || !p.part_buf || !p.part_cnt)) return 0;
    if (tuning().ksplit.load(std::memory_order_relaxed)) return 2;
    return tuning().no_pipe.load(std::memory_order_relaxed) ? 0 : 1;
}

static inline bool dma_enabled() { return !tuning().no_dma.load(std::memory_order_relaxed); }

// the decode form (fa_fwd_16_kernel.h KS = 4): at most 32 query rows per (batch, head), no mask, not causal, LDS-DMA staging (the caller checks the head dim);
// option decode_ks: 0 = where it applies, 2 = never
static inline bool fwd16_decode_shape(const FwdParams& p) {
    return p.Sq >= 1 && p.Sq <= 32 && !p.causal && p.mask_kind == MK_NONE && (p.D == 64 || p.D == 128) && dma_enabled() &&
           tuning().decode_ks.load(std::memory_order_relaxed) != 2;
}
// ... and the launch takes it when the plan said so (fwd_16_split_plan sets FwdSplitPlan::decode, the runtime copies it)
static inline bool fwd16_decode_form(const FwdParams& p) { return p.decode_form != 0 && fwd16_decode_shape(p); }


// bf16 operands, fp16 P V (FwdParams::pv16 = 1: V converted in the kernel; 2: p.v is the fp16 image of the cast pre-pass):
// defined and explicitly instantiated in fa_fwd_16_pv.hip
template <int DP, bool CAUSAL, bool HAS_MASK, typename OUT>
hipError_t launch_fwd16_pv(const FwdParams& p, hipStream_t stream);

}  // namespace umfa
