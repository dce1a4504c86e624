// fa_fwd_16_launch.h -- the one launcher of fa_fwd16_kernel instantiations, shared by fa_fwd_16.hip (bf16 / fp16 P V) and
// fa_fwd_16_pv.hip (bf16 operands with the P V product in fp16: its own translation unit so the two sets compile in parallel).
#pragma once
#include "fa_fwd_16_kernel.h"
#include "kernels.h"

namespace umfa {

template <typename T, int DP, bool CAUSAL, bool HAS_MASK, typename OUT, bool DMA, int BN, int PV16 = 0>
static inline hipError_t launch_dma(const FwdParams& pin, hipStream_t stream) {
    FwdParams p = pin;
    const uint32_t nqb = (p.Sq + 127) / 128;
    const uint32_t items = nqb * p.B * p.H;
    if (p.nsplit < 2 || !p.part_buf || !p.part_cnt) { p.n_full = items; p.nsplit = 1; }
    const uint32_t grid = p.n_full + (items - p.n_full) * p.nsplit;
    // (p.part_cnt is zero on entry and on exit: the runtime zeroes a ticket block once, the folding workgroup resets its word)
    const size_t lds = 4 * BN * DP * 2;  // 2 x ring depth (fa_fwd_16_kernel.h NS = 2) tiles
    auto kfn = fa_fwd16_kernel<T, DP, CAUSAL, HAS_MASK, OUT, DMA, BN, PV16>;
    if (hipError_t e = ensure_dynamic_lds((const void*)kfn, lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds, stream, p);
    return hipGetLastError();
}

static inline bool dma_enabled() { return !tuning().no_dma.load(std::memory_order_relaxed); }


// bf16 operands, fp16 P V (FwdParams::pv16 = 1: V converted in the kernel; 2: p.v is the fp16 image of the cast pre-pass):
// defined and explicitly instantiated in fa_fwd_16_pv.hip
template <int DP, bool CAUSAL, bool HAS_MASK, typename OUT>
hipError_t launch_fwd16_pv(const FwdParams& p, hipStream_t stream);

}  // namespace umfa
