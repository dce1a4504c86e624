// fa_fwd_16_pv.hip -- the 128-row forward kernel for bf16 operands with the P V product in fp16 (FwdParams::pv16, the default bf16
// arithmetic: the bf16-input forward inside the north-star's 1e-3; fa_fwd_16_kernel.h PV16).  Its own translation unit: 64
// instantiations of the kernel template that compile beside fa_fwd_16.hip's instead of behind them.
#include "fa_fwd_16_launch.h"

namespace umfa {

template <int DP, bool CAUSAL, bool HAS_MASK, typename OUT>
hipError_t launch_fwd16_pv(const FwdParams& p, hipStream_t stream) {
    // pv16 = 1: V tiles go through registers (converted on the way), K keeps LDS-DMA; pv16 = 2: p.v is the dense fp16 image the
    // runtime's cast pre-pass wrote, staged like K.  64-key tiles throughout (the 32-key / three-workgroup variant of the bf16
    // P V kernel has no room for V staging registers).
    const bool dma = (int)p.D == DP && dma_enabled();
    if constexpr (DP == 64 && !HAS_MASK) {
#ifdef UMFA_D64_FORMS
        const int form = dma ? fwd16_d64_form(p) : 0;
        if (form == 1)
            return p.pv16 == 2 ? launch_dma<__bf16, DP, CAUSAL, HAS_MASK, OUT, true, 64, 2, 1, 1>(p, stream)
                               : launch_dma<__bf16, DP, CAUSAL, HAS_MASK, OUT, true, 64, 1, 1, 1>(p, stream);
        if (form == 2)
            return p.pv16 == 2 ? launch_dma<__bf16, DP, CAUSAL, HAS_MASK, OUT, true, 64, 2, 2>(p, stream)
                               : launch_dma<__bf16, DP, CAUSAL, HAS_MASK, OUT, true, 64, 1, 2>(p, stream);
#endif
    }
    if (p.pv16 == 2)
        return dma ? launch_dma<__bf16, DP, CAUSAL, HAS_MASK, OUT, true, 64, 2>(p, stream) : launch_dma<__bf16, DP, CAUSAL, HAS_MASK, OUT, false, 64, 2>(p, stream);
    return dma ? launch_dma<__bf16, DP, CAUSAL, HAS_MASK, OUT, true, 64, 1>(p, stream) : launch_dma<__bf16, DP, CAUSAL, HAS_MASK, OUT, false, 64, 1>(p, stream);
}

#define UMFA_PV_INST(DP, OUT)                                                                  \
    template hipError_t launch_fwd16_pv<DP, false, false, OUT>(const FwdParams&, hipStream_t); \
    template hipError_t launch_fwd16_pv<DP, false, true, OUT>(const FwdParams&, hipStream_t);  \
    template hipError_t launch_fwd16_pv<DP, true, false, OUT>(const FwdParams&, hipStream_t);  \
    template hipError_t launch_fwd16_pv<DP, true, true, OUT>(const FwdParams&, hipStream_t);
UMFA_PV_INST(32, float) UMFA_PV_INST(32, __bf16) UMFA_PV_INST(64, float) UMFA_PV_INST(64, __bf16)
UMFA_PV_INST(128, float) UMFA_PV_INST(128, __bf16) UMFA_PV_INST(256, float) UMFA_PV_INST(256, __bf16)
#undef UMFA_PV_INST

}  // namespace umfa
