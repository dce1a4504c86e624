// fa_fwd_16_pv.hip -- the 128-row forward kernel for bf16 operands with the P V product in fp16 (FwdParams::pv16, the default bf16
// arithmetic: the bf16-input forward inside the north-star's 1e-3; fa_fwd_16_kernel.h PV16).  Its own translation unit: 64
// instantiations of the kernel template that compile beside fa_fwd_16.hip's instead of behind them.
#include "fa_fwd_16_launch.h"

namespace umfa {

template <int DP, bool CAUSAL, bool HAS_MASK, typename OUT>
hipError_t launch_fwd16_pv(const FwdParams& p, hipStream_t stream) {
    // pv16 = 1: V tiles arrive by LDS-DMA like K and are converted bf16 -> fp16 in place in LDS (fa_fwd_16_kernel.h VCONV); pv16 = 2: p.v is
    // the dense fp16 image the runtime's cast pre-pass wrote.  head_dim 128, non-causal, no mask: 32-key tiles / three resident workgroups
    // per CU as in the bf16 P V kernel (fa_fwd_16.hip) -- possible since the conversion needs no staging registers (lab option bn64: off) --
    // for launches of at least two workgroups per CU without a split plan.  Graph-replayed us, 32- / 64-key tiles (profiles/r4/bn32_probe.jsonl):
    // B16 H16 Sq512 Skv512 51.8 / 55.2, B4 H32 Sq768 Skv2048 120 / 129, FLUX 218 / 225, B1 H24 Sq4096 Skv77 19.6 / 24.1; split launches lose
    // (twice the barriers per part: B1 H2 S4096 47.1 / 40.7, B2 H8 S1024 27.9 / 24.3), so they keep 64-key tiles.
    const bool dma = (int)p.D == DP && dma_enabled();
    if constexpr (!CAUSAL && !HAS_MASK && (DP == 64 || DP == 128)) {
        if (dma && fwd16_decode_form(p))  // decode form: four key quarters per tile (fa_fwd_16_kernel.h KS = 4)
            return p.pv16 == 2 ? launch_dma<__bf16, DP, CAUSAL, HAS_MASK, OUT, true, 128, 2, 4>(p, stream)
                               : launch_dma<__bf16, DP, CAUSAL, HAS_MASK, OUT, true, 128, 1, 4>(p, stream);
    }
    if constexpr (DP == 128 && !HAS_MASK && !CAUSAL) {
        const uint64_t items = (uint64_t)((p.Sq + 127) / 128) * p.B * p.H;
        if (dma && !tuning().bn64.load(std::memory_order_relaxed) && (p.nsplit < 2 || !p.part_buf) && items >= 2 * (uint64_t)device_cu_count())
            return p.pv16 == 2 ? launch_dma<__bf16, DP, CAUSAL, HAS_MASK, OUT, true, 32, 2>(p, stream)
                               : launch_dma<__bf16, DP, CAUSAL, HAS_MASK, OUT, true, 32, 1>(p, stream);
    }
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
    if constexpr (CAUSAL && !HAS_MASK && (DP == 64 || DP == 128)) {
        if (dma && p.cbal && p.part_buf && p.part_cnt)  // balanced causal pairs (fwd_16_split_plan)
            return p.pv16 == 2 ? launch_dma<__bf16, DP, CAUSAL, HAS_MASK, OUT, true, 64, 2, 1, 0, true>(p, stream)
                               : launch_dma<__bf16, DP, CAUSAL, HAS_MASK, OUT, true, 64, 1, 1, 0, true>(p, stream);
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
