// fa_fwd16_w64_bias.hip -- ADDITIVE fp16 mask tensors on the one-wave-per-SIMD forward (round 6): the MASKA instantiations of the kernel text of
// fa_fwd16_w64.hip (fa_fwd16_w64_kernel.inc) over generated bodies of their own (tools/gen_w64_body.py Cfg(madd=True) -> fa_fwd16_w64_bias_body.inc):
// every tile body carries the eight LDS-DMA instructions that bring the wave's 64 x 64 tile of the caller's mask for the NEXT listed tile into a two-slot
// ring (the wave's output staging area, idle until the epilogue), and the masking bodies read it back (ds_read_b64 per four scores, two groups ahead) and add
// mask / scale to the raw scores with one v_fma_mix_f32 per score.  Tile classes and block lists: fa_aux.hip mask_classify_f16_kernel + mask_list_kernel.
// Replaces: the reference's additive-mask path (MFABridge.swift:157-242 mfa_prepare_mask, a dense fp32 expansion + commit-and-wait pre-pass there).
#include "fa_fwd16_w64_params.h"

namespace umfa {

#define W64_I8 0
#define W64_MADD 1
#define W64_VSC 1
#define W64_BODY_INC "fa_fwd16_w64_bias_body.inc"
// bf16 Q / K / V, P V in fp16 against the fp16 image of V (the default bf16 arithmetic)
#define W64_T __bf16
#define W64_MFMA "v_mfma_f32_32x32x16_f16"
#define W64_MFMA_QK "v_mfma_f32_32x32x16_bf16"
#define W64_MSUM "v_mfma_f32_4x4x4_16b_f16"
#define W64_ONES_BITS 0x3c003c00u
#define W64_LAZY_PARTS 2
#define W64_CVT "v_cvt_pk_f16_f32"
#define W64_KERNEL fa_fwd16_w64_bias_bf16pv16
#include "fa_fwd16_w64_kernel.inc"
#undef W64_VSC
#define W64_VSC 0
#undef W64_T
#undef W64_MFMA_QK
#undef W64_KERNEL
// fp16 operands
#define W64_T _Float16
#define W64_MFMA_QK "v_mfma_f32_32x32x16_f16"
#define W64_KERNEL fa_fwd16_w64_bias_f16
#include "fa_fwd16_w64_kernel.inc"

#undef W64_T
#undef W64_MFMA_QK
#undef W64_KERNEL
#undef W64_BODY_INC

// head_dim 64 (the generator's Cfg(d64=True, madd=True)): the same two families
#undef W64_DP
#define W64_DP 64
#define W64_BODY_INC "fa_fwd16_w64d64_bias_body.inc"
#undef W64_VSC
#define W64_VSC 1
#define W64_T __bf16
#define W64_MFMA_QK "v_mfma_f32_32x32x16_bf16"
#define W64_KERNEL fa_fwd16_w64d64_bias_bf16pv16
#include "fa_fwd16_w64_kernel.inc"
#undef W64_VSC
#define W64_VSC 0
#undef W64_T
#undef W64_MFMA_QK
#undef W64_KERNEL
#define W64_T _Float16
#define W64_MFMA_QK "v_mfma_f32_32x32x16_f16"
#define W64_KERNEL fa_fwd16_w64d64_bias_f16
#include "fa_fwd16_w64_kernel.inc"

template <typename KFN>
static hipError_t launch_bias_kernel(KFN kfn, const W64Params& wp, uint32_t grid, size_t lds, hipStream_t stream) {
    if (hipError_t e = ensure_dynamic_lds((const void*)kfn, lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds, stream, wp);
    return hipGetLastError();
}

hipError_t launch_fwd_w64_bias(const W64Params& wp, int family, bool fp32_out, uint32_t grid, size_t lds, hipStream_t stream, int head_dim) {
    if (head_dim == 64) {
        if (family == 1)
            return fp32_out ? launch_bias_kernel(fa_fwd16_w64d64_bias_bf16pv16<float, false, false, false, true, true>, wp, grid, lds, stream)
                            : launch_bias_kernel(fa_fwd16_w64d64_bias_bf16pv16<__bf16, false, false, false, true, true>, wp, grid, lds, stream);
        if (family == 2)
            return fp32_out ? launch_bias_kernel(fa_fwd16_w64d64_bias_f16<float, false, false, false, true, true>, wp, grid, lds, stream)
                            : launch_bias_kernel(fa_fwd16_w64d64_bias_f16<_Float16, false, false, false, true, true>, wp, grid, lds, stream);
        return hipErrorNotSupported;
    }
    if (family == 1)
        return fp32_out ? launch_bias_kernel(fa_fwd16_w64_bias_bf16pv16<float, false, false, false, true, true>, wp, grid, lds, stream)
                        : launch_bias_kernel(fa_fwd16_w64_bias_bf16pv16<__bf16, false, false, false, true, true>, wp, grid, lds, stream);
    if (family == 2)
        return fp32_out ? launch_bias_kernel(fa_fwd16_w64_bias_f16<float, false, false, false, true, true>, wp, grid, lds, stream)
                        : launch_bias_kernel(fa_fwd16_w64_bias_f16<_Float16, false, false, false, true, true>, wp, grid, lds, stream);
    return hipErrorNotSupported;
}

}  // namespace umfa
