// kernels.h -- host-callable launchers of the gfx950 kernels (all asynchronous on `stream`).
#pragma once
#include <hip/hip_runtime.h>

#include "fa_common.h"

namespace umfa {

// fp32-exact forward (any input type, any D <= 256, masks, causal, LSE).
hipError_t launch_fwd_exact(const FwdParams& p, hipStream_t stream, const char** name);

// bf16 / fp16 MFMA forward.  Requires in_prec in {FP16, BF16}, D % 8 == 0, D <= 256,
// 16-byte aligned operands/strides, scale > 0.  Returns hipErrorNotSupported otherwise
// (the caller then takes the exact path).
hipError_t launch_fwd_16(const FwdParams& p, hipStream_t stream, const char** name);
bool fwd_16_supported(const FwdParams& p);

// Backward: D = rowsum(dO o O), then dK/dV and dQ.
hipError_t launch_bwd(const BwdParams& p, hipStream_t stream, const char** name);

// int8 path: fused symmetric quantiser for Q, K, V (one launch) + int8-QK^T forward.
struct QuantWorkspace {
    int8_t* q8;
    int8_t* k8;
    void* v16;      // V fake-quantised, stored in the 16-bit input type
    float* q_scale; // per (b,h,block)
    float* k_scale;
    float* v_scale;
    uint32_t blk;   // rows per block (0: per tensor)
};
size_t quant_workspace_bytes(uint32_t B, uint32_t H, uint32_t Sq, uint32_t Skv, uint32_t D);
hipError_t launch_quantized_fwd(const FwdParams& p, int bits, int quant_mode, void* workspace,
                                size_t workspace_bytes, hipStream_t stream, const char** name);

}  // namespace umfa
