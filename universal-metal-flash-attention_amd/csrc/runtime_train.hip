// runtime_train.hip -- backward and runtime-quantised entry points of the C ABI.
#include <hip/hip_runtime.h>
#include "../../include/umfa_abi.h"
#include "fa_common.h"
#include "kernels.h"

extern "C" {

mfa_error_t mfa_attention_backward(mfa_context_t, mfa_buffer_t, mfa_buffer_t, mfa_buffer_t, mfa_buffer_t, mfa_buffer_t,
                                   mfa_buffer_t, mfa_buffer_t, mfa_buffer_t, mfa_buffer_t, mfa_buffer_t, uint32_t,
                                   uint32_t, uint32_t, uint32_t, uint16_t, float, bool, mfa_precision_t,
                                   mfa_precision_t, bool, bool, bool, bool) {
    return MFA_ERROR_EXECUTION_FAILED;
}
int32_t mfa_quantized_forward_with_lse(mfa_context_t, mfa_buffer_t, mfa_buffer_t, mfa_buffer_t, mfa_buffer_t,
                                       mfa_buffer_t, mfa_buffer_t, uint32_t, uint32_t, uint32_t, uint32_t, uint16_t,
                                       float, bool, int32_t, int32_t, int32_t) {
    return MFA_ERROR_EXECUTION_FAILED;
}
int32_t mfa_quantized_backward(mfa_context_t, mfa_buffer_t, mfa_buffer_t, mfa_buffer_t, mfa_buffer_t, mfa_buffer_t,
                               mfa_buffer_t, mfa_buffer_t, mfa_buffer_t, mfa_buffer_t, mfa_buffer_t, uint32_t,
                               uint32_t, uint32_t, uint32_t, uint16_t, float, bool, int32_t, int32_t, int32_t) {
    return MFA_ERROR_EXECUTION_FAILED;
}
}
