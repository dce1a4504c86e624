// fa_quant.hip -- int8/int4 quantised path (placeholder until the int8 MFMA forward lands).
#include "fa_common.h"
#include "kernels.h"
namespace umfa {
size_t quant_workspace_bytes(uint32_t, uint32_t, uint32_t, uint32_t, uint32_t) { return 0; }
hipError_t launch_quantized_fwd(const FwdParams&, int, int, void*, size_t, hipStream_t, const char** name) {
    *name = "none";
    return hipErrorNotSupported;
}
}  // namespace umfa
