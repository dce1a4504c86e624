// fa_bwd.hip -- backward kernels (placeholder until the MFMA backward lands).
#include "fa_common.h"
#include "kernels.h"
namespace umfa {
hipError_t launch_bwd(const BwdParams&, hipStream_t, const char** name) {
    *name = "none";
    return hipErrorNotSupported;
}
}  // namespace umfa
