// runtime_internal.h -- host-side objects shared by runtime.hip and runtime_train.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/umfa_abi.h"
#include "fa_common.h"
#include "kernels.h"

namespace umfa_rt {

extern const bool g_debug;
#define DBG(...)                                    \
    do {                                            \
        if (umfa_rt::g_debug) {                     \
            fprintf(stderr, "[umfa] " __VA_ARGS__); \
            fputc('\n', stderr);                    \
        }                                           \
    } while (0)

// ---- context (MFAContext + GlobalContextStore, MFABridge.swift:91-150,652-687) -------------
struct Context {
    uint32_t magic = 0x4d464143;  // 'MFAC'
    int device = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double last_latency = 0.0;
    const char* last_kernel = "none";
    void* scratch = nullptr;  // host-mask staging
    size_t scratch_bytes = 0;
    void* split_buf = nullptr;  // split-KV partials + tickets (fa_fwd_16)
    size_t split_bytes = 0;
    void* workspace = nullptr;  // quantiser output (int8 Q/K, fp16 V, scales, fp32 copies for backward)
    size_t workspace_bytes = 0;
    std::vector<float> q_scales, k_scales, v_scales;  // mfa_set_scale_arrays: stored, never read
    std::atomic<int> refs{0};
    std::mutex mu;

    void* ensure_split(size_t bytes) {
        if (bytes <= split_bytes) return split_buf;
        if (split_buf) (void)hipFree(split_buf);
        split_buf = nullptr;
        split_bytes = 0;
        if (hipMalloc(&split_buf, bytes + 256) != hipSuccess) return nullptr;
        split_bytes = bytes + 256;
        return split_buf;
    }

    // fa_fwd16_w64: zeroed ticket array (kept zero by the kernel) followed by the partials buffer
    void* w64_buf = nullptr;
    size_t w64_cnt_bytes = 0, w64_buf_bytes = 0;
    void* ensure_w64(size_t cnt_bytes, size_t buf_bytes) {
        if (w64_buf && cnt_bytes <= w64_cnt_bytes && buf_bytes <= w64_buf_bytes) return w64_buf;
        if (w64_buf) (void)hipFree(w64_buf);
        w64_buf = nullptr;
        const size_t c = cnt_bytes > w64_cnt_bytes ? cnt_bytes : w64_cnt_bytes;
        const size_t b = buf_bytes > w64_buf_bytes ? buf_bytes : w64_buf_bytes;
        if (hipMalloc(&w64_buf, c + b) != hipSuccess) { w64_cnt_bytes = w64_buf_bytes = 0; return nullptr; }
        if (hipMemset(w64_buf, 0, c) != hipSuccess) { (void)hipFree(w64_buf); w64_buf = nullptr; w64_cnt_bytes = w64_buf_bytes = 0; return nullptr; }
        w64_cnt_bytes = c;
        w64_buf_bytes = b;
        return w64_buf;
    }

    void* mflag_buf = nullptr;
    size_t mflag_bytes = 0;
    void* ensure_mask_flags(size_t bytes) {
        if (bytes <= mflag_bytes) return mflag_buf;
        if (mflag_buf) (void)hipFree(mflag_buf);
        mflag_buf = nullptr;
        mflag_bytes = 0;
        if (hipMalloc(&mflag_buf, bytes + 256) != hipSuccess) return nullptr;
        mflag_bytes = bytes + 256;
        return mflag_buf;
    }

    void* ensure_workspace(size_t bytes) {
        if (bytes <= workspace_bytes) return workspace;
        if (workspace) (void)hipFree(workspace);
        workspace = nullptr;
        workspace_bytes = 0;
        size_t want = bytes + (bytes >> 3) + 256;
        if (hipMalloc(&workspace, want) != hipSuccess) return nullptr;
        workspace_bytes = want;
        return workspace;
    }

    void* ensure_scratch(size_t bytes) {
        if (bytes <= scratch_bytes) return scratch;
        if (scratch) (void)hipFree(scratch);
        scratch = nullptr;
        scratch_bytes = 0;
        size_t want = bytes + (bytes >> 2) + 256;
        if (hipMalloc(&scratch, want) != hipSuccess) return nullptr;
        scratch_bytes = want;
        return scratch;
    }
};

extern std::mutex g_ctx_mu;
extern Context* g_ctx;

inline bool device_usable(int* dev_out) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return false;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        DBG("device %d is %s, not gfx950", dev, prop.gcnArchName);
        return false;
    }
    if (dev_out) *dev_out = dev;
    return true;
}

inline Context* as_ctx(mfa_context_t c) {
    Context* x = (Context*)c;
    return (x && x->magic == 0x4d464143) ? x : nullptr;
}

// ---- buffers (MFABuffer, MFABridge.swift:720-747, 850-1070) ---------------------------------
struct Buffer {
    uint32_t magic = 0x4d464142;  // 'MFAB'
    void* host = nullptr;   // caller-visible memory (NULL for device-native wraps)
    void* dev = nullptr;    // what kernels read/write
    size_t bytes = 0;       // 0 = unknown (mfa_buffer_from_mtl_buffer with size 0)
    bool owns_host = false, owns_dev = false;
    std::vector<int64_t> shape, strides;

    hipError_t upload(hipStream_t s) const {
        if (!host || host == dev || bytes == 0) return hipSuccess;
        return hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s);
    }
    hipError_t download(hipStream_t s) const {
        if (!host || host == dev || bytes == 0) return hipSuccess;
        return hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s);
    }
    bool fits(size_t need) const { return bytes == 0 || need <= bytes; }
};

inline Buffer* as_buf(mfa_buffer_t b) {
    Buffer* x = (Buffer*)b;
    return (x && x->magic == 0x4d464142) ? x : nullptr;
}

inline bool is_device_pointer(const void* p) {
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof(a));
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // unregistered host memory on older runtimes
        return false;
    }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

inline mfa_error_t wrap_pointer(void* ptr, size_t bytes, const int64_t* shape, const int64_t* strides,
                         uint32_t ndim, bool force_device, mfa_buffer_t* out) {
    Buffer* b = new (std::nothrow) Buffer();
    if (!b) return MFA_ERROR_MEMORY_ALLOCATION;
    b->bytes = bytes;
    if (force_device || is_device_pointer(ptr)) {
        b->dev = ptr;  // true zero copy: HBM-resident tensor (torch-ROCm path)
    } else {
        // discrete GPU: a host tensor needs an HBM mirror (Metal's unified memory made this free,
        // MFABridge.swift:892-904); staged around every synchronous op.
        b->host = ptr;
        if (bytes > 0) {
            if (hipMalloc(&b->dev, bytes) != hipSuccess) {
                delete b;
                return MFA_ERROR_MEMORY_ALLOCATION;
            }
            b->owns_dev = true;
        }
    }
    if (shape && strides && ndim) {
        b->shape.assign(shape, shape + ndim);
        b->strides.assign(strides, strides + ndim);
    }
    *out = b;
    return MFA_SUCCESS;
}


// dense path: any precision value other than 0/1 means FP32 (gemmPrecision, MFABridge.swift:1453-1462)
inline int dense_prec(int p) { return p == 0 ? umfa::P_FP16 : p == 1 ? umfa::P_BF16 : umfa::P_FP32; }
inline size_t elem_bytes(int prec) { return prec == umfa::P_FP32 ? 4 : 2; }

}  // namespace umfa_rt
