// tuning.hip -- process-wide switches of the kernel launchers (kernels.h: Tuning).
//
// The reference has no such switches (its kernel choice is fixed inside the absent submodule); these exist so that tests
// and benches can drive every kernel variant and numerics regime of this library in ONE process.  They used to be
// getenv() calls on the launch path: a numerics switch that another thread's setenv can flip mid-launch is not an API,
// and getenv is not thread-safe against setenv.  Now: the environment gives the INITIAL value, once, when the table is
// first touched; afterwards only umfa_set_option (include/umfa_abi.h) changes a switch, and a launch reads each one once
// (relaxed atomics: a switch is a hint to the NEXT launch, not a synchronisation point).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "kernels.h"

namespace umfa {

namespace {
int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}
bool env_flag(const char* name) {
    const char* e = getenv(name);
    return e && e[0] == '1';
}
}  // namespace

Tuning& tuning() {
    static Tuning* t = [] {
        Tuning* x = new Tuning();
        // softmax reference of the 64-rows-per-wave forward kernels
        const char* tau = getenv("UMFA_W64_TAU");
        const char* lazy = getenv("UMFA_W64_LAZY");
        if (tau && *tau) {
            const float v = (float)atof(tau);
            if (v >= 0.0f && v <= 16.0f) {
                x->sm_tau.store(v);
                x->sm_mode.store(v == 0.0f ? SM_EXACT : SM_DEFERRED);
            }
        }
        if (lazy && *lazy) x->sm_mode.store(lazy[0] != '0' ? SM_LAZY : (x->sm_mode.load() == SM_DEFAULT ? SM_DEFERRED : x->sm_mode.load()));
        x->force_w64.store(env_flag("UMFA_FORCE_W64"));
        x->no_w64.store(env_flag("UMFA_NO_W64"));
        x->w64_grid.store(env_int("UMFA_W64_GRID", 0));
        x->w64_skew.store(env_int("UMFA_W64_SKEW", 0));
        x->no_mask_flags.store(getenv("UMFA_NO_MASK_FLAGS") != nullptr);
        x->bwd_exact.store(getenv("UMFA_BWD_EXACT") != nullptr);
        x->bwd_dq.store(env_int("UMFA_BWD_DQ", 0));
        x->bwd_persist.store(getenv("UMFA_BWD_PERSIST") != nullptr);
        x->bwd_separate_delta.store(getenv("UMFA_LAB_SEPARATE_DELTA") != nullptr);
        x->no_split.store(env_flag("UMFA_NO_SPLIT"));
        x->force_split.store(env_int("UMFA_FORCE_SPLIT", 0));
        x->no_dma.store(env_flag("UMFA_NO_DMA"));
        x->bn64.store(env_flag("UMFA_BN64"));
        x->pv_fp16.store(env_int("UMFA_PV_FP16", 1) != 0);  // bf16 forward: P V in fp16 (inside 1e-3) unless switched off
        x->bwd_ds_store.store(env_flag("UMFA_BWD_DS_STORE"));
        x->no_w64_mask.store(env_flag("UMFA_NO_W64_MASK"));
        x->ksplit.store(env_flag("UMFA_KSPLIT"));
        x->no_pipe.store(env_flag("UMFA_NO_PIPE"));
        x->no_w64_mask_lazy.store(env_flag("UMFA_NO_W64_MASK_LAZY"));
        x->no_w64_bias.store(env_flag("UMFA_NO_W64_BIAS"));
        x->no_w64_f32_mask.store(env_flag("UMFA_NO_W64_F32_MASK"));
        x->cast_two_pass.store(env_flag("UMFA_CAST_TWO_PASS"));
        x->cast_u.store(env_int("UMFA_CAST_U", 0));
        x->quant_block_wg.store(env_flag("UMFA_QUANT_BLOCK_WG"));
        x->cast_wait_us.store(env_int("UMFA_CAST_WAIT_US", 100));
        x->bwd_ds_lab.store(env_int("UMFA_LAB_DS", 0));
        x->cbal.store(env_int("UMFA_CBAL", 0));
        x->decode_ks.store(env_int("UMFA_DECODE_KS", 0));
        x->cbal_delta.store(env_int("UMFA_CBAL_DELTA", -1));
        x->sync_chunks.store(env_int("UMFA_SYNC_CHUNKS", 1));
        return x;
    }();
    return *t;
}

// name: the environment variable's name without the UMFA_ prefix, lower case ("force_w64", "w64_tau", ...)
bool set_tuning(const char* name, const char* value) {
    if (!name || !value) return false;
    Tuning& t = tuning();
    const int iv = atoi(value);
    const bool on = value[0] != '\0' && value[0] != '0';
    if (!strcmp(name, "softmax_reference")) {
        const int m = !strcmp(value, "default") ? SM_DEFAULT : !strcmp(value, "exact") ? SM_EXACT
                      : !strcmp(value, "deferred") ? SM_DEFERRED : !strcmp(value, "lazy") ? SM_LAZY : -1;
        if (m < 0) return false;
        t.sm_mode.store(m);
        return true;
    }
    if (!strcmp(name, "softmax_tau") || !strcmp(name, "w64_tau")) {
        const float v = (float)atof(value);
        if (!(v >= 0.0f && v <= 16.0f)) return false;
        t.sm_tau.store(v);
        if (!strcmp(name, "w64_tau")) t.sm_mode.store(v == 0.0f ? SM_EXACT : SM_DEFERRED);  // the old variable's meaning
        return true;
    }
    struct { const char* n; std::atomic<int>* v; bool flag; } tab[] = {
        {"force_w64", &t.force_w64, true}, {"no_w64", &t.no_w64, true}, {"w64_grid", &t.w64_grid, false}, {"w64_skew", &t.w64_skew, false},
        {"no_mask_flags", &t.no_mask_flags, true}, {"bwd_exact", &t.bwd_exact, true}, {"bwd_dq", &t.bwd_dq, false},
        {"bwd_persist", &t.bwd_persist, true}, {"bwd_separate_delta", &t.bwd_separate_delta, true},
        {"no_split", &t.no_split, true}, {"force_split", &t.force_split, false}, {"no_dma", &t.no_dma, true},
        {"bn64", &t.bn64, true}, {"pv_fp16", &t.pv_fp16, true}, {"bwd_ds_store", &t.bwd_ds_store, true}, {"no_w64_mask", &t.no_w64_mask, true}, {"ksplit", &t.ksplit, true}, {"no_pipe", &t.no_pipe, true}, {"no_w64_mask_lazy", &t.no_w64_mask_lazy, true}, {"no_w64_bias", &t.no_w64_bias, true}, {"no_w64_f32_mask", &t.no_w64_f32_mask, true}, {"f32_mask_ratio", &t.f32_mask_ratio, false}, {"mask_pass_ratio", &t.mask_pass_ratio, false}, {"no_w64_ragged_mask", &t.no_w64_ragged_mask, true}, {"no_mask_realign", &t.no_mask_realign, true},
        {"cast_two_pass", &t.cast_two_pass, true}, {"bwd_ds_lab", &t.bwd_ds_lab, false}, {"cast_u", &t.cast_u, false}, {"quant_block_wg", &t.quant_block_wg, true},
        {"cast_wait_us", &t.cast_wait_us, false}, {"cbal", &t.cbal, false}, {"cbal_delta", &t.cbal_delta, false}, {"decode_ks", &t.decode_ks, false}, {"sync_chunks", &t.sync_chunks, false}, {"sync_chunked_calls", &t.sync_chunked_calls, false}, {"mirror_cache_hits", &t.mirror_cache_hits, false},
    };
    for (auto& e : tab)
        if (!strcmp(name, e.n)) {
            e.v->store(e.flag ? (on ? 1 : 0) : iv);
            return true;
        }
    return false;
}

// the live value of a switch as text (what set_tuning would take); false: unknown name or buffer too small
bool get_tuning(const char* name, char* out, size_t n) {
    if (!name || !out || n < 2) return false;
    Tuning& t = tuning();
    char buf[32];
    if (!strcmp(name, "softmax_reference")) {
        const int m = t.sm_mode.load();
        snprintf(buf, sizeof(buf), "%s", m == SM_EXACT ? "exact" : m == SM_DEFERRED ? "deferred" : m == SM_LAZY ? "lazy" : "default");
    } else if (!strcmp(name, "softmax_tau") || !strcmp(name, "w64_tau")) {
        snprintf(buf, sizeof(buf), "%g", (double)t.sm_tau.load());
    } else {
        struct { const char* n; std::atomic<int>* v; } tab[] = {
            {"force_w64", &t.force_w64}, {"no_w64", &t.no_w64}, {"w64_grid", &t.w64_grid}, {"w64_skew", &t.w64_skew},
            {"no_mask_flags", &t.no_mask_flags}, {"bwd_exact", &t.bwd_exact}, {"bwd_dq", &t.bwd_dq}, {"bwd_persist", &t.bwd_persist},
            {"bwd_separate_delta", &t.bwd_separate_delta}, {"no_split", &t.no_split}, {"force_split", &t.force_split},
            {"no_dma", &t.no_dma}, {"bn64", &t.bn64}, {"pv_fp16", &t.pv_fp16}, {"bwd_ds_store", &t.bwd_ds_store}, {"no_w64_mask", &t.no_w64_mask}, {"ksplit", &t.ksplit}, {"no_pipe", &t.no_pipe}, {"no_w64_mask_lazy", &t.no_w64_mask_lazy}, {"no_w64_bias", &t.no_w64_bias}, {"no_w64_f32_mask", &t.no_w64_f32_mask}, {"f32_mask_ratio", &t.f32_mask_ratio}, {"mask_pass_ratio", &t.mask_pass_ratio}, {"no_w64_ragged_mask", &t.no_w64_ragged_mask}, {"no_mask_realign", &t.no_mask_realign},
            {"cast_two_pass", &t.cast_two_pass}, {"bwd_ds_lab", &t.bwd_ds_lab}, {"cast_u", &t.cast_u}, {"quant_block_wg", &t.quant_block_wg},
            {"cast_wait_us", &t.cast_wait_us}, {"cbal", &t.cbal}, {"cbal_delta", &t.cbal_delta}, {"decode_ks", &t.decode_ks}, {"sync_chunks", &t.sync_chunks}, {"sync_chunked_calls", &t.sync_chunked_calls}, {"mirror_cache_hits", &t.mirror_cache_hits},
        };
        const std::atomic<int>* v = nullptr;
        for (auto& e : tab)
            if (!strcmp(name, e.n)) v = e.v;
        if (!v) return false;
        snprintf(buf, sizeof(buf), "%d", v->load());
    }
    if (strlen(buf) + 1 > n) return false;
    strcpy(out, buf);
    return true;
}

// ---- per-DEVICE launch state.  The in-stream entries launch on streams of any device (DeviceGuard), so nothing about a
// device may be cached per process: the CU count sizes persistent grids and ticket plans, and the dynamic-LDS attribute
// of a kernel is a property of the loaded code object of ONE device.
int device_cu_count() {
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int n = cus[dev].load(std::memory_order_relaxed);
    if (n <= 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

hipError_t ensure_dynamic_lds(const void* kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return hipSuccess;
    struct Entry { const void* k; int dev; size_t bytes; };
    static std::mutex mu;
    static std::vector<Entry> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::lock_guard<std::mutex> lock(mu);
    for (const Entry& e : done)
        if (e.k == kernel && e.dev == dev && e.bytes >= bytes) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    done.push_back({kernel, dev, bytes});
    return hipSuccess;
}

}  // namespace umfa
