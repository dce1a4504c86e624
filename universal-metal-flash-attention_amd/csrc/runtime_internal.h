// runtime_internal.h -- host-side objects shared by runtime.hip and runtime_train.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <map>
#include <mutex>
#include <new>
#include <utility>
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

// ---- scratch: grow-only device blocks, one set per (device, stream) -- and per CAPTURE ------------------------------
// The kernels need caller-invisible device scratch (split-item tickets + partials, mask tile flags, the quantiser's
// workspace).  Rules that keep the asynchronous entries correct:
//   * keyed by (device, stream): launches on one stream are ordered, so they may share; two streams (or two host
//     threads on two streams) never touch the same ticket words or partial slots;
//   * keyed by the CAPTURE as well: a call made while its stream is being captured bakes scratch addresses into kernel
//     nodes, and the graph may later be replayed on any stream, concurrently with eager launches on the capture stream
//     or with another graph captured on it (torch.cuda.graph captures everything on one process-wide stream) -- so every
//     capture sequence (hipStreamGetCaptureInfo id) gets a pool of its own, used by nothing else;
//   * grow-only: a block that is too small is RETIRED, not freed -- a launch still in flight or a captured hipGraph
//     may hold its address; retired blocks live until umfa_release_scratch (sizes are bounded by the largest call:
//     growth is geometric, so the retired total stays below the live block);
//   * never allocated while the stream is capturing (hipMalloc invalidates the capture on this runtime, also under
//     hipThreadExchangeStreamCaptureMode(relaxed): tried, round 3): the FIRST call of a capture takes over the stream's
//     eager pool -- the one the caller's warm-up run on that stream filled -- as the capture's private pool; eager calls on
//     the stream afterwards start a fresh pool (allocated outside any capture).  A call under capture whose pool is too
//     small returns MFA_ERROR_MEMORY_ALLOCATION: warm the shape up on the capture stream before EVERY capture, as
//     bench.py and torch's own graph recipe do;
//   * allocated under a device guard for the stream's device; the caller's current device is restored;
//   * never freed behind the caller's back: umfa_release_scratch(context, stream, all) frees pools the caller knows to be
//     idle (no launch in flight, no live graph that was captured with them).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (dev >= 0 && dev != prev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched && prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

inline int stream_device(hipStream_t stream) {
    int dev = 0;
    if (stream) {
        hipDevice_t d;
        if (hipStreamGetDevice(stream, &d) == hipSuccess) return (int)d;
        (void)hipGetLastError();
    }
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    return dev;
}

inline bool stream_capturing(hipStream_t stream) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &st) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return st != hipStreamCaptureStatusNone;
}

// 0 when the stream is not capturing, else a non-zero id unique to the capture sequence
inline unsigned long long capture_id(hipStream_t stream) {
    if (!stream) return 0;  // the legacy default stream cannot be captured
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    if (hipStreamGetCaptureInfo(stream, &st, &id) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return st == hipStreamCaptureStatusActive ? id + 1 : 0;
}

struct GrowBuf {
    void* ptr = nullptr;
    size_t bytes = 0;
    std::vector<void*> retired;
    // >= need bytes, or nullptr (allocation failed / stream is capturing and the block would have to grow)
    void* ensure(size_t need, hipStream_t stream, bool* grew = nullptr) {
        if (grew) *grew = false;
        if (ptr && need <= bytes) return ptr;
        if (stream_capturing(stream)) return nullptr;
        size_t want = need + (need >> 1) + 256;  // geometric growth bounds the retired total
        void* fresh = nullptr;
        if (hipMalloc(&fresh, want) != hipSuccess) {
            (void)hipGetLastError();
            want = need + 256;
            if (hipMalloc(&fresh, want) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        }
        if (ptr) retired.push_back(ptr);
        ptr = fresh;
        bytes = want;
        if (grew) *grew = true;
        return ptr;
    }
    void release() {  // caller: nothing in flight, no live graph holds these addresses
        if (ptr) (void)hipFree(ptr);
        for (void* r : retired) (void)hipFree(r);
        retired.clear();
        ptr = nullptr;
        bytes = 0;
    }
};

struct StreamScratch {
    GrowBuf split;      // fa_fwd16 split-KV: zeroed tickets (the kernel leaves them zero), then partials
    size_t split_cnt_bytes = 0, split_buf_hw = 0;
    GrowBuf w64;        // fa_fwd16_w64: zeroed tickets (the kernel leaves them zero), then partials
    size_t w64_cnt_bytes = 0, w64_buf_hw = 0;
    GrowBuf mflags;     // mask tile flags
    GrowBuf workspace;  // quantiser output (int8 Q/K, V image, scales, fp32 copies for backward); rotated K / Q of the fused-RoPE entry
    GrowBuf v16;        // default bf16 forward: per-slab exchange words + 2^e (512 bytes per (batch, KV head), zeroed once, left zero by the cast pass),
                        // then the fp16 image of V * 2^-e (never the workspace: the RoPE entry's K lives there)
    size_t v16_cnt_bytes = 0, v16_buf_hw = 0;
    GrowBuf rowc;       // bwd16: row constants [2][B*H*Sq] fp32 (-LSE log2 e, -D) from bwd16_dq for bwd16_dkdv
    GrowBuf dsbuf;      // bwd16, option bwd_ds_store: dS [B*H][Sq][Skv] in the operand type
    GrowBuf qhdr;       // runtime-quantised backward: [16 words: the overflow word][16 words: amax of dO, -, -, -, Q, K, V as fp32 bits | BwdParams::units]
                        // at a FIXED address, zeroed once when allocated and left zero by its last reader (fa_aux.hip bwd_units_kernel): no memset node

    // Ticketed scratch: tickets [0, cnt) zeroed on `stream` whenever the block is new -- and never again: the kernels leave
    // their tickets zero (the folding workgroup resets the word it drew from), so a captured graph carries no memset node.
    // Partials sit behind the tickets at `cnt_state`.  (Round 3: the per-launch hipMemsetAsync of fa_fwd16's split path, as a
    // graph node in front of a kernel whose agent-scope atomics bypass the L2, left some tickets non-zero on later replays --
    // tools/lab/value_fuzz.py run_graph_case found it; eager launches were never affected.)
    static char* ensure_ticketed(GrowBuf& g, size_t& cnt_state, size_t& buf_hw, size_t cnt_bytes, size_t buf_bytes, hipStream_t stream) {
        const size_t c = cnt_bytes > cnt_state ? cnt_bytes : cnt_state;
        // the partial area never shrinks: a fresh block (ticket area moved) must still hold the largest partials any earlier
        // call asked for, or a capture of [shape A, shape B] after a warm-up of the same two fails at A (error 2) when B has
        // more tickets and smaller partials than A (the fuzz's graph leg, seed 31954)
        if (buf_bytes > buf_hw) buf_hw = buf_bytes;
        buf_bytes = buf_hw;
        bool grew = false;
        if (c != cnt_state && g.ptr) {  // the ticket area moves: take a fresh block so old launches keep their layout
            if (stream_capturing(stream)) return nullptr;
            g.retired.push_back(g.ptr);
            g.ptr = nullptr;
            g.bytes = 0;
        }
        char* b = (char*)g.ensure(c + buf_bytes, stream, &grew);
        if (!b) return nullptr;
        if (grew) {
            if (hipMemsetAsync(b, 0, c, stream) != hipSuccess) {
                // tickets not zeroed: this block must never be handed out as valid -- retire it, the next call starts over
                (void)hipGetLastError();
                g.retired.push_back(g.ptr);
                g.ptr = nullptr;
                g.bytes = 0;
                return nullptr;
            }
            cnt_state = c;
        }
        return b;
    }
    char* ensure_w64(size_t cnt_bytes, size_t buf_bytes, hipStream_t stream) {
        return ensure_ticketed(w64, w64_cnt_bytes, w64_buf_hw, cnt_bytes, buf_bytes, stream);
    }
    char* ensure_split(size_t cnt_bytes, size_t buf_bytes, hipStream_t stream) {
        return ensure_ticketed(split, split_cnt_bytes, split_buf_hw, cnt_bytes, buf_bytes, stream);
    }
    // header for `slabs` slabs (a 64-KiB multiple: 128 slabs before the header ever moves) + `image_bytes` behind it
    static size_t v16_header_bytes(size_t slabs) { return ((slabs * (umfa::VSC_HDR_WORDS * 4) + 65535) / 65536) * 65536; }
    char* ensure_v16(size_t slabs, size_t image_bytes, hipStream_t stream) {
        return ensure_ticketed(v16, v16_cnt_bytes, v16_buf_hw, v16_header_bytes(slabs), image_bytes + 256, stream);
    }
    // The amax words of the quantised backward are updated with agent-scope atomics (fetch_max) -- the pattern that, behind a per-launch memset NODE
    // of a replayed graph, left ticket words stale in round 3 (see ensure_ticketed).  So they live in a block of their own at an address that does
    // not depend on the call's shape, zeroed when the block is new and cleaned by the launch that reads them last.
    uint32_t* ensure_qhdr(hipStream_t stream) {
        bool grew = false;
        char* b = (char*)qhdr.ensure(512, stream, &grew);
        if (!b) return nullptr;
        if (grew && hipMemsetAsync(b, 0, 512, stream) != hipSuccess) {
            (void)hipGetLastError();
            qhdr.retired.push_back(qhdr.ptr);
            qhdr.ptr = nullptr;
            qhdr.bytes = 0;
            return nullptr;
        }
        return (uint32_t*)b;
    }
    void drop_qhdr() {  // a call failed between the amax launch and bwd_units_kernel: the words may be non-zero -- the next call takes a fresh block
        if (qhdr.ptr) qhdr.retired.push_back(qhdr.ptr);
        qhdr.ptr = nullptr;
        qhdr.bytes = 0;
    }
    void release() {
        split.release(); w64.release(); mflags.release(); workspace.release(); v16.release(); rowc.release(); dsbuf.release(); qhdr.release();
        w64_cnt_bytes = 0;
        split_cnt_bytes = 0;
        v16_cnt_bytes = 0;
        w64_buf_hw = split_buf_hw = v16_buf_hw = 0;
    }
};

// ---- context (MFAContext + GlobalContextStore, MFABridge.swift:91-150,652-687) -------------
struct Context {
    uint32_t magic = 0x4d464143;  // 'MFAC'
    int device = 0;               // device of the synchronous entries (current when the singleton was created)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // the chunked synchronous forward (runtime.hip forward_sync): three non-blocking side streams and an event pair per chunk, made on first use
    hipStream_t side[3] = {nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> chunk_ev;
    double last_latency = 0.0;
    const char* last_kernel = "none";
    void* scratch = nullptr;  // host-mask staging of the synchronous entries (used under mu, then synchronised)
    size_t scratch_bytes = 0;
    std::vector<float> q_scales, k_scales, v_scales;  // mfa_set_scale_arrays: stored, never read
    std::atomic<int> refs{0};
    std::mutex mu;  // guards the pools, last_kernel / latency, and serialises lookup + launch of every entry

    struct PoolKey {
        int dev;
        hipStream_t stream;
        unsigned long long capture;  // 0 = eager launches; else the capture sequence the pool belongs to
        bool operator<(const PoolKey& o) const {
            return dev != o.dev ? dev < o.dev : stream != o.stream ? stream < o.stream : capture < o.capture;
        }
    };
    std::map<PoolKey, StreamScratch> pools;
    // A capture-private pool lives as long as the graph it was captured into (and the executables instantiated from it): the
    // first call of a capture hangs a HIP user object on the graph being built; its destructor -- which may not call HIP --
    // only raises `released`, and the next eager call through this context frees the pool.  (Before: never freed unless the
    // caller said umfa_release_scratch -- a process that keeps re-capturing leaked one pool per capture; the fuzz's graph leg
    // filled 288 GB in 2 240 captures.)  If the runtime refuses the user object the old behaviour remains.
    struct CaptureToken { std::atomic<int> released{0}; };
    std::map<PoolKey, CaptureToken*> tokens;
    static void capture_graph_destroyed(void* tok) { static_cast<CaptureToken*>(tok)->released.store(1, std::memory_order_release); }
    void reap_released_captures() {  // mu held, caller's stream is not capturing
        bool any = false;
        for (auto& kv : tokens) any |= kv.second->released.load(std::memory_order_acquire) != 0;
        if (!any) return;
        // hipFree is a "potentially unsafe" call while ANOTHER thread captures in global mode (torch's default): it would fail
        // and invalidate that capture.  This thread is not capturing: the relaxed mode lifts the prohibition for it.
        hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
        const bool swapped = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess;
        struct Restore {
            bool on; hipStreamCaptureMode m;
            ~Restore() { if (on) (void)hipThreadExchangeStreamCaptureMode(&m); }
        } restore{swapped, mode};
        for (auto it = tokens.begin(); it != tokens.end();) {
            if (!it->second->released.load(std::memory_order_acquire)) { ++it; continue; }
            auto pit = pools.find(it->first);
            if (pit != pools.end()) {
                DeviceGuard g(it->first.dev);
                pit->second.release();
                pools.erase(pit);
            }
            delete it->second;
            it = tokens.erase(it);
        }
    }
    // call with mu held; the returned object is stable (std::map nodes never move)
    StreamScratch& pool(int dev, hipStream_t stream) {
        const unsigned long long cap = capture_id(stream);
        if (!cap) {
            if (!tokens.empty()) reap_released_captures();
            return pools[PoolKey{dev, stream, 0}];
        }
        const PoolKey ck{dev, stream, cap};
        auto it = pools.find(ck);
        if (it == pools.end()) {
            // first call of this capture: the stream's eager pool (filled by the caller's warm-up) becomes the capture's own
            it = pools.emplace(ck, StreamScratch()).first;
            auto eager = pools.find(PoolKey{dev, stream, 0});
            if (eager != pools.end()) {
                it->second = std::move(eager->second);
                pools.erase(eager);
            }
            hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
            unsigned long long id = 0;
            hipGraph_t graph = nullptr;
            if (hipStreamGetCaptureInfo_v2(stream, &st, &id, &graph, nullptr, nullptr) == hipSuccess && graph) {
                CaptureToken* tok = new CaptureToken;
                hipUserObject_t obj = nullptr;
                if (hipUserObjectCreate(&obj, tok, capture_graph_destroyed, 1, hipUserObjectNoDestructorSync) == hipSuccess) {
                    if (hipGraphRetainUserObject(graph, obj, 1, hipGraphUserObjectMove) == hipSuccess) {
                        tokens[ck] = tok;
                    } else {
                        (void)hipGetLastError();
                        (void)hipUserObjectRelease(obj, 1);  // its destructor still owns tok's flag: a few bytes stay behind
                    }
                } else {
                    (void)hipGetLastError();
                    delete tok;
                }
            } else {
                (void)hipGetLastError();
            }
        }
        return it->second;
    }
    // frees the pools of `stream` (eager and capture-private), or every pool; call with mu held, devices idle
    size_t release_pools(hipStream_t stream, bool all) {
        size_t n = 0;
        for (auto it = pools.begin(); it != pools.end();) {
            if (all || it->first.stream == stream) {
                DeviceGuard g(it->first.dev);
                it->second.release();
                it = pools.erase(it);
                ++n;
            } else {
                ++it;
            }
        }
        // (tokens of pools released here stay until their graphs go: reap_released_captures then finds no pool and drops them)
        return n;
    }

    void* ensure_scratch(size_t bytes) {
        if (bytes <= scratch_bytes) return scratch;
        if (scratch) (void)hipFree(scratch);  // synchronous entries only: nothing is in flight between calls
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
    bool cached_mirror = false;  // `dev` is an HBM mirror made by wrap_pointer: goes back to mirror_cache() on destroy
    int mirror_dev = 0;
    bool registered = false;  // wrap_pointer pinned the caller's host range (hipHostRegister): copies on it are asynchronous
    std::vector<int64_t> shape, strides;

    hipError_t upload(hipStream_t s) const {
        if (!host || host == dev || bytes == 0) return hipSuccess;
        return hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s);
    }
    hipError_t download(hipStream_t s) const {
        if (!host || host == dev || bytes == 0) return hipSuccess;
        return hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s);
    }
    bool mirrored() const { return host && dev && host != dev && bytes > 0; }
    // a byte range of the mirror (the chunked synchronous forward)
    hipError_t upload_range(size_t off, size_t n, hipStream_t s) const {
        return hipMemcpyAsync((char*)dev + off, (const char*)host + off, n, hipMemcpyHostToDevice, s);
    }
    hipError_t download_range(size_t off, size_t n, hipStream_t s) const {
        return hipMemcpyAsync((char*)host + off, (const char*)dev + off, n, hipMemcpyDeviceToHost, s);
    }
    // bytes == 0: size unknown (device wraps only); a host pointer of unknown size has no HBM mirror (dev == NULL)
    bool fits(size_t need) const { return dev != nullptr && (bytes == 0 || need <= bytes); }
};

// HBM mirrors of host-wrapping buffers, kept between wrappers: a caller that wraps its arrays per call (the reference's Python binding does:
// examples/python-ffi/src/umfa/core.py) paid a hipMalloc per array and -- the expensive half, ~0.7 ms each at the FLUX shape: it waits for the device and
// unmaps -- a hipFree per array on every call (profiles/r6/wrapper_cost_probe.jsonl).  mfa_destroy_buffer hands the block here, the next wrap of the
// same size on the same device takes it.  Bounded (32 blocks, 4 GiB); umfa_release_scratch(all) empties it.  A recycled mirror holds old bytes
// where a fresh one held arbitrary ones: every synchronous entry uploads what it reads and downloads only what its kernels wrote or the caller's own bytes.
struct MirrorCache {
    struct Blk {
        void* p;
        size_t bytes;
        int dev;
    };
    std::mutex mu;
    std::vector<Blk> blocks;
    size_t held = 0;
    void* take(size_t bytes, int dev) {
        std::lock_guard<std::mutex> lock(mu);
        for (size_t i = 0; i < blocks.size(); ++i)
            if (blocks[i].bytes == bytes && blocks[i].dev == dev) {
                void* p = blocks[i].p;
                umfa::tuning().mirror_cache_hits.fetch_add(1, std::memory_order_relaxed);
                held -= bytes;
                blocks.erase(blocks.begin() + (long)i);
                return p;
            }
        return nullptr;
    }
    bool give(void* p, size_t bytes, int dev) {
        std::lock_guard<std::mutex> lock(mu);
        if (blocks.size() >= 32 || held + bytes > ((size_t)4 << 30)) return false;
        blocks.push_back({p, bytes, dev});
        held += bytes;
        return true;
    }
    void clear() {
        std::lock_guard<std::mutex> lock(mu);
        for (auto& b : blocks) (void)hipFree(b.p);
        blocks.clear();
        held = 0;
    }
};
inline MirrorCache& mirror_cache() {
    static MirrorCache* c = new MirrorCache();  // (never destroyed: no HIP calls from static destructors)
    return *c;
}

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
            if (hipGetDevice(&b->mirror_dev) != hipSuccess) b->mirror_dev = 0;
            b->dev = mirror_cache().take(bytes, b->mirror_dev);
            if (!b->dev && hipMalloc(&b->dev, bytes) != hipSuccess) {
                mirror_cache().clear();  // (memory held for the next wrap must not fail this one)
                if (hipMalloc(&b->dev, bytes) != hipSuccess) {
                    delete b;
                    return MFA_ERROR_MEMORY_ALLOCATION;
                }
            }
            b->owns_dev = true;
            b->cached_mirror = true;
        }
    }
    if (shape && strides && ndim) {
        b->shape.assign(shape, shape + ndim);
        b->strides.assign(strides, strides + ndim);
    }
    *out = b;
    return MFA_SUCCESS;
}


// ---- the synchronous entries on host-wrapping buffers, in head chunks (runtime.hip forward_sync_chunked, runtime_train.hip) ----
// A chunk is either whole batches or a head range of one batch: slabs [slab0, slab0 + nslab) of the dense [B, H, ., .] tensors are contiguous,
// and the kernels' dense indexing ((b H + h) S) holds with the chunk's own B and H.
struct SyncChunk {
    uint32_t b0, nb, h0, nh;
};
inline std::vector<SyncChunk> plan_sync_chunks(uint32_t B, uint32_t H, uint32_t want) {
    std::vector<SyncChunk> chunks;
    if (B >= want) {
        const uint32_t per = (B + want - 1) / want;
        for (uint32_t b = 0; b < B; b += per) chunks.push_back({b, std::min(per, B - b), 0, H});
    } else {
        const uint32_t pieces = std::min(H, (want + B - 1) / B), per = (H + pieces - 1) / pieces;
        for (uint32_t b = 0; b < B; ++b)
            for (uint32_t h = 0; h < H; h += per) chunks.push_back({b, 1, h, std::min(per, H - h)});
    }
    return chunks;
}
// the option's chunk count for a call that moves `moved` bytes over the host link: 0 = by size -- about 14 MB per chunk, at most 8, from 64 MB
// on (profiles/r6/host_boundary_probe.jsonl: at 42 MB the chunks' launches, copy set-ups and the pinning cost what the overlap returns); 1: one upload
inline int sync_chunk_count(size_t moved) {
    int want = umfa::tuning().sync_chunks.load(std::memory_order_relaxed);
    if (want == 0) want = moved >= (64u << 20) ? (int)std::min<size_t>(8, moved / (14u << 20)) : 1;
    return moved >= (16u << 20) ? std::min(want, 16) : 1;
}
// side streams + an event pair per chunk; ctx->mu held.  The side streams then wait for whatever the null stream holds.
inline bool sync_chunks_begin(Context* ctx, size_t nchunks) {
    for (auto& s : ctx->side)
        if (!s && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return false;
    while (ctx->chunk_ev.size() < 2 * nchunks) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return false;
        ctx->chunk_ev.push_back(e);
    }
    umfa::tuning().sync_chunked_calls.fetch_add(1, std::memory_order_relaxed);
    (void)hipEventRecord(ctx->ev0, nullptr);
    for (auto s : ctx->side) (void)hipStreamWaitEvent(s, ctx->ev0, 0);
    return true;
}
// drains the side streams; kernel-only GPU time of the chunks' launches (not of the copies they ran under) -> mfa_get_gpu_latency
inline hipError_t sync_chunks_end(Context* ctx, size_t nchunks, bool ok) {
    hipError_t e = hipSuccess;
    for (auto s : ctx->side) {
        const hipError_t e2 = hipStreamSynchronize(s);
        if (e == hipSuccess) e = e2;
    }
    if (!ok || e != hipSuccess) return e;
    double sum = 0.0;
    for (size_t c = 0; c < nchunks; ++c) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->chunk_ev[2 * c], ctx->chunk_ev[2 * c + 1]) == hipSuccess) sum += ms * 1e-3;
    }
    ctx->last_latency = sum;
    return hipSuccess;
}
// A chunked call needs every host range pinned (copies on pinned memory are asynchronous: that is what lets one chunk's download run under another's
// upload).  Pinned here, by the first call that wants chunks -- not at wrap time: a wrapper that only ever serves small calls costs what it did --
// and for as long as the wrapper lives (mfa_destroy_buffer releases it).  Ranges under 1 MiB (they may share pages with other allocations of the
// caller's) and ranges the runtime refuses (registered by the caller already, ...) stay pageable, and the call takes the one-upload form.
inline bool pin_for_chunks(std::initializer_list<Buffer*> bs) {
    // (1 MiB: a lab build that pinned the sub-page arrays of the fuzz's host leg ran 6000 seeds of that leg clean and then died in the all-legs soak --
    // "write access to a read-only page" on a host heap address, profiles/r6/lab_notes.md section 24: small heap ranges share their pages with the
    // process's other objects.  Not offered as an option.)
    for (Buffer* b : bs)
        if (!b->mirrored() || (!b->registered && b->bytes < (1u << 20))) return false;
    for (Buffer* b : bs) {
        if (b->registered) continue;
        if (hipHostRegister(b->host, b->bytes, hipHostRegisterDefault) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        b->registered = true;
    }
    return true;
}

// dense path: any precision value other than 0/1 means FP32 (gemmPrecision, MFABridge.swift:1453-1462)
inline int dense_prec(int p) { return p == 0 ? umfa::P_FP16 : p == 1 ? umfa::P_BF16 : umfa::P_FP32; }
inline size_t elem_bytes(int prec) { return prec == umfa::P_FP32 ? 4 : 2; }

// Normalise a <= 4-D mask onto (b, h, q, k) element strides (runtime.hip; mfa_prepare_mask, MFABridge.swift:157-242); false: no mask
bool normalise_mask(const int64_t* shape, const int64_t* strides, uint32_t ndim, int type, int scalar, umfa::FwdParams& p);

}  // namespace umfa_rt
