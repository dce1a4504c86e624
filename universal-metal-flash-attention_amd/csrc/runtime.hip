// runtime.hip -- host side of libMFAFFI.so on ROCm: the C ABI of include/umfa_abi.h.
//
// Replaces the reference's Swift bridge (Sources/MFABridge/*.swift): handle lifetime,
// buffer wrapping, mask normalisation, kernel selection, launch and timing.  There is no
// CPU fallback: every compute entry point needs a live gfx950 context.
#include <string>

#include "runtime_internal.h"

using namespace umfa;
using namespace umfa_rt;

namespace umfa_rt {
const bool g_debug = [] {
    const char* e = getenv("MFA_DEBUG");
    return e && e[0] == '1';
}();
std::mutex g_ctx_mu;
Context* g_ctx = nullptr;

// Normalise a <=4-D mask onto (b, h, q, k) strides; size-1 dims broadcast (MFABridge.swift:186-198).
bool normalise_mask(const int64_t* shape, const int64_t* strides, uint32_t ndim, int type, int scalar,
                    FwdParams& p) {
    p.mask_kind = MK_NONE;
    for (int i = 0; i < 4; ++i) p.ms[i] = 0;
    if (type == MFA_MASK_TYPE_NONE || ndim == 0 || ndim > 4 || !shape || !strides) return false;
    for (uint32_t i = 0; i < ndim; ++i) {
        const int coord = 4 - (int)ndim + (int)i;
        p.ms[coord] = shape[i] == 1 ? 0 : strides[i];
    }
    if (type == MFA_MASK_TYPE_BOOL) p.mask_kind = MK_BOOL;
    else if (type == MFA_MASK_TYPE_ADDITIVE) {
        p.mask_kind = scalar == MFA_MASK_SCALAR_FP32 ? MK_F32
                      : scalar == MFA_MASK_SCALAR_FP16 ? MK_F16
                      : scalar == MFA_MASK_SCALAR_BF16 ? MK_BF16
                                                       : MK_NONE;  // additive bytes: 0.0 (MFABridge.swift:227-229)
    }
    return p.mask_kind != MK_NONE;
}
}  // namespace umfa_rt

namespace {

int parse_precision(const char* s) {  // MFABridge.swift:1438-1451
    if (!s) return MFA_PRECISION_FP32;
    std::string t(s);
    for (auto& ch : t) ch = (char)tolower(ch);
    if (t == "fp16" || t == "float16") return MFA_PRECISION_FP16;
    if (t == "bf16" || t == "bfloat16") return MFA_PRECISION_BF16;
    if (t == "fp32" || t == "float32") return MFA_PRECISION_FP32;
    if (t == "int8") return MFA_PRECISION_INT8;
    if (t == "int4") return MFA_PRECISION_INT4;
    return MFA_PRECISION_FP32;
}


void dense_strides(FwdParams& p, bool tq, bool tk, bool tv, bool to) {
    // row-major per head [S, D]; "transposed" = per-head [D, S] storage (mfa_ffi.h:266-269)
    auto set = [](int64_t* s, uint32_t H, uint32_t S, uint32_t D, bool t) {
        s[0] = (int64_t)H * S * D;
        s[1] = (int64_t)S * D;
        s[2] = t ? 1 : D;
        s[3] = t ? S : 1;
    };
    set(p.qs, p.H, p.Sq, p.D, tq);
    set(p.ks, p.H, p.Skv, p.D, tk);
    set(p.vs, p.H, p.Skv, p.D, tv);
    p.os[0] = to ? 1 : p.D;
    p.os[1] = to ? p.Sq : 1;
}

// hipError_t of a launcher -> mfa_error_t (hipErrorOutOfMemory: scratch could not be provided, e.g. it would have to
// grow while the stream is capturing)
mfa_error_t rc_of(hipError_t e) {
    return e == hipSuccess ? MFA_SUCCESS
           : e == hipErrorInvalidValue ? MFA_ERROR_INVALID_ARGS
           : e == hipErrorOutOfMemory ? MFA_ERROR_MEMORY_ALLOCATION
                                      : MFA_ERROR_EXECUTION_FAILED;
}

// the name of a call that enqueued two guarded routes (an fp32 mask: the device picks); ctx->mu held, the strings live as long as the library
const char* both_kernels(const char* first, const char* second) {
    static std::map<std::pair<const char*, const char*>, std::string> names;
    std::string& s = names[std::make_pair(first, second)];
    if (s.empty()) s = std::string(first) + " | " + second + " (fp32 mask: chosen on the device)";
    return s.c_str();
}

// Kernel selection for the dense forward.  Call with ctx->mu held; `sc` is the scratch pool of (device, stream).
hipError_t dispatch_forward(Context* ctx, StreamScratch& sc, const FwdParams& p, int intermediate_prec, hipStream_t stream) {
    const char* name = "none";
    hipError_t e = hipSuccess;
    const bool lowp = p.in_prec != P_FP32 && intermediate_prec != P_FP32;
    // bf16 operands: the P V product runs in fp16 by default (FwdParams::pv16: P rounded to fp16, V as an fp16 image shifted by one power
    // of two per (batch, head) slab) -- the bf16-input forward inside the north-star's 1e-3 for EVERY bf16 input: the shift is chosen from
    // the data on the device (cast pre-pass / the converting kernel itself), so there is no status word, no second call and no state.
    FwdParams pv = p;
    if (lowp && p.in_prec == P_BF16 && tuning().pv_fp16.load(std::memory_order_relaxed)) pv.pv16 = 1;
    // the fp16 image of V for the kernels that take one: one HBM-speed cast pass per call into a block of its own (the workspace may hold
    // the rotated K / Q of the fused-RoPE entry).  A broadcast batch / head dimension of V (the zero-copy GQA views: stride 0) stays one:
    // its slab is cast once.  false with e == hipSuccess: no block (the pool may not grow while the stream is capturing) -- the caller
    // takes a kernel that converts V itself.
    // mask_scratch != NULL: the bool mask of `q` is re-packed for the one-wave-per-SIMD kernel in the same launch (fills q.mk_*)
    // defer != NULL: nothing is launched here -- *defer receives the pass's arguments and the caller hands them to a launch the pass rides in (additive masks:
    // the classification pass, fa_aux.hip launch_mask_classify)
    auto cast_v = [&](FwdParams& q, void* mask_scratch = nullptr, CastRowsCall* defer = nullptr) -> bool {
        const uint32_t vB = p.vs[0] == 0 ? 1u : p.B, vH = p.vs[1] == 0 ? 1u : p.H;
        const size_t slabs = (size_t)vB * vH, vbytes = slabs * p.Skv * p.D * 2;
        char* blk = sc.ensure_v16(slabs, vbytes, stream);
        if (!blk) return false;
        void* v16 = blk + sc.v16_cnt_bytes;
        if (defer) {
            *defer = CastRowsCall{p.v, {p.vs[0], p.vs[1], p.vs[2], p.vs[3]}, v16, vB, vH, p.Skv, p.D, (uint32_t*)blk};
            e = hipSuccess;
        } else {
            e = mask_scratch ? launch_cast_rows_and_mask_pack(p.v, p.vs, v16, vB, vH, p.Skv, p.D, (uint32_t*)blk, q, mask_scratch, stream)
                             : launch_cast_rows_bf16_to_f16(p.v, p.vs, v16, vB, vH, p.Skv, p.D, (uint32_t*)blk, stream);
        }
        if (e != hipSuccess) return false;
        q.v = v16;
        q.vs[0] = p.vs[0] == 0 ? 0 : (int64_t)vH * p.Skv * p.D; q.vs[1] = p.vs[1] == 0 ? 0 : (int64_t)p.Skv * p.D; q.vs[2] = p.D; q.vs[3] = 1;
        q.vsc = (const float*)blk;
        q.vsc_bs = p.vs[0] == 0 ? 0u : vH; q.vsc_hs = p.vs[1] == 0 ? 0u : 1u;
        q.pv16 = 2;
        return true;
    };
    bool done = false;
    // fp32 additive masks the bias kernels could take as fp16 (fwd_w64_supported): whether the fp16 copy is exact is known on the device only, so BOTH routes
    // are enqueued and guarded by the classification pass's verdict word (FwdParams::guard) -- first the bias kernel on the copy, then, below, the 128-row
    // kernel on the caller's tensor; exactly one of them runs
    const uint32_t* guard = nullptr;
    const char* guarded_first = nullptr;
    FwdParams v16_from;  // ... and the second route takes the fp16 image of V the first one's cast pass wrote
    bool have_v16 = false;
    if (lowp && fwd_w64_supported(pv)) {
        // the one-wave-per-SIMD kernels take V as the dense fp16 image.  (Converting inside these kernels was built and measured: +15 %,
        // every workgroup re-converts every tile; the 128-row kernel below does convert in-kernel.)
        FwdParams pw = pv;
        // bool mask tensor on the one-wave-per-SIMD structure: one pre-pass re-packs it into per-lane bit words, per-wave tile classes
        // and the visited-tile list of every 256-row block (fa_aux.hip mask_pack_kernel); the kernel then never stages a tile no row
        // of the block attends to and reads no mask bytes at all.  Every block this family needs is asked for BEFORE anything is
        // launched: a call that cannot have them (capture without a warm-up) goes to the 128-row kernel untouched.
        const bool mask_f32 = pw.mask_kind == MK_F32;
        const bool mask_add = pw.mask_kind == MK_F16 || pw.mask_kind == MK_BF16 || mask_f32;  // additive: classified only (+ a bf16 / fp32 mask's fp16 copy) -- fa_aux.hip launch_mask_classify
        const bool mask_w64 = pw.mask_kind == MK_BOOL || mask_add;
        void* mk = mask_w64 ? sc.mflags.ensure(((mask_pack_bytes(pw) + 255) & ~(size_t)255) + mask_copy_bytes(pw), stream) : nullptr;
        bool ok = !mask_w64 || mk != nullptr;
        const FwdW64Plan plan = fwd_w64_plan(pw);
        char* w64 = ok ? sc.ensure_w64(plan.cnt_bytes, plan.buf_bytes, stream) : nullptr;
        ok = ok && w64 != nullptr;
        bool packed = false;
        CastRowsCall cast_call;
        const bool cast_rides = mask_add && mk != nullptr;  // (additive masks: the cast pass rides in the classification's launch)
        bool cast_deferred = false;
        if (ok && pv.pv16) {
            ok = cast_v(pw, pw.mask_kind == MK_BOOL ? mk : nullptr, cast_rides ? &cast_call : nullptr);  // (with a bool mask: the re-pack rides in the cast's launch)
            if (!ok && e != hipSuccess) return e;
            packed = ok && mk != nullptr && pw.mask_kind == MK_BOOL;
            cast_deferred = ok && cast_rides;
        }
        if (ok) {
            if (mk && !packed && (e = (mask_add ? launch_mask_classify(pw, mk, stream, cast_deferred ? &cast_call : nullptr) : launch_mask_pack(pw, mk, stream))) != hipSuccess) return e;
            e = launch_fwd_w64(pw, (float*)(w64 + sc.w64_cnt_bytes), (uint32_t*)w64, stream, &name);
            if (mask_f32) {
                if (e != hipSuccess) return e;
                guard = pw.guard;
                guarded_first = name;
                if (pw.pv16 == 2) { v16_from = pw; have_v16 = true; }
            } else {
                done = true;
            }
        } else if (p.rope_cos || !fwd_16_supported(p)) {
            return hipErrorOutOfMemory;  // (only this kernel family rotates Q in registers)
        }
        // else: a scratch block this family needs could not be provided (the first capture of a shape without an eager warm-up): the
        // 128-row kernel below runs the call with what it can get -- V converted in the kernel, masks read per score
    }
    if (done) {
    } else if (p.rope_cos) {
        return hipErrorNotSupported;  // only the 256-row kernel rotates Q in registers (the entry asks before it sets this)
    } else if (lowp && fwd_16_supported(p)) {
        FwdParams pp = pv;
        if (guard) { pp.guard = guard; pp.guard_want = 1; }
        if (have_v16) {
            pp.v = v16_from.v; pp.vsc = v16_from.vsc; pp.vsc_bs = v16_from.vsc_bs; pp.vsc_hs = v16_from.vsc_hs; pp.pv16 = 2;
            for (int i = 0; i < 4; ++i) pp.vs[i] = v16_from.vs[i];
        } else if (pp.pv16 && (size_t)(p.vs[0] == 0 ? 1u : p.B) * (p.vs[1] == 0 ? 1u : p.H) * p.Skv * p.D * 2 >= ((size_t)16 << 20) && p.Sq >= 1024) {
            // the 128-row kernel converts V in-kernel (24 ... 48 vector instructions per tile per wave in a vector-bound kernel, repeated
            // by every workgroup): right for short launches, where a pre-pass costs its launch; from 16 MB of V on the HBM-speed cast
            // pass is cheaper (FLUX-size masked calls: ~11 us against ~15 % of the kernel) -- IF the tiles are re-read: with fewer than eight
            // 128-row q-blocks per head (decode-like calls: K / V are swept once) two more passes over V cost more than the kernel's own
            // sweep (B8 H32 Sq1 Skv8192: 393 us with the pass, see profiles/r4/lab_notes.md section 6).  Without a block (capture): in-kernel.
            if (!cast_v(pp) && e != hipSuccess) return e;
        }
        const FwdSplitPlan plan = fwd_16_split_plan(p);
        pp.decode_form = plan.decode;
        if (plan.nsplit > 1) {
            // tickets first (16-byte multiple at the allocation start), partials behind them
            char* buf = sc.ensure_split(plan.cnt_bytes, plan.buf_bytes, stream);  // tickets zero: at allocation, then by the kernel
            if (!buf) return hipErrorOutOfMemory;  // same shape, same kernel plan, every time: never a silent other plan
            pp.n_full = plan.n_full;
            pp.nsplit = plan.nsplit;
            pp.part_cnt = (uint32_t*)buf;
            pp.part_buf = (float*)(buf + sc.split_cnt_bytes);
        } else if (plan.cbal) {
            // balanced causal pairs: flags first, the pairs' slots behind them.  Without the block (the pool may not grow while the stream
            // is capturing) the launch runs unpaired -- the same arithmetic per tile, another order of the row sums
            if (char* buf = sc.ensure_split(plan.cnt_bytes, plan.buf_bytes, stream)) {
                pp.cbal = 1;
                pp.cbal_delta = plan.cbal_delta;
                pp.part_cnt = (uint32_t*)buf;
                pp.part_buf = (float*)(buf + sc.split_cnt_bytes);
            }
        }
        // a mask the kernel would read per score with scalar loads (rows not aligned to four elements: any odd sequence length): one pass makes an aligned copy, rows padded
        // to four keys (fa_aux.hip launch_mask_realign), in the block the tile flags sit in front of.  Not under a guard (the pair's second route reads the caller's tensor), not
        // when the block cannot be had (capture without a warm-up: in place, as before)
        void* realign_blk = nullptr;
        if (!guard && mask_rows_scalar(pp) && !tuning().no_mask_realign.load(std::memory_order_relaxed)) {
            const size_t fb = (mask_flags_bytes(pp) + 255) & ~(size_t)255, rbytes = mask_realign_bytes(pp);
            if (rbytes <= ((size_t)1 << 30)) {
                if (void* blk = sc.mflags.ensure(fb + rbytes, stream)) {
                    if ((e = launch_mask_realign(pp, (char*)blk + fb, stream)) != hipSuccess) return e;
                    realign_blk = blk;
                }
            }
        }
        if (guard) {
            // (guarded: the classification pass of the first route wrote this route's tile flags on its way through the mask -- behind the verdict word; behind them,
            // the realigned copy of a mask with unaligned rows, made by a launch that checks the same verdict)
            uint8_t* const fl = (uint8_t*)const_cast<uint32_t*>(guard) + 256;
            const size_t fb = (mask_flags_bytes(pp) + 255) & ~(size_t)255;
            if (mask_rows_scalar(pp) && !tuning().no_mask_realign.load(std::memory_order_relaxed) && mask_realign_bytes(pp) <= ((size_t)1 << 30)) {
                if ((e = launch_mask_realign(pp, fl + fb, stream)) != hipSuccess) return e;
            }
            if (!tuning().no_mask_flags.load(std::memory_order_relaxed)) mask_flags_describe(pp, fl);
        } else if (pp.mask_kind != MK_NONE && !tuning().no_mask_flags.load(std::memory_order_relaxed) && mask_flags_worthwhile(pp)) {
            // tile early-exit for masks: one pre-pass over the distinct mask elements classifies every (32 rows x 64
            // keys) tile; fully masked tiles are skipped, fully open ones run without reading the mask.  Results are
            // bit-identical with and without the flags, so a pool that may not grow (capture) just runs without them.
            void* fl = realign_blk ? realign_blk : sc.mflags.ensure(mask_flags_bytes(pp), stream);
            if (fl && launch_mask_flags(pp, (uint8_t*)fl, stream) != hipSuccess) pp.mask_flags = nullptr;
        }
        e = launch_fwd_16(pp, stream, &name);
        if (guard && e == hipSuccess) name = both_kernels(guarded_first, name);
    } else {
        e = p.D > 256 ? launch_fwd_wide(p, stream, &name) : launch_fwd_exact(p, stream, &name);
    }
    ctx->last_kernel = name;
    DBG("forward B%u H%u Sq%u Skv%u D%u causal%d mask%d -> %s (%s)", p.B, p.H, p.Sq, p.Skv, p.D, p.causal,
        p.mask_kind, name, hipGetErrorString(e));
    return e;
}

// The synchronous forward in head chunks (see forward_sync; the plan: runtime_internal.h plan_sync_chunks).  ctx->mu held, the context's device
// current; p holds the whole call (device mirrors, dense strides): the operand strides stay the whole tensors'.
mfa_error_t forward_sync_chunked(Context* ctx, Buffer* bq, Buffer* bk, Buffer* bv, Buffer* bo, Buffer* bl, const FwdParams& p, int inter, uint32_t want) {
    const std::vector<SyncChunk> chunks = plan_sync_chunks(p.B, p.H, want);
    if (!sync_chunks_begin(ctx, chunks.size())) return MFA_ERROR_EXECUTION_FAILED;
    const size_t eb = elem_bytes(p.in_prec);
    const size_t qslab = (size_t)p.Sq * p.D, kslab = (size_t)p.Skv * p.D;
    mfa_error_t rc = MFA_SUCCESS;
    for (size_t c = 0; c < chunks.size() && rc == MFA_SUCCESS; ++c) {
        const SyncChunk& ch = chunks[c];
        hipStream_t s = ctx->side[c % 3];
        const size_t slab0 = (size_t)ch.b0 * p.H + ch.h0, nslab = (size_t)ch.nb * ch.nh;
        if (bq->upload_range(slab0 * qslab * eb, nslab * qslab * eb, s) != hipSuccess || bk->upload_range(slab0 * kslab * eb, nslab * kslab * eb, s) != hipSuccess ||
            bv->upload_range(slab0 * kslab * eb, nslab * kslab * eb, s) != hipSuccess) { rc = MFA_ERROR_EXECUTION_FAILED; break; }
        FwdParams pc = p;
        pc.B = ch.nb; pc.H = ch.nh;
        pc.q = (const char*)p.q + slab0 * qslab * eb;
        pc.k = (const char*)p.k + slab0 * kslab * eb;
        pc.v = (const char*)p.v + slab0 * kslab * eb;
        pc.o = (char*)p.o + slab0 * qslab * 4;
        pc.lse = p.lse ? p.lse + slab0 * p.Sq : nullptr;
        (void)hipEventRecord(ctx->chunk_ev[2 * c], s);
        const hipError_t e = dispatch_forward(ctx, ctx->pool(ctx->device, s), pc, inter, s);
        if (e != hipSuccess) { rc = rc_of(e); break; }
        (void)hipEventRecord(ctx->chunk_ev[2 * c + 1], s);
        if (bo->download_range(slab0 * qslab * 4, nslab * qslab * 4, s) != hipSuccess) { rc = MFA_ERROR_EXECUTION_FAILED; break; }
        if (bl && bl->mirrored() && bl->download_range(slab0 * p.Sq * 4, nslab * p.Sq * 4, s) != hipSuccess) { rc = MFA_ERROR_EXECUTION_FAILED; break; }
    }
    const hipError_t e = sync_chunks_end(ctx, chunks.size(), rc == MFA_SUCCESS);
    if (rc != MFA_SUCCESS) return rc;
    return e == hipSuccess ? MFA_SUCCESS : MFA_ERROR_EXECUTION_FAILED;
}

// Shared body of the synchronous dense forwards.
mfa_error_t forward_sync(mfa_context_t context, mfa_buffer_t q, mfa_buffer_t k, mfa_buffer_t v, mfa_buffer_t out,
                         mfa_buffer_t lse, uint32_t B, uint32_t Sq, uint32_t Skv, uint32_t H, uint16_t D,
                         float scale, bool causal, int in_prec_raw, int inter_prec_raw, bool tq, bool tk, bool tv,
                         bool to, const void* mask_ptr, size_t mask_bytes, const int64_t* mshape,
                         const int64_t* mstrides, uint32_t mndim, int mtype, int mscalar, bool want_lse) {
    Context* ctx = as_ctx(context);
    Buffer *bq = as_buf(q), *bk = as_buf(k), *bv = as_buf(v), *bo = as_buf(out), *bl = as_buf(lse);
    if (!ctx || !bq || !bk || !bv || !bo || (want_lse && !bl)) return MFA_ERROR_INVALID_ARGS;
    // a wrapped host pointer of unknown size has no HBM mirror: nothing a kernel could read or write
    if (!bq->dev || !bk->dev || !bv->dev || !bo->dev || (want_lse && !bl->dev)) return MFA_ERROR_INVALID_ARGS;
    // mask types beyond NONE / BOOL / ADDITIVE (e.g. the in-stream entry's UMFA_MASK_TYPE_WINDOW) do not exist on the
    // synchronous ABI: refuse instead of computing an unmasked result
    if (mtype != MFA_MASK_TYPE_NONE && mtype != MFA_MASK_TYPE_BOOL && mtype != MFA_MASK_TYPE_ADDITIVE) return MFA_ERROR_INVALID_ARGS;
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);  // the caller's current device is restored on return
    hipStream_t stream = nullptr;

    FwdParams p;
    memset(&p, 0, sizeof(p));
    p.B = B; p.H = H; p.Sq = Sq; p.Skv = Skv; p.D = D;
    p.scale = scale;
    p.causal = causal ? 1 : 0;
    p.in_prec = dense_prec(in_prec_raw);
    p.out_prec = P_FP32;  // output_precision is ignored by the reference: O is fp32 (MFABridge.swift:1089)
    const int inter = dense_prec(inter_prec_raw);
    dense_strides(p, tq, tk, tv, to);

    const size_t nq = (size_t)B * H * Sq * D, nkv = (size_t)B * H * Skv * D;
    const size_t eb = elem_bytes(p.in_prec);
    // bounds: refuse to overrun a wrapped buffer
    if (!bq->fits(nq * eb) || !bk->fits(nkv * eb) || !bv->fits(nkv * eb) || !bo->fits(nq * 4)) return MFA_ERROR_INVALID_ARGS;
    if (want_lse && !bl->fits((size_t)B * H * Sq * 4)) return MFA_ERROR_INVALID_ARGS;
    if (nq == 0 || nkv == 0) return MFA_SUCCESS;  // nothing to compute
    if (D > 1024) return MFA_ERROR_INVALID_ARGS;  // the reference's callers' own limit (metal_sdpa_backend.cpp:1082-1084); 257 ... 1024: fa_fwd_wide.hip

    p.q = bq->dev; p.k = bk->dev; p.v = bv->dev; p.o = bo->dev;
    p.lse = want_lse ? (float*)bl->dev : nullptr;

    // mask: copy the caller's bytes to HBM (the reference copies them too, MFABridge.swift:394-408)
    const bool mask_given = mtype != MFA_MASK_TYPE_NONE && mask_ptr && mask_bytes > 0 && mshape && mstrides && mndim > 0;
    if (mask_given && mndim <= 4 && normalise_mask(mshape, mstrides, mndim, mtype, mscalar, p)) {
        if (is_device_pointer(mask_ptr)) {
            p.mask = mask_ptr;
        } else {
            void* d = ctx->ensure_scratch(mask_bytes);
            if (!d) return MFA_ERROR_MEMORY_ALLOCATION;
            if (hipMemcpyAsync(d, mask_ptr, mask_bytes, hipMemcpyHostToDevice, stream) != hipSuccess)
                return MFA_ERROR_EXECUTION_FAILED;
            p.mask = d;
        }
    }

    // Host-wrapping buffers at sizes where the host link is the call (FLUX shape: 126 MB over the link, 2.3 ms, against 0.19 ms of kernels):
    // the heads go through in chunks on three side streams, so that one chunk's download runs under the next ones' uploads (the link is full
    // duplex) and the kernels under both.  Dense row-major operands, host ranges that could be pinned (pin_for_chunks), a mask only if it has no batch / head
    // extent; everything else takes the one-upload form below.  Same kernels, same numbers: a chunk is a launch of its own over whole heads.
    {
        const size_t moved = nq * eb + 2 * nkv * eb + nq * 4;
        const int want = sync_chunk_count(moved);
        if (want > 1 && !tq && !tk && !tv && !to && (uint64_t)B * H >= 2 && (!p.mask || (p.ms[0] == 0 && p.ms[1] == 0)) && pin_for_chunks({bq, bk, bv, bo}))
            return forward_sync_chunked(ctx, bq, bk, bv, bo, want_lse ? bl : nullptr, p, inter, (uint32_t)want);
    }
    if (bq->upload(stream) != hipSuccess || bk->upload(stream) != hipSuccess || bv->upload(stream) != hipSuccess)
        return MFA_ERROR_EXECUTION_FAILED;
    (void)hipEventRecord(ctx->ev0, stream);  // kernel-only GPU time -> mfa_get_gpu_latency
    hipError_t e = dispatch_forward(ctx, ctx->pool(ctx->device, stream), p, inter, stream);
    if (e != hipSuccess) return rc_of(e);
    (void)hipEventRecord(ctx->ev1, stream);
    if (bo->download(stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
    if (want_lse && bl->download(stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) == hipSuccess) ctx->last_latency = ms * 1e-3;
    return MFA_SUCCESS;
}

}  // namespace

extern "C" {

// ============================ context ============================
mfa_error_t mfa_create_context(mfa_context_t* context) {
    std::lock_guard<std::mutex> lock(g_ctx_mu);
    if (!g_ctx) {
        int dev = 0;
        if (!device_usable(&dev)) return MFA_ERROR_DEVICE_NOT_SUPPORTED;
        Context* c = new (std::nothrow) Context();
        if (!c) return MFA_ERROR_MEMORY_ALLOCATION;
        c->device = dev;
        if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
            delete c;
            return MFA_ERROR_MEMORY_ALLOCATION;
        }
        g_ctx = c;
    }
    g_ctx->refs.fetch_add(1);
    if (context) *context = g_ctx;
    return MFA_SUCCESS;
}

void mfa_destroy_context(mfa_context_t context) {
    Context* c = as_ctx(context);
    if (!c) return;
    c->refs.fetch_sub(1);  // the process-wide singleton itself stays alive (GlobalContextStore)
}

// ============================ buffers ============================
mfa_error_t mfa_create_buffer(mfa_context_t context, size_t size_bytes, mfa_buffer_t* buffer) {
    Context* ctx = as_ctx(context);
    if (!ctx || !buffer) return MFA_ERROR_INVALID_ARGS;
    Buffer* b = new (std::nothrow) Buffer();
    if (!b) return MFA_ERROR_MEMORY_ALLOCATION;
    b->bytes = size_bytes;
    const size_t alloc = size_bytes ? size_bytes : 16;
    // pinned host side (mfa_buffer_contents) + HBM mirror
    if (hipHostMalloc(&b->host, alloc, hipHostMallocDefault) != hipSuccess) {
        delete b;
        return MFA_ERROR_MEMORY_ALLOCATION;
    }
    b->owns_host = true;
    memset(b->host, 0, alloc);
    if (hipMalloc(&b->dev, alloc) != hipSuccess) {
        (void)hipHostFree(b->host);
        delete b;
        return MFA_ERROR_MEMORY_ALLOCATION;
    }
    b->owns_dev = true;
    *buffer = b;
    return MFA_SUCCESS;
}

mfa_error_t mfa_buffer_from_ptr(mfa_context_t context, void* data_ptr, size_t size_bytes, mfa_buffer_t* buffer) {
    if (!as_ctx(context) || !data_ptr || !buffer) return MFA_ERROR_INVALID_ARGS;
    return wrap_pointer(data_ptr, size_bytes, nullptr, nullptr, 0, false, buffer);
}

mfa_error_t mfa_buffer_from_ptr_with_strides(mfa_context_t context, void* data_ptr, size_t size_bytes,
                                             const int64_t* shape, const int64_t* strides, uint32_t ndim,
                                             mfa_buffer_t* buffer) {
    if (!as_ctx(context) || !data_ptr || !buffer || !shape || !strides || ndim == 0) return MFA_ERROR_INVALID_ARGS;
    return wrap_pointer(data_ptr, size_bytes, shape, strides, ndim, false, buffer);
}

// "metal_buffer" = raw device pointer on ROCm; context is unused by the reference too (MFABridge.swift:981)
mfa_error_t mfa_buffer_from_mtl_buffer(mfa_context_t, void* metal_buffer, size_t size_bytes, mfa_buffer_t* buffer) {
    if (!metal_buffer || !buffer) return MFA_ERROR_INVALID_ARGS;
    return wrap_pointer(metal_buffer, size_bytes, nullptr, nullptr, 0, true, buffer);
}

mfa_error_t mfa_buffer_from_mtl_buffer_with_strides(mfa_context_t, void* metal_buffer, size_t size_bytes,
                                                    const int64_t* shape, const int64_t* strides, uint32_t ndim,
                                                    mfa_buffer_t* buffer) {
    if (!metal_buffer || !buffer || !shape || !strides || ndim == 0) return MFA_ERROR_INVALID_ARGS;
    return wrap_pointer(metal_buffer, size_bytes, shape, strides, ndim, true, buffer);
}

void* mfa_buffer_contents(mfa_buffer_t buffer) {
    Buffer* b = as_buf(buffer);
    if (!b) return nullptr;
    return b->host ? b->host : b->dev;
}

void mfa_destroy_buffer(mfa_buffer_t buffer) {
    Buffer* b = as_buf(buffer);
    if (!b) return;
    if (b->registered && b->host && hipHostUnregister(b->host) != hipSuccess) (void)hipGetLastError();  // (the caller may have freed the range already)
    // (every synchronous entry has synchronised before it returned: nothing in flight reads or writes the mirror)
    if (b->owns_dev && b->dev && !(b->cached_mirror && mirror_cache().give(b->dev, b->bytes, b->mirror_dev))) (void)hipFree(b->dev);
    if (b->owns_host && b->host) (void)hipHostFree(b->host);
    b->magic = 0;
    delete b;
}

// ============================ dense forward ============================
mfa_error_t mfa_attention_forward(mfa_context_t context, mfa_buffer_t q, mfa_buffer_t k, mfa_buffer_t v,
                                  mfa_buffer_t out, uint32_t batch_size, uint32_t seq_len_q, uint32_t seq_len_kv,
                                  uint32_t num_heads, uint16_t head_dim, float softmax_scale, bool causal,
                                  mfa_precision_t input_precision, mfa_precision_t intermediate_precision,
                                  mfa_precision_t /*output_precision*/, bool transpose_q, bool transpose_k,
                                  bool transpose_v, bool transpose_o, const void* mask_ptr, size_t mask_size_bytes,
                                  const int64_t* mask_shape, const int64_t* mask_strides, uint32_t mask_ndim,
                                  mfa_mask_type_t mask_type, mfa_mask_scalar_t mask_scalar_type) {
    return forward_sync(context, q, k, v, out, nullptr, batch_size, seq_len_q, seq_len_kv, num_heads, head_dim,
                        softmax_scale, causal, input_precision, intermediate_precision, transpose_q, transpose_k,
                        transpose_v, transpose_o, mask_ptr, mask_size_bytes, mask_shape, mask_strides, mask_ndim,
                        mask_type, mask_scalar_type, false);
}

mfa_error_t mfa_attention_forward_str(mfa_context_t context, mfa_buffer_t q, mfa_buffer_t k, mfa_buffer_t v,
                                      mfa_buffer_t out, uint32_t batch_size, uint32_t seq_len_q,
                                      uint32_t seq_len_kv, uint32_t num_heads, uint16_t head_dim,
                                      float softmax_scale, bool causal, const char* input_precision,
                                      const char* intermediate_precision, const char* output_precision,
                                      bool transpose_q, bool transpose_k, bool transpose_v, bool transpose_o,
                                      const void* mask_ptr, size_t mask_size_bytes, const int64_t* mask_shape,
                                      const int64_t* mask_strides, uint32_t mask_ndim, mfa_mask_type_t mask_type,
                                      mfa_mask_scalar_t mask_scalar_type) {
    return mfa_attention_forward(context, q, k, v, out, batch_size, seq_len_q, seq_len_kv, num_heads, head_dim,
                                 softmax_scale, causal, parse_precision(input_precision),
                                 parse_precision(intermediate_precision), parse_precision(output_precision),
                                 transpose_q, transpose_k, transpose_v, transpose_o, mask_ptr, mask_size_bytes,
                                 mask_shape, mask_strides, mask_ndim, mask_type, mask_scalar_type);
}

int32_t mfa_attention_forward_with_lse(mfa_context_t context, mfa_buffer_t q, mfa_buffer_t k, mfa_buffer_t v,
                                       mfa_buffer_t out, mfa_buffer_t lse, uint32_t batch_size, uint32_t seq_len_q,
                                       uint32_t seq_len_kv, uint32_t num_heads, uint16_t head_dim,
                                       float softmax_scale, bool causal, int32_t input_precision,
                                       int32_t intermediate_precision, bool transpose_q, bool transpose_k,
                                       bool transpose_v, bool transpose_o) {
    return forward_sync(context, q, k, v, out, lse, batch_size, seq_len_q, seq_len_kv, num_heads, head_dim,
                        softmax_scale, causal, input_precision, intermediate_precision, transpose_q, transpose_k,
                        transpose_v, transpose_o, nullptr, 0, nullptr, nullptr, 0, MFA_MASK_TYPE_NONE, 0, true);
}

mfa_error_t umfa_attention_forward_stream(mfa_context_t context, void* stream, const void* q, const int64_t* q_strides,
                                          const void* k, const int64_t* k_strides, const void* v,
                                          const int64_t* v_strides, void* out, int32_t out_precision, float* lse,
                                          const void* mask, const int64_t* mask_shape, const int64_t* mask_strides,
                                          uint32_t mask_ndim, mfa_mask_type_t mask_type,
                                          mfa_mask_scalar_t mask_scalar_type, uint32_t batch_size, uint32_t seq_len_q,
                                          uint32_t seq_len_kv, uint32_t num_heads, uint16_t head_dim,
                                          float softmax_scale, bool causal, int32_t input_precision,
                                          int32_t intermediate_precision) {
    Context* ctx = as_ctx(context);
    if (!ctx || !q || !k || !v || !out) return MFA_ERROR_INVALID_ARGS;
    if (head_dim == 0 || head_dim > 1024) return MFA_ERROR_INVALID_ARGS;
    FwdParams p;
    memset(&p, 0, sizeof(p));
    p.B = batch_size; p.H = num_heads; p.Sq = seq_len_q; p.Skv = seq_len_kv; p.D = head_dim;
    p.scale = softmax_scale;
    p.causal = causal ? 1 : 0;
    p.in_prec = dense_prec(input_precision);
    p.out_prec = dense_prec(out_precision);
    dense_strides(p, false, false, false, false);
    auto take = [](int64_t* dst, const int64_t* src) -> bool {
        if (!src) return true;
        for (int i = 0; i < 4; ++i) dst[i] = src[i];
        return src[3] == 1 && src[0] >= 0 && src[1] >= 0 && src[2] >= 0;  // last dim contiguous, strides > 0
    };
    if (!take(p.qs, q_strides) || !take(p.ks, k_strides) || !take(p.vs, v_strides)) return MFA_ERROR_INVALID_ARGS;
    p.q = q; p.k = k; p.v = v; p.o = out; p.lse = lse;
    if ((int)mask_type == UMFA_MASK_TYPE_WINDOW) {
        // MI355X extra: sliding window without a mask tensor -- mask_shape = {left, right}: key attends iff
        // row - left <= key <= row + right (combine with `causal` for a look-back window); tiles outside the band are
        // never staged, tiles inside it run unmasked (fa_fwd_16_kernel.h)
        if (!mask_shape || mask_shape[0] < 0 || mask_shape[1] < 0) return MFA_ERROR_INVALID_ARGS;
        p.mask_kind = MK_WINDOW;
        p.win_left = (uint32_t)(mask_shape[0] > 0x3fffffff ? 0x3fffffff : mask_shape[0]);
        p.win_right = (uint32_t)(mask_shape[1] > 0x3fffffff ? 0x3fffffff : mask_shape[1]);
    } else if (mask_type != MFA_MASK_TYPE_NONE) {
        if (!mask || !mask_shape || !mask_strides || mask_ndim == 0 || mask_ndim > 4) return MFA_ERROR_INVALID_ARGS;
        normalise_mask(mask_shape, mask_strides, mask_ndim, mask_type, mask_scalar_type, p);
        p.mask = mask;
    }
    if ((size_t)batch_size * num_heads * seq_len_q * seq_len_kv == 0) return MFA_SUCCESS;
    // Scratch belongs to (device, stream): launches on different streams, from different host threads, never share
    // ticket words or partial slots; mu only covers the pool lookup and the (asynchronous) launch.
    std::lock_guard<std::mutex> lock(ctx->mu);
    const int dev = stream_device((hipStream_t)stream);
    DeviceGuard guard(dev);
    return rc_of(dispatch_forward(ctx, ctx->pool(dev, (hipStream_t)stream), p, dense_prec(intermediate_precision),
                                  (hipStream_t)stream));
}

// MI355X extra: RoPE + SDPA in one call (the reference's rope_scaled_dot_product_attention rotates Q and K into dense
// copies with mfa_rope_rotate_encode_mtl and then attends, metal_sdpa_backend.cpp:1472-1641).  Here K is rotated ONCE
// by the rotate kernel into the stream's workspace (every query block re-reads it: it must exist rotated), and Q is
// rotated in registers right after the Q fragment load of the attention kernel -- the dense Q_rot round trip (one read
// + one write of Q) and its launch are gone.  Bit-identical to rotate-then-attend (one shared rotation routine).
// cos / sin: fp32 [S, D] (table_batch_stride = 0) or [B, S, D] (= S * D), pair-duplicated; Sq == Skv; no mask.
mfa_error_t umfa_rope_attention_forward_stream(mfa_context_t context, void* stream, const void* q, const int64_t* q_strides,
                                               const void* k, const int64_t* k_strides, const void* v,
                                               const int64_t* v_strides, void* out, int32_t out_precision, float* lse,
                                               const float* cos_table, const float* sin_table, int64_t table_batch_stride,
                                               uint32_t batch_size, uint32_t seq_len_q, uint32_t seq_len_kv,
                                               uint32_t num_heads, uint16_t head_dim, float softmax_scale, bool causal,
                                               int32_t input_precision, int32_t intermediate_precision) {
    Context* ctx = as_ctx(context);
    if (!ctx || !q || !k || !v || !out || !cos_table || !sin_table) return MFA_ERROR_INVALID_ARGS;
    // head dims 257 ... 1024 (the callers' limit, metal_sdpa_backend.cpp:1078-1086): rotate, then the wide fp32 forward
    if (head_dim == 0 || head_dim > 1024 || (head_dim & 1) || seq_len_q != seq_len_kv) return MFA_ERROR_INVALID_ARGS;
    FwdParams p;
    memset(&p, 0, sizeof(p));
    p.B = batch_size; p.H = num_heads; p.Sq = seq_len_q; p.Skv = seq_len_kv; p.D = head_dim;
    p.scale = softmax_scale;
    p.causal = causal ? 1 : 0;
    p.in_prec = dense_prec(input_precision);
    p.out_prec = dense_prec(out_precision);
    dense_strides(p, false, false, false, false);
    auto take = [](int64_t* dst, const int64_t* src) -> bool {
        if (!src) return true;
        for (int i = 0; i < 4; ++i) dst[i] = src[i];
        return src[3] == 1 && src[0] >= 0 && src[1] >= 0 && src[2] >= 0;
    };
    int64_t ks_in[4];
    for (int i = 0; i < 4; ++i) ks_in[i] = p.ks[i];
    if (!take(p.qs, q_strides) || !take(ks_in, k_strides) || !take(p.vs, v_strides)) return MFA_ERROR_INVALID_ARGS;
    p.q = q; p.v = v; p.o = out; p.lse = lse;
    const size_t nelem = (size_t)batch_size * num_heads * seq_len_kv * head_dim;
    if (nelem == 0 || seq_len_q == 0) return MFA_SUCCESS;
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> lock(ctx->mu);
    const int dev = stream_device(st);
    DeviceGuard guard(dev);
    StreamScratch& sc = ctx->pool(dev, st);
    const size_t eb = elem_bytes(p.in_prec);
    const bool lowp = p.in_prec != P_FP32 && dense_prec(intermediate_precision) != P_FP32;
    FwdParams probe = p;  // dense K for the kernel-selection predicates; WITH the rotation (head_dim 64 has no fused-rope kernel)
    probe.k = k;
    probe.rope_cos = cos_table; probe.rope_sin = sin_table; probe.rope_tb = table_batch_stride;
    // in-kernel Q rotation: the 256-row kernel only, O in the operand type (fa_fwd16_w64.hip); everything else takes Q
    // through the rotate kernel too (same result, one more pass)
    const bool fuse_q = lowp && p.out_prec == p.in_prec && fwd_w64_supported(probe);
    char* ws = (char*)sc.workspace.ensure((fuse_q ? 1 : 2) * nelem * eb + 512, st);
    if (!ws) return MFA_ERROR_MEMORY_ALLOCATION;
    RopeParams r;
    memset(&r, 0, sizeof(r));
    r.cos_table = cos_table; r.sin_table = sin_table; r.table_batch_stride = table_batch_stride;
    r.B = batch_size; r.H = num_heads; r.S = seq_len_kv; r.D = head_dim;
    r.src = k; r.dst = ws;
    r.src_batch_stride = ks_in[0]; r.src_head_stride = ks_in[1]; r.src_seq_stride = ks_in[2];
    if (launch_rope(r, p.in_prec, st) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
    p.k = ws;  // dense BHSD (dense_strides above)
    if (fuse_q) {
        p.rope_cos = cos_table; p.rope_sin = sin_table; p.rope_tb = table_batch_stride;
    } else {  // fp32 / exact kernels take Q through the rotate kernel as well
        char* qrot = ws + ((nelem * eb + 255) & ~(size_t)255);
        r.src = q; r.dst = qrot; r.S = seq_len_q;
        r.src_batch_stride = p.qs[0]; r.src_head_stride = p.qs[1]; r.src_seq_stride = p.qs[2];
        if (launch_rope(r, p.in_prec, st) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
        p.q = qrot;
        p.qs[0] = (int64_t)num_heads * seq_len_q * head_dim; p.qs[1] = (int64_t)seq_len_q * head_dim; p.qs[2] = head_dim; p.qs[3] = 1;
    }
    return rc_of(dispatch_forward(ctx, sc, p, dense_prec(intermediate_precision), st));
}

// In-stream encode (MFABridge.swift:2377-2543): never commits, never waits.
mfa_error_t mfa_attention_encode_mtl(mfa_context_t context, void* command_buffer, void* q_buffer, int64_t q_offset,
                                     const int64_t* q_strides, void* k_buffer, int64_t k_offset,
                                     const int64_t* k_strides, void* v_buffer, int64_t v_offset,
                                     const int64_t* v_strides, void* out_buffer, int64_t out_offset,
                                     void* mask_buffer, int64_t mask_offset, const int64_t* mask_shape,
                                     const int64_t* mask_strides, uint32_t mask_ndim, mfa_mask_type_t mask_type,
                                     mfa_mask_scalar_t mask_scalar_type, uint32_t batch_size, uint32_t seq_len_q,
                                     uint32_t seq_len_kv, uint32_t num_heads, uint16_t head_dim, float softmax_scale,
                                     bool causal, const char* input_precision, const char* intermediate_precision) {
    if (!as_ctx(context) || !q_buffer || !k_buffer || !v_buffer || !out_buffer) return MFA_ERROR_INVALID_ARGS;
    if (q_offset < 0 || k_offset < 0 || v_offset < 0 || out_offset < 0) return MFA_ERROR_INVALID_ARGS;
    const void* mask = nullptr;
    if (mask_type != MFA_MASK_TYPE_NONE) {
        if (!mask_buffer || mask_offset < 0) return MFA_ERROR_INVALID_ARGS;
        mask = (const char*)mask_buffer + mask_offset;
    }
    return umfa_attention_forward_stream(
        context, command_buffer, (const char*)q_buffer + q_offset, q_strides, (const char*)k_buffer + k_offset,
        k_strides, (const char*)v_buffer + v_offset, v_strides, (char*)out_buffer + out_offset, MFA_PRECISION_FP32,
        nullptr, mask, mask_shape, mask_strides, mask_ndim, mask_type, mask_scalar_type, batch_size, seq_len_q,
        seq_len_kv, num_heads, head_dim, softmax_scale, causal, parse_precision(input_precision),
        parse_precision(intermediate_precision));
}

// ============================ utilities ============================
const char* mfa_error_string(mfa_error_t error) {
    const char* s = "Unknown error";
    switch (error) {
    case 0: s = "Success"; break;
    case 1: s = "Invalid arguments"; break;
    case 2: s = "Memory allocation failed"; break;
    case 3: s = "Device not supported"; break;
    case 4: s = "Kernel compilation failed"; break;
    case 5: s = "Execution failed"; break;
    }
    return strdup(s);  // caller frees (MFABridge.swift:1538)
}

bool mfa_is_device_supported(void) { return device_usable(nullptr); }

void mfa_get_version(int* major, int* minor, int* patch) {
    if (major) *major = 1;
    if (minor) *minor = 0;
    if (patch) *patch = 0;
}

double mfa_get_gpu_latency(mfa_context_t context) {
    Context* c = as_ctx(context);
    return c ? c->last_latency : 0.0;
}

int32_t mfa_has_native_bfloat(void) { return device_usable(nullptr) ? 1 : 0; }
int32_t mfa_has_native_bfloat_msl32(void) { return device_usable(nullptr) ? 1 : 0; }

const char* umfa_last_kernel_name(mfa_context_t context) {
    Context* c = as_ctx(context);
    return c ? c->last_kernel : "none";
}

mfa_error_t umfa_release_scratch(mfa_context_t context, void* stream, int32_t all_streams) {
    Context* c = as_ctx(context);
    if (!c) return MFA_ERROR_INVALID_ARGS;
    std::lock_guard<std::mutex> lock(c->mu);
    // the caller vouches for "no live graph"; in-flight launches are waited for here
    int n = 0, prev = 0;
    if (hipGetDeviceCount(&n) == hipSuccess && hipGetDevice(&prev) == hipSuccess) {
        for (auto& kv : c->pools)
            if (all_streams || kv.first.stream == (hipStream_t)stream) {
                (void)hipSetDevice(kv.first.dev);
                (void)hipDeviceSynchronize();
            }
        (void)hipSetDevice(prev);
    }
    c->release_pools((hipStream_t)stream, all_streams != 0);
    if (all_streams) mirror_cache().clear();  // HBM mirrors of destroyed host wrappers, kept for the next wrap
    return MFA_SUCCESS;
}

mfa_error_t umfa_set_option(mfa_context_t context, const char* name, const char* value) {
    Context* c = as_ctx(context);
    if (!c) return MFA_ERROR_INVALID_ARGS;
    if (!set_tuning(name, value)) return MFA_ERROR_INVALID_ARGS;
    return MFA_SUCCESS;
}

mfa_error_t umfa_get_option(mfa_context_t context, const char* name, char* value, size_t value_size) {
    Context* c = as_ctx(context);
    if (!c || !name || !value || value_size < 2) return MFA_ERROR_INVALID_ARGS;
    // read-only names of round 4's status words (the condition they reported no longer exists: V's range is handled per slab on the device) -- kept
    // for round-4 callers, always "0"
    if (!strcmp(name, "pv_fp16_status") || !strcmp(name, "pv_fp16_fallbacks")) { value[0] = '0'; value[1] = 0; return MFA_SUCCESS; }
    return get_tuning(name, value, value_size) ? MFA_SUCCESS : MFA_ERROR_INVALID_ARGS;
}

mfa_error_t mfa_set_scale_arrays(mfa_context_t context, const float* q_scales, uint32_t q_scales_count,
                                 const float* k_scales, uint32_t k_scales_count, const float* v_scales,
                                 uint32_t v_scales_count) {
    Context* c = as_ctx(context);
    if (!c) return MFA_ERROR_INVALID_ARGS;
    std::lock_guard<std::mutex> lock(c->mu);
    auto set = [](std::vector<float>& dst, const float* src, uint32_t n) {
        if (src && n) dst.assign(src, src + n);
        else dst.clear();
    };
    set(c->q_scales, q_scales, q_scales_count);
    set(c->k_scales, k_scales, k_scales_count);
    set(c->v_scales, v_scales, v_scales_count);
    return MFA_SUCCESS;
}

void mfa_get_quantized_layout(mfa_quantized_kernel_t, mfa_quantized_layout_t* out_layout) {
    if (!out_layout) return;
    int32_t* f = (int32_t*)out_layout;
    for (size_t i = 0; i < sizeof(mfa_quantized_layout_t) / sizeof(int32_t); ++i) f[i] = -1;
}

void mfa_get_quantized_capabilities(void* out_capabilities) {
    if (!out_capabilities) return;
    mfa_quantized_capabilities_t caps;
    memset(&caps, 0, sizeof(caps));
    caps.supports_multi_head_backward = true;
    caps.supports_blockwise_backward = true;
    caps.max_heads = 128;
    caps.max_block_size = 256;
    memcpy(out_capabilities, &caps, sizeof(caps));
}

// ============================ legacy "quantized" forwards ============================
// All of them run the dense forward on FP32 data (MFABridge+Quantized.swift:26-35,78-80,137-154;
// callers pass FP32 tensors, metal_sdpa_backend.cpp:2419-2422).
mfa_error_t mfa_attention_forward_quantized_direct(
    mfa_context_t context, mfa_buffer_t q, mfa_buffer_t k, mfa_buffer_t v, mfa_buffer_t out, uint32_t batch_size,
    uint32_t seq_len_q, uint32_t seq_len_kv, uint32_t num_heads, uint16_t head_dim, float softmax_scale, bool causal,
    float, int32_t, float, int32_t, float, int32_t, int32_t, int32_t, int32_t, int32_t, bool transpose_q,
    bool transpose_k, bool transpose_v, bool transpose_o) {
    if (!as_ctx(context) || !as_buf(q) || !as_buf(k) || !as_buf(v) || !as_buf(out)) return MFA_ERROR_INVALID_ARGS;
    if (!batch_size || !num_heads || !seq_len_q || !seq_len_kv || !head_dim)
        return MFA_ERROR_MEMORY_ALLOCATION;  // sic: the reference returns 2 here (MFABridge+Quantized.swift:83-99)
    return forward_sync(context, q, k, v, out, nullptr, batch_size, seq_len_q, seq_len_kv, num_heads, head_dim,
                        softmax_scale, causal, MFA_PRECISION_FP32, MFA_PRECISION_FP32, transpose_q, transpose_k,
                        transpose_v, transpose_o, nullptr, 0, nullptr, nullptr, 0, MFA_MASK_TYPE_NONE, 0, false);
}

mfa_error_t mfa_attention_forward_quantized(
    mfa_context_t context, mfa_buffer_t q, mfa_buffer_t k, mfa_buffer_t v, mfa_buffer_t out, uint32_t batch_size,
    uint32_t seq_len_q, uint32_t seq_len_kv, uint32_t num_heads, uint16_t head_dim, float softmax_scale, bool causal,
    float q_scale, int32_t q_zp, float k_scale, int32_t k_zp, float v_scale, int32_t v_zp, mfa_precision_t qp,
    mfa_precision_t kp, mfa_precision_t vp, mfa_precision_t op, bool tq, bool tk, bool tv, bool to) {
    return mfa_attention_forward_quantized_direct(context, q, k, v, out, batch_size, seq_len_q, seq_len_kv, num_heads,
                                                  head_dim, softmax_scale, causal, q_scale, q_zp, k_scale, k_zp,
                                                  v_scale, v_zp, qp, kp, vp, op, tq, tk, tv, to);
}

mfa_error_t mfa_attention_forward_quantized_unified(
    mfa_context_t context, mfa_buffer_t q, mfa_buffer_t k, mfa_buffer_t v, mfa_buffer_t out, uint32_t batch_size,
    uint32_t seq_len_q, uint32_t seq_len_kv, uint32_t num_heads, uint16_t head_dim, float softmax_scale, bool causal,
    float q_scale, int32_t q_zp, float k_scale, int32_t k_zp, float v_scale, int32_t v_zp, mfa_precision_t qp,
    mfa_precision_t kp, mfa_precision_t vp, mfa_precision_t op, int32_t, uint32_t, uint32_t, uint32_t, bool, bool,
    bool tq, bool tk, bool tv, bool to) {
    return mfa_attention_forward_quantized_direct(context, q, k, v, out, batch_size, seq_len_q, seq_len_kv, num_heads,
                                                  head_dim, softmax_scale, causal, q_scale, q_zp, k_scale, k_zp,
                                                  v_scale, v_zp, qp, kp, vp, op, tq, tk, tv, to);
}

mfa_error_t mfa_attention_forward_quantized_enhanced(
    mfa_context_t context, mfa_buffer_t q, mfa_buffer_t k, mfa_buffer_t v, mfa_buffer_t out, uint32_t batch_size,
    uint32_t seq_len_q, uint32_t seq_len_kv, uint32_t num_heads, uint16_t head_dim, float softmax_scale, bool causal,
    float q_scale, int32_t q_zp, float k_scale, int32_t k_zp, float v_scale, int32_t v_zp, mfa_precision_t qp,
    mfa_precision_t kp, mfa_precision_t vp, mfa_precision_t op, int32_t, uint32_t, uint32_t, uint32_t, bool, bool,
    bool tq, bool tk, bool tv, bool to) {
    return mfa_attention_forward_quantized_direct(context, q, k, v, out, batch_size, seq_len_q, seq_len_kv, num_heads,
                                                  head_dim, softmax_scale, causal, q_scale, q_zp, k_scale, k_zp,
                                                  v_scale, v_zp, qp, kp, vp, op, tq, tk, tv, to);
}

mfa_error_t mfa_multihead_attention_quantized_direct(
    mfa_context_t context, mfa_buffer_t q, mfa_buffer_t k, mfa_buffer_t v, mfa_buffer_t out, uint32_t batch_size,
    uint32_t seq_len_q, uint32_t seq_len_kv, uint32_t num_heads, uint16_t head_dim, float softmax_scale, bool causal,
    float q_scale, int32_t q_zp, float k_scale, int32_t k_zp, float v_scale, int32_t v_zp, int32_t qp, int32_t kp,
    int32_t vp) {
    return mfa_attention_forward_quantized_direct(context, q, k, v, out, batch_size, seq_len_q, seq_len_kv, num_heads,
                                                  head_dim, softmax_scale, causal, q_scale, q_zp, k_scale, k_zp,
                                                  v_scale, v_zp, qp, kp, vp, MFA_PRECISION_FP32, false, false, false,
                                                  false);
}

// ============================ not built this round: link, return 3 ============================
#define NOT_BUILT return MFA_ERROR_DEVICE_NOT_SUPPORTED
// In-stream rotary rotation (MFABridge.swift:2286-2375): never commits, never waits.
int mfa_rope_rotate_encode_mtl(void* context, void* command_buffer, void* src_buffer, int64_t src_offset,
                               int64_t src_batch_stride, int64_t src_head_stride, int64_t src_seq_stride,
                               void* dst_buffer, int64_t dst_offset, void* cos_buffer, int64_t cos_offset,
                               void* sin_buffer, int64_t sin_offset, int64_t table_batch_stride, bool negate_sin,
                               uint32_t batch_size, uint32_t num_heads, uint32_t seq_len, uint32_t head_dim,
                               const char* precision) {
    if (!as_ctx(context) || !src_buffer || !dst_buffer || !cos_buffer || !sin_buffer) return MFA_ERROR_INVALID_ARGS;
    if (src_offset < 0 || dst_offset < 0 || cos_offset < 0 || sin_offset < 0 || (head_dim & 1)) return MFA_ERROR_INVALID_ARGS;
    RopeParams p;
    memset(&p, 0, sizeof(p));
    p.src = (const char*)src_buffer + src_offset;
    p.dst = (char*)dst_buffer + dst_offset;
    p.cos_table = (const float*)((const char*)cos_buffer + cos_offset);
    p.sin_table = (const float*)((const char*)sin_buffer + sin_offset);
    p.src_batch_stride = src_batch_stride; p.src_head_stride = src_head_stride; p.src_seq_stride = src_seq_stride;
    p.table_batch_stride = table_batch_stride;
    p.B = batch_size; p.H = num_heads; p.S = seq_len; p.D = head_dim;
    p.negate_sin = negate_sin ? 1 : 0;
    const hipError_t e = launch_rope(p, dense_prec(parse_precision(precision)), (hipStream_t)command_buffer);
    return e == hipSuccess ? MFA_SUCCESS : e == hipErrorInvalidValue ? MFA_ERROR_INVALID_ARGS : MFA_ERROR_EXECUTION_FAILED;
}

// Group-wise Hadamard rotation, in place, synchronous (MFABridge.swift:3433-3459).  The ABI carries no element
// type: it is inferred from the buffer size (4 bytes per element -> fp32, 2 -> fp16).
int32_t mfa_hadamard_rotate(mfa_buffer_t data, uint32_t block_size, uint32_t num_blocks) {
    Buffer* b = as_buf(data);
    if (!b || block_size == 0 || num_blocks == 0) return MFA_ERROR_INVALID_ARGS;
    std::lock_guard<std::mutex> lock(g_ctx_mu);
    if (!g_ctx) return MFA_ERROR_INVALID_ARGS;  // the reference uses the global context (MFABridge.swift:3445)
    const size_t n = (size_t)block_size * num_blocks;
    int prec = P_FP32;
    if (b->bytes == n * 2) prec = P_FP16;
    else if (b->bytes != 0 && b->bytes < n * 4) return MFA_ERROR_INVALID_ARGS;
    hipStream_t stream = nullptr;
    if (b->upload(stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
    const hipError_t e = launch_hadamard(b->dev, block_size, num_blocks, prec, stream);
    if (e != hipSuccess) return e == hipErrorInvalidValue ? MFA_ERROR_INVALID_ARGS : MFA_ERROR_EXECUTION_FAILED;
    if (b->download(stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
    return MFA_SUCCESS;
}
mfa_error_t mfa_sparse_indexer_scores(mfa_context_t, mfa_buffer_t, mfa_buffer_t, uint32_t, uint32_t, uint32_t,
                                      uint32_t, uint16_t, float, mfa_buffer_t, mfa_buffer_t*) { NOT_BUILT; }
mfa_error_t mfa_mla_create_context(mfa_mla_context_t* context) {
    if (context) *context = nullptr;
    NOT_BUILT;
}
void mfa_mla_destroy_context(mfa_mla_context_t) {}
mfa_error_t mfa_mla_init_weights(mfa_mla_context_t, uint32_t, uint32_t, uint32_t) { NOT_BUILT; }
mfa_error_t mfa_mla_load_weights(mfa_mla_context_t, mfa_buffer_t, mfa_buffer_t) { NOT_BUILT; }
mfa_error_t mfa_mla_forward(mfa_mla_context_t, mfa_context_t, mfa_buffer_t, mfa_buffer_t*, mfa_buffer_t*, uint32_t,
                            uint32_t, uint32_t, uint32_t, uint32_t) { NOT_BUILT; }

}  // extern "C"
