// runtime_train.hip -- backward and runtime-quantised entry points of the C ABI.
//
// mfa_attention_backward           MFABridge.swift:3171-3282
// mfa_quantized_forward_with_lse   MFABridge+Quantized.swift:227-358
// mfa_quantized_backward           MFABridge+Quantized.swift:365-533
// mfa_attention_backward_{query,kv}_quantized[_ex]   MFABridge.swift:1623-2163 (pre-quantised operands)
#include "runtime_internal.h"

using namespace umfa;
using namespace umfa_rt;

namespace {

struct LatencyScope {  // kernel-only GPU time of a synchronous op -> mfa_get_gpu_latency
    Context* c;
    hipStream_t s;
    LatencyScope(Context* c_, hipStream_t s_) : c(c_), s(s_) { (void)hipEventRecord(c->ev0, s); }
    void stop() { (void)hipEventRecord(c->ev1, s); }
    void publish() {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) c->last_latency = ms * 1e-3;
    }
};

}  // namespace

extern "C" {

mfa_error_t mfa_attention_backward(mfa_context_t context, mfa_buffer_t dout, mfa_buffer_t q, mfa_buffer_t k,
                                   mfa_buffer_t v, mfa_buffer_t out, mfa_buffer_t softmax_lse, mfa_buffer_t dq,
                                   mfa_buffer_t dk, mfa_buffer_t dv, mfa_buffer_t d_buffer, uint32_t batch_size,
                                   uint32_t seq_len_q, uint32_t seq_len_kv, uint32_t num_heads, uint16_t head_dim,
                                   float softmax_scale, bool causal, mfa_precision_t input_precision,
                                   mfa_precision_t intermediate_precision, bool transpose_q, bool transpose_k,
                                   bool transpose_v, bool transpose_o) {
    Context* ctx = as_ctx(context);
    Buffer *bdo = as_buf(dout), *bq = as_buf(q), *bk = as_buf(k), *bv = as_buf(v), *bo = as_buf(out),
           *bl = as_buf(softmax_lse), *bdq = as_buf(dq), *bdk = as_buf(dk), *bdv = as_buf(dv), *bd = as_buf(d_buffer);
    if (!ctx || !bdo || !bq || !bk || !bv || !bo || !bl || !bdq || !bdk || !bdv || !bd) return MFA_ERROR_INVALID_ARGS;
    if (transpose_q || transpose_k || transpose_v || transpose_o) return MFA_ERROR_INVALID_ARGS;  // no caller sets them
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    hipStream_t stream = nullptr;
    const uint32_t B = batch_size, H = num_heads, Sq = seq_len_q, Skv = seq_len_kv, D = head_dim;
    const size_t nq = (size_t)B * H * Sq * D, nkv = (size_t)B * H * Skv * D, nr = (size_t)B * H * Sq;
    const int prec = dense_prec(input_precision);
    const size_t eb = elem_bytes(prec);
    if (!bdo->fits(nq * eb) || !bq->fits(nq * eb) || !bk->fits(nkv * eb) || !bv->fits(nkv * eb) || !bo->fits(nq * 4) ||
        !bl->fits(nr * 4) || !bdq->fits(nq * 4) || !bdk->fits(nkv * 4) || !bdv->fits(nkv * 4) || !bd->fits(nr * 4))
        return MFA_ERROR_INVALID_ARGS;
    if (nq == 0 || nkv == 0) return MFA_SUCCESS;
    if (D > 1024) return MFA_ERROR_INVALID_ARGS;  // like the forward: the reference callers' limit (metal_sdpa_backend.cpp:1078-1086)

    BwdParams p;
    memset(&p, 0, sizeof(p));
    p.dout = bdo->dev; p.q = bq->dev; p.k = bk->dev; p.v = bv->dev;
    p.o = (const float*)bo->dev; p.lse = (const float*)bl->dev;
    p.dq = (float*)bdq->dev; p.dk = (float*)bdk->dev; p.dv = (float*)bdv->dev; p.dvec = (float*)bd->dev;
    p.B = B; p.H = H; p.Sq = Sq; p.Skv = Skv; p.D = D;
    p.scale = softmax_scale; p.causal = causal ? 1 : 0;
    p.in_prec = prec; p.dout_prec = prec;
    const char* name = "none";
    // 16-bit operands with 16-bit intermediates -> MFMA backward; everything else -> fp32-exact backward
    const bool lowp = dense_prec(intermediate_precision) != P_FP32 && !tuning().bwd_exact.load(std::memory_order_relaxed);
    // Host-wrapping buffers where the host link is the call (FLUX shape: 302 MB over it): head chunks on side streams, as the synchronous forward
    // (runtime.hip forward_sync_chunked) -- a chunk's gradients go down under the next ones' operands coming up.  The small per-row tensors (LSE in,
    // D out) travel whole on the null stream, before and after.
    {
        const size_t moved = 2 * nq * eb + 2 * nkv * eb + nq * 4 + nq * 4 + 2 * nkv * 4;
        const int want = sync_chunk_count(moved);
        if (want > 1 && (uint64_t)B * H >= 2 && pin_for_chunks({bdo, bq, bk, bv, bo, bdq, bdk, bdv})) {
            if (bl->upload(stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
            const std::vector<SyncChunk> chunks = plan_sync_chunks(B, H, (uint32_t)want);
            if (!sync_chunks_begin(ctx, chunks.size())) return MFA_ERROR_EXECUTION_FAILED;
            const size_t qslab = (size_t)Sq * D, kslab = (size_t)Skv * D;
            mfa_error_t rc = MFA_SUCCESS;
            for (size_t c = 0; c < chunks.size(); ++c) {
                const SyncChunk& ch = chunks[c];
                hipStream_t s = ctx->side[c % 3];
                const size_t slab0 = (size_t)ch.b0 * H + ch.h0, nslab = (size_t)ch.nb * ch.nh;
                if (bdo->upload_range(slab0 * qslab * eb, nslab * qslab * eb, s) != hipSuccess || bq->upload_range(slab0 * qslab * eb, nslab * qslab * eb, s) != hipSuccess ||
                    bk->upload_range(slab0 * kslab * eb, nslab * kslab * eb, s) != hipSuccess || bv->upload_range(slab0 * kslab * eb, nslab * kslab * eb, s) != hipSuccess ||
                    bo->upload_range(slab0 * qslab * 4, nslab * qslab * 4, s) != hipSuccess) { rc = MFA_ERROR_EXECUTION_FAILED; break; }
                BwdParams pc = p;
                pc.B = ch.nb; pc.H = ch.nh;
                pc.dout = (const char*)p.dout + slab0 * qslab * eb;
                pc.q = (const char*)p.q + slab0 * qslab * eb;
                pc.k = (const char*)p.k + slab0 * kslab * eb;
                pc.v = (const char*)p.v + slab0 * kslab * eb;
                pc.o = p.o + slab0 * qslab; pc.lse = p.lse + slab0 * Sq;
                pc.dq = p.dq + slab0 * qslab; pc.dk = p.dk + slab0 * kslab; pc.dv = p.dv + slab0 * kslab; pc.dvec = p.dvec + slab0 * Sq;
                const bool m16 = lowp && bwd_16_supported(pc);
                if (m16) {
                    pc.rowc = (float*)ctx->pool(ctx->device, s).rowc.ensure(2 * nslab * Sq * sizeof(float), s);
                    if (!pc.rowc) { rc = MFA_ERROR_MEMORY_ALLOCATION; break; }
                }
                (void)hipEventRecord(ctx->chunk_ev[2 * c], s);
                const hipError_t e = m16 ? launch_bwd_16(pc, s, &name) : launch_bwd(pc, s, &name);
                ctx->last_kernel = name;
                if (e != hipSuccess) { rc = e == hipErrorInvalidValue ? MFA_ERROR_INVALID_ARGS : MFA_ERROR_EXECUTION_FAILED; break; }
                (void)hipEventRecord(ctx->chunk_ev[2 * c + 1], s);
                if (bdq->download_range(slab0 * qslab * 4, nslab * qslab * 4, s) != hipSuccess || bdk->download_range(slab0 * kslab * 4, nslab * kslab * 4, s) != hipSuccess ||
                    bdv->download_range(slab0 * kslab * 4, nslab * kslab * 4, s) != hipSuccess) { rc = MFA_ERROR_EXECUTION_FAILED; break; }
            }
            const hipError_t e = sync_chunks_end(ctx, chunks.size(), rc == MFA_SUCCESS);
            if (rc != MFA_SUCCESS) return rc;
            if (e != hipSuccess || bd->download(stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
            return MFA_SUCCESS;
        }
    }
    for (Buffer* b : {bdo, bq, bk, bv, bo, bl})
        if (b->upload(stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
    LatencyScope lat(ctx, stream);
    const bool mfma16 = lowp && bwd_16_supported(p);
    if (mfma16) {
        p.rowc = (float*)ctx->pool(ctx->device, stream).rowc.ensure(2 * nr * sizeof(float), stream);
        if (!p.rowc) return MFA_ERROR_MEMORY_ALLOCATION;
    }
    hipError_t e = mfma16 ? launch_bwd_16(p, stream, &name) : launch_bwd(p, stream, &name);
    ctx->last_kernel = name;
    if (e != hipSuccess) return e == hipErrorInvalidValue ? MFA_ERROR_INVALID_ARGS : MFA_ERROR_EXECUTION_FAILED;
    lat.stop();
    for (Buffer* b : {bdq, bdk, bdv, bd})
        if (b->download(stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
    if (hipStreamSynchronize(stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
    lat.publish();
    return MFA_SUCCESS;
}

// MI355X extra (not in the reference): the backward of mfa_attention_backward in-stream -- raw device pointers, the
// caller's stream, no upload / download / synchronise.  grads_in_input_type = true writes dQ, dK, dV rounded once to the
// operand type from the kernels' epilogues (the torch caller casts the fp32 gradients back immediately otherwise,
// metal_sdpa_backend.cpp:2799-2802); only the 16-bit MFMA backward offers it (else error 1 and the caller uses fp32).
// out_in_input_type = true: `out` is the O tensor in the operand type (what the forward returned to the framework)
// instead of a separate fp32 copy kept only for D = rowsum(dO o O): halves the activation the autograd graph holds.
mfa_error_t umfa_attention_backward_stream(mfa_context_t context, void* stream, const void* dout, const void* q,
                                           const void* k, const void* v, const void* out, const float* softmax_lse,
                                           void* dq, void* dk, void* dv, float* d_buffer, uint32_t batch_size,
                                           uint32_t seq_len_q, uint32_t seq_len_kv, uint32_t num_heads,
                                           uint16_t head_dim, float softmax_scale, bool causal, int32_t input_precision,
                                           int32_t intermediate_precision, bool grads_in_input_type,
                                           bool out_in_input_type) {
    Context* ctx = as_ctx(context);
    if (!ctx || !dout || !q || !k || !v || !out || !softmax_lse || !dq || !dk || !dv || !d_buffer) return MFA_ERROR_INVALID_ARGS;
    if (head_dim == 0 || head_dim > 1024) return MFA_ERROR_INVALID_ARGS;
    if ((size_t)batch_size * num_heads * seq_len_q * seq_len_kv == 0) return MFA_SUCCESS;
    BwdParams p;
    memset(&p, 0, sizeof(p));
    p.dout = dout; p.q = q; p.k = k; p.v = v; p.o = (const float*)out; p.lse = softmax_lse;
    p.dq = (float*)dq; p.dk = (float*)dk; p.dv = (float*)dv; p.dvec = d_buffer;
    p.B = batch_size; p.H = num_heads; p.Sq = seq_len_q; p.Skv = seq_len_kv; p.D = head_dim;
    p.scale = softmax_scale; p.causal = causal ? 1 : 0;
    p.in_prec = dense_prec(input_precision); p.dout_prec = p.in_prec;
    p.grad_in_type = grads_in_input_type ? 1 : 0;
    p.o_in_type = (out_in_input_type && p.in_prec != P_FP32) ? 1 : 0;  // D = rowsum(dO o O) from the rounded O the caller kept
    const bool lowp = dense_prec(intermediate_precision) != P_FP32 && !tuning().bwd_exact.load(std::memory_order_relaxed);
    const bool mfma16 = lowp && bwd_16_supported(p);
    if (grads_in_input_type && !mfma16) return MFA_ERROR_INVALID_ARGS;
    const char* name = "none";
    std::lock_guard<std::mutex> lock(ctx->mu);
    const int dev = stream_device((hipStream_t)stream);
    DeviceGuard guard(dev);
    if (mfma16) {  // row constants of the second kernel, from the scratch pool of this (device, stream)
        StreamScratch& sc = ctx->pool(dev, (hipStream_t)stream);
        p.rowc = (float*)sc.rowc.ensure((size_t)2 * batch_size * num_heads * seq_len_q * sizeof(float), (hipStream_t)stream);
        if (!p.rowc) return MFA_ERROR_MEMORY_ALLOCATION;
        // option bwd_ds_store: the dS-store form (5 products instead of 7 for a [B H Sq Skv] scratch in the operand type: HBM
        // capacity traded for matrix work on a 288 GB part).  Head_dim 128, non-causal, whole 128-blocks, scratch <= 8 GiB.
        if (tuning().bwd_ds_store.load(std::memory_order_relaxed) && head_dim == 128 && !causal && seq_len_q % 128 == 0 && seq_len_kv % 128 == 0) {
            const size_t bytes = (size_t)batch_size * num_heads * seq_len_q * seq_len_kv * 2;
            // (one (batch, head) slab of dS is addressed through a 32-bit buffer descriptor: Sq * Skv * 2 bytes must stay below 2 GiB)
            if (bytes <= ((size_t)8 << 30) && (uint64_t)seq_len_q * seq_len_kv * 2 < ((uint64_t)1 << 31))
                p.ds = sc.dsbuf.ensure(bytes, (hipStream_t)stream);  // NULL (capture, allocation): the recomputing form
        }
    }
    hipError_t e = mfma16 ? launch_bwd_16(p, (hipStream_t)stream, &name) : launch_bwd(p, (hipStream_t)stream, &name);
    ctx->last_kernel = name;
    return e == hipSuccess ? MFA_SUCCESS : e == hipErrorInvalidValue ? MFA_ERROR_INVALID_ARGS : MFA_ERROR_EXECUTION_FAILED;
}

// MI355X extra: the in-stream backward for grouped-query attention WITHOUT expanded K / V (the reference's route is
// repeat_interleave of K and V before both passes, metal_sdpa_backend.cpp:1694-1702): k, v are [B, Hkv, Skv, D]; query head
// h reads K / V head h / (Hq / Hkv) in place; dK / dV are produced per query head in fp32 scratch and summed over each
// group (deterministic order) into [B, Hkv, Skv, D] in the operand type or fp32.  16-bit MFMA backward only (head_dim
// 64 / 128 / 256, 16-bit operands): other calls return MFA_ERROR_INVALID_ARGS and the caller expands K / V itself.
mfa_error_t umfa_attention_backward_gqa_stream(mfa_context_t context, void* stream, const void* dout, const void* q,
                                               const void* k, const void* v, const void* out, const float* softmax_lse,
                                               void* dq, void* dk, void* dv, float* d_buffer, uint32_t batch_size,
                                               uint32_t seq_len_q, uint32_t seq_len_kv, uint32_t num_heads,
                                               uint32_t num_kv_heads, uint16_t head_dim, float softmax_scale, bool causal,
                                               int32_t input_precision, bool grads_in_input_type, bool out_in_input_type) {
    Context* ctx = as_ctx(context);
    if (!ctx || !dout || !q || !k || !v || !out || !softmax_lse || !dq || !dk || !dv || !d_buffer) return MFA_ERROR_INVALID_ARGS;
    if (num_kv_heads == 0 || num_kv_heads > num_heads || num_heads % num_kv_heads) return MFA_ERROR_INVALID_ARGS;
    if ((size_t)batch_size * num_heads * seq_len_q * seq_len_kv == 0) return MFA_SUCCESS;
    BwdParams p;
    memset(&p, 0, sizeof(p));
    p.dout = dout; p.q = q; p.k = k; p.v = v; p.o = (const float*)out; p.lse = softmax_lse;
    p.dq = (float*)dq; p.dvec = d_buffer;
    p.B = batch_size; p.H = num_heads; p.Hkv = num_kv_heads; p.Sq = seq_len_q; p.Skv = seq_len_kv; p.D = head_dim;
    p.scale = softmax_scale; p.causal = causal ? 1 : 0;
    p.in_prec = dense_prec(input_precision); p.dout_prec = p.in_prec;
    p.grad_in_type = grads_in_input_type ? 1 : 0;
    p.o_in_type = (out_in_input_type && p.in_prec != P_FP32) ? 1 : 0;
    const bool grouped = num_kv_heads != num_heads;
    p.dkdv_fp32 = grouped ? 1 : 0;
    p.dk = (float*)dk; p.dv = (float*)dv;  // (alignment check below; replaced by scratch when grouped)
    if (tuning().bwd_exact.load(std::memory_order_relaxed) || !bwd_16_supported(p)) return MFA_ERROR_INVALID_ARGS;
    std::lock_guard<std::mutex> lock(ctx->mu);
    const int dev = stream_device((hipStream_t)stream);
    DeviceGuard guard(dev);
    StreamScratch& sc = ctx->pool(dev, (hipStream_t)stream);
    const size_t nr = (size_t)batch_size * num_heads * seq_len_q, nkvq = (size_t)batch_size * num_heads * seq_len_kv * head_dim;
    p.rowc = (float*)sc.rowc.ensure(2 * nr * sizeof(float), (hipStream_t)stream);
    if (!p.rowc) return MFA_ERROR_MEMORY_ALLOCATION;
    if (grouped) {
        float* tmp = (float*)sc.workspace.ensure(2 * nkvq * sizeof(float) + 256, (hipStream_t)stream);
        if (!tmp) return MFA_ERROR_MEMORY_ALLOCATION;
        p.dk = tmp;
        p.dv = tmp + nkvq;
    }
    const char* name = "none";
    hipError_t e = launch_bwd_16(p, (hipStream_t)stream, &name);
    ctx->last_kernel = name;
    if (e != hipSuccess) return e == hipErrorInvalidValue ? MFA_ERROR_INVALID_ARGS : MFA_ERROR_EXECUTION_FAILED;
    if (grouped) {
        const int oprec = grads_in_input_type ? p.in_prec : P_FP32;
        const int64_t slab = (int64_t)seq_len_kv * head_dim;
        if (launch_group_sum(p.dk, dk, batch_size, num_heads, num_kv_heads, slab, (hipStream_t)stream, oprec) != hipSuccess ||
            launch_group_sum(p.dv, dv, batch_size, num_heads, num_kv_heads, slab, (hipStream_t)stream, oprec) != hipSuccess)
            return MFA_ERROR_EXECUTION_FAILED;
    }
    return MFA_SUCCESS;
}

int32_t mfa_quantized_forward_with_lse(mfa_context_t context, mfa_buffer_t q, mfa_buffer_t k, mfa_buffer_t v,
                                       mfa_buffer_t out, mfa_buffer_t lse, mfa_buffer_t mask, uint32_t batch_size,
                                       uint32_t seq_len_q, uint32_t seq_len_kv, uint32_t num_heads, uint16_t head_dim,
                                       float softmax_scale, bool causal, int32_t target_precision, int32_t quant_mode,
                                       int32_t input_precision) {
    Context* ctx = as_ctx(context);
    Buffer *bq = as_buf(q), *bk = as_buf(k), *bv = as_buf(v), *bo = as_buf(out), *bl = as_buf(lse), *bm = as_buf(mask);
    if (!ctx || !bq || !bk || !bv || !bo || !bl) return MFA_ERROR_INVALID_ARGS;
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    hipStream_t stream = nullptr;
    const int pool_dev = ctx->device;
    const uint32_t B = batch_size, H = num_heads, Sq = seq_len_q, Skv = seq_len_kv, D = head_dim;
    const size_t nq = (size_t)B * H * Sq * D, nkv = (size_t)B * H * Skv * D, nr = (size_t)B * H * Sq;
    // inputPrecision: 0 fp16, 1 bf16, anything else fp32 (MFABridge+Quantized.swift:274-279)
    const int prec = dense_prec(input_precision);
    const size_t eb = elem_bytes(prec);
    if (!bq->fits(nq * eb) || !bk->fits(nkv * eb) || !bv->fits(nkv * eb) || !bo->fits(nq * 4) || !bl->fits(nr * 4))
        return MFA_ERROR_INVALID_ARGS;
    if (bm && !bm->fits(nr * Skv * 4)) return MFA_ERROR_INVALID_ARGS;
    if (nq == 0 || nkv == 0) return MFA_SUCCESS;
    if (!quantized_supported(D) || !(softmax_scale > 0.0f)) return MFA_ERROR_INVALID_ARGS;
    const int bits = target_precision == MFA_PRECISION_INT4 ? 4 : 8;  // unknown raw value -> INT8 (:267)
    // default tensor-wise (:268-272); 3 = UMFA_QUANT_BLOCKWISE_FP8PV (additive, include/umfa_abi.h): block-wise int8 Q / K, fp8 P V
    const int mode = quant_mode == 2 ? 2 : quant_mode == 3 ? 3 : 0;

    void* ws = ctx->pool(pool_dev, stream).workspace.ensure(quant_workspace_bytes(B, H, Sq, Skv, D, false), stream);
    if (!ws) return MFA_ERROR_MEMORY_ALLOCATION;
    for (Buffer* b : {bq, bk, bv})
        if (b->upload(stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
    if (bm && bm->upload(stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
    FwdParams p;
    memset(&p, 0, sizeof(p));
    p.q = bq->dev; p.k = bk->dev; p.v = bv->dev; p.o = bo->dev; p.lse = (float*)bl->dev;
    p.mask = bm ? bm->dev : nullptr;
    p.B = B; p.H = H; p.Sq = Sq; p.Skv = Skv; p.D = D;
    p.scale = softmax_scale; p.causal = causal ? 1 : 0;
    p.in_prec = prec; p.out_prec = P_FP32;
    if (fwd_w64_i8_supported(p)) {  // scratch of the 64-rows-per-wave kernel (tickets + partials), as for the dense path
        const FwdW64Plan plan = fwd_w64_plan(p);
        StreamScratch& sc = ctx->pool(pool_dev, (hipStream_t)stream);
        char* w64 = sc.ensure_w64(plan.cnt_bytes, plan.buf_bytes, (hipStream_t)stream);
        if (!w64) return MFA_ERROR_MEMORY_ALLOCATION;
        p.part_cnt = (uint32_t*)w64;
        p.part_buf = (float*)(w64 + sc.w64_cnt_bytes);
    }
    {   // slab headers for the fp16 V image's power of two (fa_quant.hip QuantParams::vhdr): the default bf16 forward's pool, headers only
        char* vh = ctx->pool(pool_dev, stream).ensure_v16((size_t)B * H, 0, stream);
        if (!vh) return MFA_ERROR_MEMORY_ALLOCATION;
        p.vsc = (const float*)vh; p.vsc_bs = H; p.vsc_hs = 1;
    }
    LatencyScope lat(ctx, stream);
    const char* name = "none";
    hipError_t e = launch_quantized_fwd(p, bits, mode, ws, stream, &name);
    ctx->last_kernel = name;
    DBG("quantized forward bits%d mode%d -> %s (%s)", bits, mode, name, hipGetErrorString(e));
    if (e != hipSuccess) return e == hipErrorInvalidValue ? MFA_ERROR_INVALID_ARGS : MFA_ERROR_EXECUTION_FAILED;
    lat.stop();
    if (bo->download(stream) != hipSuccess || bl->download(stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
    if (hipStreamSynchronize(stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
    lat.publish();
    return MFA_SUCCESS;
}

// MI355X extra (not in the reference): mfa_quantized_forward_with_lse in-stream -- dense BHSD device pointers, the
// caller's stream, no upload / download / synchronise (the reference's entry blocks; a serving loop should not).
// O is fp32 [B,H,Sq,D]; lse and mask (fp32 additive [B,H,Sq,Skv]) optional.  The quantiser workspace and the split-item
// scratch belong to the (device, stream) pool, so calls on different streams do not interfere.
static mfa_error_t quantized_forward_stream_impl(mfa_context_t context, void* stream, const void* q, const void* k, const void* v, float* out, float* lse,
                                                const void* mask, const int64_t* mask_shape, const int64_t* mask_strides, uint32_t mask_ndim,
                                                int32_t mask_type, int32_t mask_scalar_type, uint32_t batch_size, uint32_t seq_len_q, uint32_t seq_len_kv,
                                                uint32_t num_heads, uint16_t head_dim, float softmax_scale, bool causal, int32_t target_precision,
                                                int32_t quant_mode, int32_t input_precision) {
    Context* ctx = as_ctx(context);
    if (!ctx || !q || !k || !v || !out) return MFA_ERROR_INVALID_ARGS;
    const uint32_t B = batch_size, H = num_heads, Sq = seq_len_q, Skv = seq_len_kv, D = head_dim;
    if ((size_t)B * H * Sq * Skv == 0) return MFA_SUCCESS;
    if (!quantized_supported(D) || !(softmax_scale > 0.0f)) return MFA_ERROR_INVALID_ARGS;
    const int bits = target_precision == MFA_PRECISION_INT4 ? 4 : 8;
    const int mode = quant_mode == 2 ? 2 : quant_mode == 3 ? 3 : 0;
    std::lock_guard<std::mutex> lock(ctx->mu);  // pool lookup + launch; scratch is per (device, stream)
    const int pool_dev = stream_device((hipStream_t)stream);
    DeviceGuard guard(pool_dev);
    void* ws = ctx->pool(pool_dev, (hipStream_t)stream).workspace.ensure(quant_workspace_bytes(B, H, Sq, Skv, D, false), (hipStream_t)stream);
    if (!ws) return MFA_ERROR_MEMORY_ALLOCATION;
    FwdParams p;
    memset(&p, 0, sizeof(p));
    p.q = q; p.k = k; p.v = v; p.o = out; p.lse = lse;
    p.B = B; p.H = H; p.Sq = Sq; p.Skv = Skv; p.D = D;
    p.scale = softmax_scale; p.causal = causal ? 1 : 0;
    p.in_prec = dense_prec(input_precision); p.out_prec = P_FP32;
    if (mask && mask_ndim == 0) {
        p.mask = mask;  // the reference ABI's form: dense fp32 additive [B, H, Sq, Skv] (launch_quantized_fwd: mask_kind MK_NONE + a mask)
    } else if (mask && mask_type != MFA_MASK_TYPE_NONE) {
        if (mask_ndim > 4 || !mask_shape || !mask_strides) return MFA_ERROR_INVALID_ARGS;
        if (normalise_mask(mask_shape, mask_strides, mask_ndim, mask_type, mask_scalar_type, p)) p.mask = mask;  // (additive bytes: no mask, as on the dense path)
    }
    if (fwd_w64_i8_supported(p)) {
        const FwdW64Plan plan = fwd_w64_plan(p);
        StreamScratch& sc = ctx->pool(pool_dev, (hipStream_t)stream);
        // a bool mask tensor on the one-wave-per-SIMD int8 kernel (round 6): packed once per call like the 16-bit route's (runtime.hip dispatch_forward); a pool
        // that may not grow (capture without a warm-up) leaves the call to fa_fwd_i8, which reads the mask in place
        void* mk = p.mask_kind == MK_BOOL ? sc.mflags.ensure(mask_pack_bytes(p), (hipStream_t)stream) : nullptr;
        char* w64 = (p.mask_kind != MK_BOOL || mk) ? sc.ensure_w64(plan.cnt_bytes, plan.buf_bytes, (hipStream_t)stream) : nullptr;
        if (!w64 && p.mask_kind != MK_BOOL) return MFA_ERROR_MEMORY_ALLOCATION;
        if (w64) {
            if (mk && launch_mask_pack(p, mk, (hipStream_t)stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
            p.part_cnt = (uint32_t*)w64;
            p.part_buf = (float*)(w64 + sc.w64_cnt_bytes);
        }
    }
    {   // slab headers for the fp16 V image's power of two, see mfa_quantized_forward_with_lse
        char* vh = ctx->pool(pool_dev, (hipStream_t)stream).ensure_v16((size_t)B * H, 0, (hipStream_t)stream);
        if (!vh) return MFA_ERROR_MEMORY_ALLOCATION;
        p.vsc = (const float*)vh; p.vsc_bs = H; p.vsc_hs = 1;
    }
    const char* name = "none";
    hipError_t e = launch_quantized_fwd(p, bits, mode, ws, (hipStream_t)stream, &name);
    ctx->last_kernel = name;
    return e == hipSuccess ? MFA_SUCCESS : e == hipErrorInvalidValue ? MFA_ERROR_INVALID_ARGS : MFA_ERROR_EXECUTION_FAILED;
}

mfa_error_t umfa_quantized_forward_stream(mfa_context_t context, void* stream, const void* q, const void* k,
                                          const void* v, float* out, float* lse, const float* mask, uint32_t batch_size,
                                          uint32_t seq_len_q, uint32_t seq_len_kv, uint32_t num_heads, uint16_t head_dim,
                                          float softmax_scale, bool causal, int32_t target_precision, int32_t quant_mode,
                                          int32_t input_precision) {
    return quantized_forward_stream_impl(context, stream, q, k, v, out, lse, mask, nullptr, nullptr, 0, MFA_MASK_TYPE_ADDITIVE, MFA_MASK_SCALAR_FP32, batch_size,
                                         seq_len_q, seq_len_kv, num_heads, head_dim, softmax_scale, causal, target_precision, quant_mode, input_precision);
}

// MI355X extra: umfa_quantized_forward_stream with the mask the caller HAS -- any <= 4-D broadcastable bool / fp16 / bf16 / fp32 tensor with
// element strides, as umfa_attention_forward_stream takes it (mfa_prepare_mask's semantics, MFABridge.swift:157-242) -- instead of the dense
// fp32 [B, H, Sq, Skv] expansion the reference's quantised entry is handed (4.3 GB at B1 H16 S8192: the call was bound by reading it).
mfa_error_t umfa_quantized_forward_masked_stream(mfa_context_t context, void* stream, const void* q, const void* k, const void* v, float* out,
                                                 float* lse, const void* mask, const int64_t* mask_shape, const int64_t* mask_strides,
                                                 uint32_t mask_ndim, int32_t mask_type, int32_t mask_scalar_type, uint32_t batch_size,
                                                 uint32_t seq_len_q, uint32_t seq_len_kv, uint32_t num_heads, uint16_t head_dim,
                                                 float softmax_scale, bool causal, int32_t target_precision, int32_t quant_mode,
                                                 int32_t input_precision) {
    if (mask && mask_type != MFA_MASK_TYPE_NONE && (mask_ndim == 0 || mask_ndim > 4)) return MFA_ERROR_INVALID_ARGS;
    return quantized_forward_stream_impl(context, stream, q, k, v, out, lse, mask_type == MFA_MASK_TYPE_NONE ? nullptr : mask, mask_shape, mask_strides,
                                         mask_type == MFA_MASK_TYPE_NONE ? 0 : mask_ndim, mask_type, mask_scalar_type, batch_size, seq_len_q, seq_len_kv,
                                         num_heads, head_dim, softmax_scale, causal, target_precision, quant_mode, input_precision);
}

// The 16-bit MFMA backward as the engine of the quantised backward entries: head_dim 64 / 128 / 256, no mask, not forced off
static bool bwd16_shape_ok(uint32_t D, bool has_mask) {
    return !has_mask && (D == 64 || D == 128 || D == 256) && !tuning().bwd_exact.load(std::memory_order_relaxed);
}

// Backward of the quantised forward: re-quantise Q, K, V deterministically (same kernels, same scales as the
// forward), then the fp32 backward on the de-quantised operands -- the reference's "dequantise-on-load into FP32
// tiles -> FP32 math" (AGENTS.md:143-152); gradients flow straight through the rounding (STE).
int32_t mfa_quantized_backward(mfa_context_t context, mfa_buffer_t q, mfa_buffer_t k, mfa_buffer_t v, mfa_buffer_t out,
                               mfa_buffer_t grad_out, mfa_buffer_t lse, mfa_buffer_t grad_q, mfa_buffer_t grad_k,
                               mfa_buffer_t grad_v, mfa_buffer_t mask, uint32_t batch_size, uint32_t seq_len_q,
                               uint32_t seq_len_kv, uint32_t num_heads, uint16_t head_dim, float softmax_scale,
                               bool causal, int32_t target_precision, int32_t quant_mode, int32_t input_precision) {
    Context* ctx = as_ctx(context);
    Buffer *bq = as_buf(q), *bk = as_buf(k), *bv = as_buf(v), *bo = as_buf(out), *bdo = as_buf(grad_out),
           *bl = as_buf(lse), *bdq = as_buf(grad_q), *bdk = as_buf(grad_k), *bdv = as_buf(grad_v), *bm = as_buf(mask);
    if (!ctx || !bq || !bk || !bv || !bo || !bdo || !bl || !bdq || !bdk || !bdv) return MFA_ERROR_INVALID_ARGS;
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    hipStream_t stream = nullptr;
    const int pool_dev = ctx->device;
    const uint32_t B = batch_size, H = num_heads, Sq = seq_len_q, Skv = seq_len_kv, D = head_dim;
    const size_t nq = (size_t)B * H * Sq * D, nkv = (size_t)B * H * Skv * D, nr = (size_t)B * H * Sq;
    const int prec = dense_prec(input_precision);
    const size_t eb = elem_bytes(prec);
    if (!bq->fits(nq * eb) || !bk->fits(nkv * eb) || !bv->fits(nkv * eb) || !bo->fits(nq * 4) || !bdo->fits(nq * eb) ||
        !bl->fits(nr * 4) || !bdq->fits(nq * 4) || !bdk->fits(nkv * 4) || !bdv->fits(nkv * 4))
        return MFA_ERROR_INVALID_ARGS;
    if (bm && !bm->fits(nr * Skv * 4)) return MFA_ERROR_INVALID_ARGS;
    if (nq == 0 || nkv == 0) return MFA_SUCCESS;
    if (!quantized_supported(D)) return MFA_ERROR_INVALID_ARGS;  // head_dim <= 1024, multiple of 8 (257 ... 1024: the fp32 engines on q * s images, no mask)
    const int bits = target_precision == MFA_PRECISION_INT4 ? 4 : 8;
    const int mode = quant_mode == 2 ? 2 : 0;

    for (Buffer* b : {bq, bk, bv, bo, bdo, bl})
        if (b->upload(stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
    if (bm && bm->upload(stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
    // Two ways to run it.  FAST (head_dim 64 / 128 / 256, no mask): the de-quantised operands q * s go out of the quantiser
    // as fp16 (a 7-bit integer times a scale: fp16 holds it to 2^-12; the forward's V image is the same thing), dO is cast
    // to fp16, and the 16-bit MFMA backward runs (MFABridge+Quantized.swift:365-533 dispatches two kernels on quantised
    // operands as well).  A value outside fp16's range raises a device flag, read after the call's synchronise: the call is
    // then repeated on the EXACT path (fp32 copies -> fp32-exact backward), which is also what masks and other head dims take.
    const bool try_fast = bwd16_shape_ok(D, bm != nullptr);
    for (int attempt = try_fast ? 0 : 1; attempt < 2; ++attempt) {
        const bool fast = attempt == 0;
        // workspace: quantiser output + copies | D vector the callee owns (MFABridge+Quantized.swift:470-474) | fast: dO fp16, row constants, flag
        const size_t wq = (quant_workspace_bytes(B, H, Sq, Skv, D, true) + 255) & ~(size_t)255;
        const size_t o_dvec = wq, o_do16 = o_dvec + ((nr * 4 + 255) & ~(size_t)255), o_rowc = o_do16 + ((nq * 2 + 255) & ~(size_t)255),
                     o_end = o_rowc + ((2 * nr * 4 + 255) & ~(size_t)255);
        char* ws = (char*)ctx->pool(pool_dev, stream).workspace.ensure((fast ? o_end : o_do16) + 256, stream);
        if (!ws) return MFA_ERROR_MEMORY_ALLOCATION;
        // overflow word + units header: a block of its own at a fixed address, zero between calls (StreamScratch::ensure_qhdr; bwd_units_kernel cleans it)
        uint32_t* flag = fast ? ctx->pool(pool_dev, stream).ensure_qhdr(stream) : nullptr;
        if (fast && !flag) return MFA_ERROR_MEMORY_ALLOCATION;
        // FAST: every operand goes to the fp16 engine as a power-of-two multiple with its largest magnitude in [1, 2) -- the de-quantised Q, K, V
        // (the quantiser's fp16 copies) and dO; the exponents are found on the device (one amax pass per tensor) and come back through
        // BwdParams::units.  Nothing can leave fp16's range then, dS = P (dP - D) included (|dS| <= 8 head_dim): no flag to read, no repeat.
        uint32_t* unit = fast ? flag + 16 : nullptr;  // 16 words: kernels.h launch_bwd_units
        LatencyScope lat(ctx, stream);
        if (fast) {  // one launch for the four amax words (dO's is unit[0])
            const void* const srcs[4] = {bq->dev, bk->dev, bv->dev, bdo->dev};
            const int64_t ns[4] = {(int64_t)nq, (int64_t)nkv, (int64_t)nkv, (int64_t)nq};
            uint32_t* const words[4] = {unit + 4, unit + 5, unit + 6, unit};
            if (launch_amax_dense_n(4, srcs, prec, ns, words, stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
        }
        auto dirty = [&](mfa_error_t rc) { if (fast) ctx->pool(pool_dev, stream).drop_qhdr(); return rc; };  // amax words set, bwd_units_kernel not reached
        QuantViews views;
        hipError_t e = launch_quantize(bq->dev, bk->dev, bv->dev, prec, B, H, Sq, Skv, D, bits, mode, ws, fast ? 2 : 1, &views, stream,
                                       fast ? flag : nullptr, nullptr, fast ? unit + 4 : nullptr);
        if (e != hipSuccess) return dirty(MFA_ERROR_EXECUTION_FAILED);
        BwdParams p;
        memset(&p, 0, sizeof(p));
        p.o = (const float*)bo->dev; p.lse = (const float*)bl->dev;
        p.dq = (float*)bdq->dev; p.dk = (float*)bdk->dev; p.dv = (float*)bdv->dev;
        p.dvec = (float*)(ws + o_dvec);
        p.B = B; p.H = H; p.Sq = Sq; p.Skv = Skv; p.D = D;
        p.scale = softmax_scale; p.causal = causal ? 1 : 0;
        const char* name = "none";
        if (fast) {
            p.q = views.qh; p.k = views.kh; p.v = views.vh;
            p.in_prec = P_FP16; p.dout_prec = P_FP16;
            // dO as dO * 2^-e in fp16, e from its largest magnitude on the device (gradients of 1e-7 are ordinary; as a plain cast they
            // were fp16 subnormals): fa_aux.hip launch_cast_f16_unit, 2^e comes back in the kernels' epilogues
            if (launch_cast_f16_unit(bdo->dev, prec, ws + o_do16, (int64_t)nq, unit, stream, true) != hipSuccess || launch_bwd_units(unit, stream, flag) != hipSuccess)
                return dirty(MFA_ERROR_EXECUTION_FAILED);
            p.dout = ws + o_do16;
            p.units = (const float*)(unit + 8);
            p.rowc = (float*)(ws + o_rowc);
            e = bwd_16_supported(p) ? launch_bwd_16(p, stream, &name) : hipErrorNotSupported;
            if (e == hipErrorNotSupported) continue;  // (alignment of a wrapped caller buffer): the exact path
        } else {
            p.dout = bdo->dev; p.q = views.qf; p.k = views.kf; p.v = views.vf;
            p.mask = bm ? (const float*)bm->dev : nullptr;
            p.in_prec = P_FP32; p.dout_prec = prec;
            e = launch_bwd(p, stream, &name);
        }
        ctx->last_kernel = name;
        if (e != hipSuccess) return e == hipErrorInvalidValue ? MFA_ERROR_INVALID_ARGS : MFA_ERROR_EXECUTION_FAILED;
        lat.stop();
        for (Buffer* b : {bdq, bdk, bdv})
            if (b->download(stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
        if (hipStreamSynchronize(stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
        lat.publish();
        break;  // (the second attempt is the EXACT path for shapes / alignments the fp16 engine does not take: `continue` above)
    }
    return MFA_SUCCESS;
}

// MI355X extra (not in the reference): mfa_quantized_backward in-stream -- dense BHSD device pointers, the caller's stream,
// never synchronises.  Engine as in the blocking entry: the 16-bit MFMA backward on fp16 de-quantised operands where the
// shape allows (head_dim 64 / 128 / 256), else the fp32-exact one.  An in-stream call cannot repeat itself, so a value
// outside fp16's range is REPORTED instead: `status` (device, one uint32, may be NULL) is zeroed on the stream and ORed
// with 1 by the kernels that saw one -- the gradients are then not valid and the caller uses the blocking entry (or sets
// the option "bwd_exact").  out / gradients fp32, dout in the input precision, lse fp32 [B*H*Sq].
mfa_error_t umfa_quantized_backward_stream(mfa_context_t context, void* stream_handle, const void* q, const void* k,
                                           const void* v, const float* out, const void* dout, const float* lse, float* dq,
                                           float* dk, float* dv, uint32_t* status, uint32_t batch_size, uint32_t seq_len_q,
                                           uint32_t seq_len_kv, uint32_t num_heads, uint16_t head_dim, float softmax_scale,
                                           bool causal, int32_t target_precision, int32_t quant_mode, int32_t input_precision) {
    Context* ctx = as_ctx(context);
    if (!ctx || !q || !k || !v || !out || !dout || !lse || !dq || !dk || !dv) return MFA_ERROR_INVALID_ARGS;
    const uint32_t B = batch_size, H = num_heads, Sq = seq_len_q, Skv = seq_len_kv, D = head_dim;
    const size_t nq = (size_t)B * H * Sq * D, nr = (size_t)B * H * Sq;
    if (nq == 0 || (size_t)B * H * Skv * D == 0) return MFA_SUCCESS;
    if (!quantized_supported(D)) return MFA_ERROR_INVALID_ARGS;
    const int prec = dense_prec(input_precision);
    const int bits = target_precision == MFA_PRECISION_INT4 ? 4 : 8;
    const int mode = quant_mode == 2 ? 2 : 0;
    hipStream_t stream = (hipStream_t)stream_handle;
    std::lock_guard<std::mutex> lock(ctx->mu);
    const int dev = stream_device(stream);
    DeviceGuard guard(dev);
    const bool fast = bwd16_shape_ok(D, false);
    const size_t wq = (quant_workspace_bytes(B, H, Sq, Skv, D, true) + 255) & ~(size_t)255;
    const size_t o_dvec = wq, o_do16 = o_dvec + ((nr * 4 + 255) & ~(size_t)255), o_rowc = o_do16 + ((nq * 2 + 255) & ~(size_t)255),
                 o_end = o_rowc + ((2 * nr * 4 + 255) & ~(size_t)255);
    char* ws = (char*)ctx->pool(dev, stream).workspace.ensure((fast ? o_end : o_do16) + 256, stream);
    if (!ws) return MFA_ERROR_MEMORY_ALLOCATION;
    // The overflow word and the units header (amax words updated by agent-scope fetch_max) sit in a block of their own at a fixed address: zeroed
    // when it is allocated, left zero by bwd_units_kernel, their last reader -- NO per-call memset node in front of them.  (As a node of a replayed
    // graph such a memset left the forward's ticket words stale in round 3, runtime_internal.h ensure_ticketed; a stale amax is max(previous,
    // current): exponents too large, operands pushed towards fp16's subnormals, and nothing left in the engine that would say so.)
    uint32_t* const qhdr = fast ? ctx->pool(dev, stream).ensure_qhdr(stream) : nullptr;
    if (fast && !qhdr) return MFA_ERROR_MEMORY_ALLOCATION;  // (under capture: warm the shape up on the capture stream first, like every pool block)
    uint32_t* flag = status ? status : qhdr;
    if (status && hipMemsetAsync(status, 0, 4, stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;  // the caller's word (kept in the signature; nothing raises it)
    uint32_t* unit = fast ? qhdr + 16 : nullptr;  // every operand as a power-of-two multiple, see mfa_quantized_backward
    const size_t nkv_in = (size_t)B * H * Skv * D;
    if (fast) {  // one launch for the four amax words (dO's is unit[0])
        const void* const srcs[4] = {q, k, v, dout};
        const int64_t ns[4] = {(int64_t)nq, (int64_t)nkv_in, (int64_t)nkv_in, (int64_t)nq};
        uint32_t* const words[4] = {unit + 4, unit + 5, unit + 6, unit};
        if (launch_amax_dense_n(4, srcs, prec, ns, words, stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
    }
    auto dirty = [&](mfa_error_t rc) { if (fast) ctx->pool(dev, stream).drop_qhdr(); return rc; };  // amax words set, bwd_units_kernel not reached
    QuantViews views;
    if (launch_quantize(q, k, v, prec, B, H, Sq, Skv, D, bits, mode, ws, fast ? 2 : 1, &views, stream, fast ? flag : nullptr, nullptr, fast ? unit + 4 : nullptr) != hipSuccess)
        return dirty(MFA_ERROR_EXECUTION_FAILED);
    BwdParams p;
    memset(&p, 0, sizeof(p));
    p.o = out; p.lse = lse; p.dq = dq; p.dk = dk; p.dv = dv;
    p.dvec = (float*)(ws + o_dvec);
    p.B = B; p.H = H; p.Sq = Sq; p.Skv = Skv; p.D = D;
    p.scale = softmax_scale; p.causal = causal ? 1 : 0;
    const char* name = "none";
    hipError_t e;
    if (fast) {
        p.q = views.qh; p.k = views.kh; p.v = views.vh;
        p.in_prec = P_FP16; p.dout_prec = P_FP16;
        if (launch_cast_f16_unit(dout, prec, ws + o_do16, (int64_t)nq, unit, stream, true) != hipSuccess || launch_bwd_units(unit, stream, qhdr) != hipSuccess)
            return dirty(MFA_ERROR_EXECUTION_FAILED);
        p.dout = ws + o_do16;
        p.units = (const float*)(unit + 8);
        p.rowc = (float*)(ws + o_rowc);
        if (!bwd_16_supported(p)) return MFA_ERROR_INVALID_ARGS;  // 16-byte alignment of the caller's tensors
        e = launch_bwd_16(p, stream, &name);  // (nothing in it can leave fp16's range: `status` stays 0)
    } else {
        p.dout = dout; p.q = views.qf; p.k = views.kf; p.v = views.vf;
        p.in_prec = P_FP32; p.dout_prec = prec;
        e = launch_bwd(p, stream, &name);
    }
    ctx->last_kernel = name;
    return e == hipSuccess ? MFA_SUCCESS : e == hipErrorInvalidValue ? MFA_ERROR_INVALID_ARGS : MFA_ERROR_EXECUTION_FAILED;
}

// ---- pre-quantised backward ABI (MFABridge.swift:1623-2163, mfa_ffi.h:480-624) ----------------------------------------
// Q, K, V arrive already quantised (or in fp16 / bf16 / fp32: the *_precision arguments are mfa_precision_t values) with
// per-tensor scale / zero point, or per-block scales when `*_block_size` > 0 and a scale buffer is given.  Block
// geometry is this library's (the same as its runtime quantiser): `block_size` consecutive rows of one (batch,
// head) slab, scales laid out [batch][head][block] -- the reference's is not recoverable from its sources (parity
// unpinned).  O, dO, LSE, D and the gradients are fp32; the softmax scale is 1/sqrt(head_dim) (the
// ABI carries none); K / V may have fewer heads than Q (grouped), their gradients are summed over the group.
// Implementation = the reference's contract ("dequantise on load, fp32 math", AGENTS.md:143-152): de-quantise into
// fp32 copies, then the fp32-exact backward; the query call produces dQ and D, the kv call consumes D.
namespace {

struct PreQuant {
    Buffer *q, *k, *v, *qs, *qz, *ks, *kz, *vs, *vz;
    uint32_t B, Sq, Skv, H, Hkv, D, qbs, kbs, vbs;
    float q_scale, k_scale, v_scale;
    int q_zp, k_zp, v_zp, qp, kp, vp;
    bool tq, tk, tv;
};

size_t quant_bytes(int prec, size_t n) {
    switch (prec) {
    case P_INT8: return n;
    case P_INT4: return (n + 1) / 2;
    case P_FP32: return n * 4;
    default: return n * 2;
    }
}
int raw_prec(int32_t v) { return (v >= 0 && v <= 4) ? (int)v : P_FP16; }  // unknown raw value -> FP16 (:1796)

// validates, uploads and de-quantises Q, K, V into ws = [Q | K (H heads) | V (H heads)], fp32 -- or, `half` (operands of the
// 16-bit MFMA backward), fp16 in the same slots with *overflow raised by values outside fp16's range
mfa_error_t prequant_stage(Context* ctx, const PreQuant& a, float** qf, float** kf, float** vf, size_t extra_bytes,
                           char** extra, hipStream_t stream, bool half = false, size_t overflow_off = 0) {
    if (a.D == 0 || a.D > 1024 || a.H == 0 || a.Hkv == 0 || a.H % a.Hkv) return MFA_ERROR_INVALID_ARGS;  // (257 ... 1024: de-quantise -> the wide fp32 backward)
    const size_t nq = (size_t)a.B * a.H * a.Sq * a.D, nkv_src = (size_t)a.B * a.Hkv * a.Skv * a.D;
    const size_t nkv = (size_t)a.B * a.H * a.Skv * a.D;
    if (!a.q->fits(quant_bytes(a.qp, nq)) || !a.k->fits(quant_bytes(a.kp, nkv_src)) || !a.v->fits(quant_bytes(a.vp, nkv_src)))
        return MFA_ERROR_INVALID_ARGS;
    auto blocks_ok = [&](Buffer* s, Buffer* z, uint32_t bs, uint32_t heads, uint32_t S) {
        if (!bs || !s) return true;
        const size_t nb = (size_t)a.B * heads * ((S + bs - 1) / bs);
        return s->fits(nb * 4) && (!z || z->fits(nb * 4));
    };
    if (!blocks_ok(a.qs, a.qz, a.qbs, a.H, a.Sq) || !blocks_ok(a.ks, a.kz, a.kbs, a.Hkv, a.Skv) ||
        !blocks_ok(a.vs, a.vz, a.vbs, a.Hkv, a.Skv))
        return MFA_ERROR_INVALID_ARGS;
    const size_t fbytes = ((nq + 2 * nkv) * 4 + 255) & ~(size_t)255;
    char* ws = (char*)ctx->pool(ctx->device, stream).workspace.ensure(fbytes + extra_bytes + 256, stream);  // synchronous entries only
    if (!ws) return MFA_ERROR_MEMORY_ALLOCATION;
    const size_t esz = half ? 2 : 4;  // the slots keep their fp32 size: one workspace plan for both attempts
    *qf = (float*)ws;
    *kf = (float*)(ws + nq * esz);
    *vf = (float*)(ws + (nq + nkv) * esz);
    *extra = ws + fbytes;
    for (Buffer* b : {a.q, a.k, a.v, a.qs, a.qz, a.ks, a.kz, a.vs, a.vz})
        if (b && b->upload(stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
    // (half: the flag word the kernels below OR into + the 16-word units header behind it, kernels.h launch_bwd_units: word 0 dO's amax, 4 ... 6 those of Q, K, V)
    // (a per-call memset is fine HERE: these entries are synchronous on the legacy stream and cannot be captured; the in-stream entry's header is self-cleaning)
    if (half && hipMemsetAsync(*extra + overflow_off, 0, 256, stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
    uint32_t* const unit = half ? (uint32_t*)(*extra + overflow_off) + 16 : nullptr;
    auto deq = [&](Buffer* src, float* dst, int prec, uint32_t hs, uint32_t S, float sc, int zp, Buffer* bsc, Buffer* bzp,
                   uint32_t bs, bool t, int which) {
        DequantParams d;
        memset(&d, 0, sizeof(d));
        d.src = src->dev; d.dst = half ? nullptr : dst;
        d.dst16 = half ? (void*)dst : nullptr;
        d.overflow = half ? (uint32_t*)(*extra + overflow_off) : nullptr;  // a word inside the caller's extra region
        d.block_scales = (bs && bsc) ? (const float*)bsc->dev : nullptr;
        d.block_zero_points = (bs && bsc && bzp) ? (const int32_t*)bzp->dev : nullptr;
        d.B = a.B; d.H_src = hs; d.H_dst = a.H; d.S = S; d.D = a.D; d.block_size = bs;
        d.scale = sc; d.zero_point = zp; d.prec = prec; d.transposed = t ? 1 : 0;
        if (half) {
            // fp16 images as power-of-two multiples with the tensor's largest magnitude in [1, 2) (BwdParams::units, as in mfa_quantized_backward):
            // one launch for the amax of the de-quantised tensor, one that stores x * 2^-e -- caller-side scales of 1e-9 or 1e9 make no difference
            d.amax_word = unit + 4 + which;
            if (hipError_t e = launch_dequant(d, stream); e != hipSuccess) return e;
            d.amax_word = nullptr;
            d.unit_amax = unit + 4 + which;
        }
        return launch_dequant(d, stream);
    };
    if (deq(a.q, *qf, a.qp, a.H, a.Sq, a.q_scale, a.q_zp, a.qs, a.qz, a.qbs, a.tq, 0) != hipSuccess ||
        deq(a.k, *kf, a.kp, a.Hkv, a.Skv, a.k_scale, a.k_zp, a.ks, a.kz, a.kbs, a.tk, 1) != hipSuccess ||
        deq(a.v, *vf, a.vp, a.Hkv, a.Skv, a.v_scale, a.v_zp, a.vs, a.vz, a.vbs, a.tv, 2) != hipSuccess)
        return MFA_ERROR_EXECUTION_FAILED;
    return MFA_SUCCESS;
}

}  // namespace

int32_t mfa_attention_backward_query_quantized_ex(
    mfa_context_t context, mfa_buffer_t q, mfa_buffer_t k, mfa_buffer_t v, mfa_buffer_t output, mfa_buffer_t grad_output,
    mfa_buffer_t logsumexp, mfa_buffer_t grad_query, mfa_buffer_t d_values, uint32_t batch_size, uint32_t seq_len_q,
    uint32_t seq_len_kv, uint32_t num_heads, uint32_t num_kv_heads, uint16_t head_dim, float q_scale, int32_t q_zero_point,
    float k_scale, int32_t k_zero_point, float v_scale, int32_t v_zero_point, int32_t q_precision, int32_t k_precision,
    int32_t v_precision, bool causal, bool transpose_q, bool transpose_k, bool transpose_v, bool transpose_o,
    mfa_buffer_t q_block_scales, mfa_buffer_t q_block_zero_points, mfa_buffer_t k_block_scales,
    mfa_buffer_t k_block_zero_points, mfa_buffer_t v_block_scales, mfa_buffer_t v_block_zero_points, uint32_t q_block_size,
    uint32_t k_block_size, uint32_t v_block_size, uint32_t options) {
    (void)options;  // ignored by the reference too (MFABridge.swift:1755)
    Context* ctx = as_ctx(context);
    Buffer *bo = as_buf(output), *bdo = as_buf(grad_output), *bl = as_buf(logsumexp), *bdq = as_buf(grad_query),
           *bd = as_buf(d_values);
    PreQuant a{as_buf(q), as_buf(k), as_buf(v), as_buf(q_block_scales), as_buf(q_block_zero_points), as_buf(k_block_scales),
               as_buf(k_block_zero_points), as_buf(v_block_scales), as_buf(v_block_zero_points), batch_size, seq_len_q,
               seq_len_kv, num_heads ? num_heads : 1u, num_kv_heads ? num_kv_heads : (num_heads ? num_heads : 1u), head_dim,
               q_block_size, k_block_size, v_block_size, q_scale, k_scale, v_scale, q_zero_point, k_zero_point, v_zero_point,
               raw_prec(q_precision), raw_prec(k_precision), raw_prec(v_precision), transpose_q, transpose_k, transpose_v};
    if (!ctx || !a.q || !a.k || !a.v || !bo || !bdo || !bl || !bdq || !bd) return MFA_ERROR_INVALID_ARGS;
    if (transpose_o) return MFA_ERROR_INVALID_ARGS;  // O / dO are read dense (no caller transposes them)
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    hipStream_t stream = nullptr;
    const size_t nq = (size_t)a.B * a.H * a.Sq * a.D, nr = (size_t)a.B * a.H * a.Sq;
    if (!bo->fits(nq * 4) || !bdo->fits(nq * 4) || !bl->fits(nr * 4) || !bdq->fits(nq * 4) || !bd->fits(nr * 4))
        return MFA_ERROR_INVALID_ARGS;
    if (nq == 0 || a.Skv == 0) return MFA_SUCCESS;
    for (Buffer* b : {bo, bdo, bl})
        if (b->upload(stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
    // FAST: operands de-quantised to fp16 and dO cast to fp16 -- each as a power-of-two multiple with its largest magnitude in [1, 2) (prequant_stage,
    // BwdParams::units) --, the dQ kernel of the 16-bit MFMA backward (it also leaves D, in true units).  EXACT (fp32) for what that engine does not take.
    // See mfa_quantized_backward.
    const bool try_fast = bwd16_shape_ok(a.D, false);
    for (int attempt = try_fast ? 0 : 1; attempt < 2; ++attempt) {
        const bool fast = attempt == 0;
        const size_t o_do16 = 0, o_rowc = (nq * 2 + 255) & ~(size_t)255, o_flag = o_rowc + ((2 * nr * 4 + 255) & ~(size_t)255);
        float *qf, *kf, *vf;
        char* extra;
        LatencyScope lat(ctx, stream);
        mfa_error_t st = prequant_stage(ctx, a, &qf, &kf, &vf, fast ? o_flag + 256 : 0, &extra, stream, fast, o_flag);
        if (st != MFA_SUCCESS) return st;
        BwdParams p;
        memset(&p, 0, sizeof(p));
        p.dout = bdo->dev; p.q = qf; p.k = kf; p.v = vf;
        p.o = (const float*)bo->dev; p.lse = (const float*)bl->dev;
        p.dq = (float*)bdq->dev; p.dvec = (float*)bd->dev;
        p.B = a.B; p.H = a.H; p.Sq = a.Sq; p.Skv = a.Skv; p.D = a.D;
        p.scale = 1.0f / sqrtf((float)a.D); p.causal = causal ? 1 : 0;
        p.in_prec = P_FP32; p.dout_prec = P_FP32;
        p.phases = 1 | 2;  // D vector + dQ
        const char* name = "none";
        hipError_t e;
        if (fast) {
            p.in_prec = P_FP16; p.dout_prec = P_FP16;
            // every operand a power-of-two multiple (prequant_stage took Q, K, V; dO here: the same exponents in the query and the kv call -- the same
            // tensors), see mfa_quantized_backward
            uint32_t* unit = (uint32_t*)(extra + o_flag) + 16;
            if (launch_cast_f16_unit(bdo->dev, P_FP32, extra + o_do16, (int64_t)nq, unit, stream) != hipSuccess || launch_bwd_units(unit, stream) != hipSuccess)
                return MFA_ERROR_EXECUTION_FAILED;
            p.dout = extra + o_do16;
            p.units = (const float*)(unit + 8);
            p.rowc = (float*)(extra + o_rowc);
            e = bwd_16_supported(p) ? launch_bwd_16(p, stream, &name) : hipErrorNotSupported;
            if (e == hipErrorNotSupported) continue;
        } else {
            e = launch_bwd(p, stream, &name);
        }
        ctx->last_kernel = name;
        if (e != hipSuccess) return e == hipErrorInvalidValue ? MFA_ERROR_INVALID_ARGS : MFA_ERROR_EXECUTION_FAILED;
        lat.stop();
        if (bdq->download(stream) != hipSuccess || bd->download(stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
        if (hipStreamSynchronize(stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
        lat.publish();
        break;  // (the second attempt exists for what the fp16 engine does not take -- `continue` above; nothing in it can leave fp16's range: no flag to read)
    }
    return MFA_SUCCESS;
}

int32_t mfa_attention_backward_kv_quantized_ex(
    mfa_context_t context, mfa_buffer_t q, mfa_buffer_t k, mfa_buffer_t v, mfa_buffer_t grad_output, mfa_buffer_t logsumexp,
    mfa_buffer_t d_values, mfa_buffer_t grad_key, mfa_buffer_t grad_value, uint32_t batch_size, uint32_t seq_len_q,
    uint32_t seq_len_kv, uint32_t num_heads, uint32_t num_kv_heads, uint16_t head_dim, float q_scale, int32_t q_zero_point,
    float k_scale, int32_t k_zero_point, float v_scale, int32_t v_zero_point, int32_t q_precision, int32_t k_precision,
    int32_t v_precision, bool causal, bool transpose_q, bool transpose_k, bool transpose_v, bool transpose_o,
    mfa_buffer_t q_block_scales, mfa_buffer_t q_block_zero_points, mfa_buffer_t k_block_scales,
    mfa_buffer_t k_block_zero_points, mfa_buffer_t v_block_scales, mfa_buffer_t v_block_zero_points, uint32_t q_block_size,
    uint32_t k_block_size, uint32_t v_block_size, uint32_t options) {
    (void)options;
    Context* ctx = as_ctx(context);
    Buffer *bdo = as_buf(grad_output), *bl = as_buf(logsumexp), *bd = as_buf(d_values), *bdk = as_buf(grad_key),
           *bdv = as_buf(grad_value);
    PreQuant a{as_buf(q), as_buf(k), as_buf(v), as_buf(q_block_scales), as_buf(q_block_zero_points), as_buf(k_block_scales),
               as_buf(k_block_zero_points), as_buf(v_block_scales), as_buf(v_block_zero_points), batch_size, seq_len_q,
               seq_len_kv, num_heads ? num_heads : 1u, num_kv_heads ? num_kv_heads : (num_heads ? num_heads : 1u), head_dim,
               q_block_size, k_block_size, v_block_size, q_scale, k_scale, v_scale, q_zero_point, k_zero_point, v_zero_point,
               raw_prec(q_precision), raw_prec(k_precision), raw_prec(v_precision), transpose_q, transpose_k, transpose_v};
    if (!ctx || !a.q || !a.k || !a.v || !bdo || !bl || !bd || !bdk || !bdv) return MFA_ERROR_INVALID_ARGS;
    if (transpose_o) return MFA_ERROR_INVALID_ARGS;
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    hipStream_t stream = nullptr;
    const size_t nq = (size_t)a.B * a.H * a.Sq * a.D, nr = (size_t)a.B * a.H * a.Sq;
    const size_t nkv_out = (size_t)a.B * a.Hkv * a.Skv * a.D, nkv = (size_t)a.B * a.H * a.Skv * a.D;
    if (!bdo->fits(nq * 4) || !bl->fits(nr * 4) || !bd->fits(nr * 4) || !bdk->fits(nkv_out * 4) || !bdv->fits(nkv_out * 4))
        return MFA_ERROR_INVALID_ARGS;
    if (nq == 0 || nkv_out == 0) return MFA_SUCCESS;
    const bool grouped = a.Hkv != a.H;
    for (Buffer* b : {bdo, bl, bd})
        if (b->upload(stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
    const bool try_fast = bwd16_shape_ok(a.D, false);  // FAST / EXACT as in the query entry; the row constants come from (LSE, D)
    for (int attempt = try_fast ? 0 : 1; attempt < 2; ++attempt) {
        const bool fast = attempt == 0;
        const size_t o_grp = 0, o_do16 = grouped ? ((2 * nkv * 4 + 255) & ~(size_t)255) : 0, o_rowc = o_do16 + ((nq * 2 + 255) & ~(size_t)255),
                     o_flag = o_rowc + ((2 * nr * 4 + 255) & ~(size_t)255);
        float *qf, *kf, *vf;
        char* extra;
        LatencyScope lat(ctx, stream);
        mfa_error_t st = prequant_stage(ctx, a, &qf, &kf, &vf, fast ? o_flag + 256 : o_do16, &extra, stream, fast, o_flag);
        if (st != MFA_SUCCESS) return st;
        BwdParams p;
        memset(&p, 0, sizeof(p));
        p.dout = bdo->dev; p.q = qf; p.k = kf; p.v = vf;
        p.lse = (const float*)bl->dev; p.dvec = (float*)bd->dev;
        p.dk = grouped ? (float*)(extra + o_grp) : (float*)bdk->dev;
        p.dv = grouped ? (float*)(extra + o_grp) + nkv : (float*)bdv->dev;
        p.B = a.B; p.H = a.H; p.Sq = a.Sq; p.Skv = a.Skv; p.D = a.D;
        p.scale = 1.0f / sqrtf((float)a.D); p.causal = causal ? 1 : 0;
        p.in_prec = P_FP32; p.dout_prec = P_FP32;
        p.phases = 4;  // dK / dV from the caller's D vector
        const char* name = "none";
        hipError_t e;
        if (fast) {
            p.in_prec = P_FP16; p.dout_prec = P_FP16;
            // every operand a power-of-two multiple (prequant_stage took Q, K, V; dO here: the same exponents in the query and the kv call -- the same
            // tensors), see mfa_quantized_backward
            uint32_t* unit = (uint32_t*)(extra + o_flag) + 16;
            if (launch_cast_f16_unit(bdo->dev, P_FP32, extra + o_do16, (int64_t)nq, unit, stream) != hipSuccess || launch_bwd_units(unit, stream) != hipSuccess)
                return MFA_ERROR_EXECUTION_FAILED;
            p.dout = extra + o_do16;
            p.units = (const float*)(unit + 8);
            p.rowc = (float*)(extra + o_rowc);
            e = bwd_16_supported(p) ? launch_bwd_16(p, stream, &name) : hipErrorNotSupported;
            if (e == hipErrorNotSupported) continue;
        } else {
            e = launch_bwd(p, stream, &name);
        }
        ctx->last_kernel = name;
        if (e != hipSuccess) return e == hipErrorInvalidValue ? MFA_ERROR_INVALID_ARGS : MFA_ERROR_EXECUTION_FAILED;
        if (grouped) {
            const int64_t slab = (int64_t)a.Skv * a.D;
            if (launch_group_sum(p.dk, (float*)bdk->dev, a.B, a.H, a.Hkv, slab, stream) != hipSuccess ||
                launch_group_sum(p.dv, (float*)bdv->dev, a.B, a.H, a.Hkv, slab, stream) != hipSuccess)
                return MFA_ERROR_EXECUTION_FAILED;
        }
        lat.stop();
        if (bdk->download(stream) != hipSuccess || bdv->download(stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
        if (hipStreamSynchronize(stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
        lat.publish();
        break;  // (as in the query entry)
    }
    return MFA_SUCCESS;
}

// the older entries = the _ex ones with equal head counts and no block tables (MFABridge.swift:1623-1697, 1894-1968)
int32_t mfa_attention_backward_query_quantized(
    mfa_context_t context, mfa_buffer_t q, mfa_buffer_t k, mfa_buffer_t v, mfa_buffer_t output, mfa_buffer_t grad_output,
    mfa_buffer_t logsumexp, mfa_buffer_t grad_query, mfa_buffer_t d_values, uint32_t batch_size, uint32_t seq_len_q,
    uint32_t seq_len_kv, uint32_t num_heads, uint16_t head_dim, float q_scale, int32_t q_zero_point, float k_scale,
    int32_t k_zero_point, float v_scale, int32_t v_zero_point, int32_t q_precision, int32_t k_precision, int32_t v_precision,
    bool causal, bool transpose_q, bool transpose_k, bool transpose_v, bool transpose_o) {
    return mfa_attention_backward_query_quantized_ex(
        context, q, k, v, output, grad_output, logsumexp, grad_query, d_values, batch_size, seq_len_q, seq_len_kv, num_heads,
        num_heads, head_dim, q_scale, q_zero_point, k_scale, k_zero_point, v_scale, v_zero_point, q_precision, k_precision,
        v_precision, causal, transpose_q, transpose_k, transpose_v, transpose_o, nullptr, nullptr, nullptr, nullptr, nullptr,
        nullptr, 0, 0, 0, 0);
}

int32_t mfa_attention_backward_kv_quantized(
    mfa_context_t context, mfa_buffer_t q, mfa_buffer_t k, mfa_buffer_t v, mfa_buffer_t grad_output, mfa_buffer_t logsumexp,
    mfa_buffer_t d_values, mfa_buffer_t grad_key, mfa_buffer_t grad_value, uint32_t batch_size, uint32_t seq_len_q,
    uint32_t seq_len_kv, uint32_t num_heads, uint16_t head_dim, float q_scale, int32_t q_zero_point, float k_scale,
    int32_t k_zero_point, float v_scale, int32_t v_zero_point, int32_t q_precision, int32_t k_precision, int32_t v_precision,
    bool causal, bool transpose_q, bool transpose_k, bool transpose_v, bool transpose_o) {
    return mfa_attention_backward_kv_quantized_ex(
        context, q, k, v, grad_output, logsumexp, d_values, grad_key, grad_value, batch_size, seq_len_q, seq_len_kv, num_heads,
        num_heads, head_dim, q_scale, q_zero_point, k_scale, k_zero_point, v_scale, v_zero_point, q_precision, k_precision,
        v_precision, causal, transpose_q, transpose_k, transpose_v, transpose_o, nullptr, nullptr, nullptr, nullptr, nullptr,
        nullptr, 0, 0, 0, 0);
}

// MI355X extra (not in the reference ABI): run the runtime quantiser on one device tensor [BH, S, D] and hand back its
// int8 image (rows padded to 64 / 128 / 256 bytes) and the per-64-row-block scales -- the integer half of the quantised
// path, exposed so that it can be checked bit-for-bit against the oracle (tests/test_gpu_quantized.py).
int32_t umfa_quantize_rows(mfa_context_t context, void* stream_handle, const void* src, int32_t input_precision,
                           uint32_t batch_heads, uint32_t rows, uint32_t head_dim, int32_t bits, int32_t quant_mode,
                           void* q8_out, void* scales_out, uint32_t* padded_row_bytes) {
    Context* ctx = as_ctx(context);
    if (!ctx || !src || !q8_out || !scales_out || !quantized_supported(head_dim) || head_dim > 256) return MFA_ERROR_INVALID_ARGS;  // (int8 rows: the register kernels' head dims)
    std::lock_guard<std::mutex> lock(ctx->mu);
    hipStream_t stream = (hipStream_t)stream_handle;
    const int pool_dev = stream_device(stream);
    DeviceGuard guard(pool_dev);
    void* ws = ctx->pool(pool_dev, stream).workspace.ensure(quant_workspace_bytes(1, batch_heads, rows, rows, head_dim, false), stream);
    if (!ws) return MFA_ERROR_MEMORY_ALLOCATION;
    QuantViews v;
    if (launch_quantize(src, src, src, dense_prec(input_precision), 1, batch_heads, rows, rows, head_dim, bits == 4 ? 4 : 8,
                        quant_mode == 2 ? 2 : 0, ws, false, &v, stream) != hipSuccess)
        return MFA_ERROR_EXECUTION_FAILED;
    if (hipMemcpyAsync(q8_out, v.q8, (size_t)batch_heads * rows * v.dpq, hipMemcpyDeviceToDevice, stream) != hipSuccess ||
        hipMemcpyAsync(scales_out, v.q_scale, (size_t)batch_heads * v.nqblk * sizeof(float), hipMemcpyDeviceToDevice, stream) !=
            hipSuccess)
        return MFA_ERROR_EXECUTION_FAILED;
    if (padded_row_bytes) *padded_row_bytes = v.dpq;
    return MFA_SUCCESS;
}

}  // extern "C"
