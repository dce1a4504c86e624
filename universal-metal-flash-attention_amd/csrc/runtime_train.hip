// runtime_train.hip -- backward and runtime-quantised entry points of the C ABI.
//
// mfa_attention_backward           MFABridge.swift:3171-3282
// mfa_quantized_forward_with_lse   MFABridge+Quantized.swift:227-358
// mfa_quantized_backward           MFABridge+Quantized.swift:365-533
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
    (void)hipSetDevice(ctx->device);
    hipStream_t stream = nullptr;
    const uint32_t B = batch_size, H = num_heads, Sq = seq_len_q, Skv = seq_len_kv, D = head_dim;
    const size_t nq = (size_t)B * H * Sq * D, nkv = (size_t)B * H * Skv * D, nr = (size_t)B * H * Sq;
    const int prec = dense_prec(input_precision);
    const size_t eb = elem_bytes(prec);
    if (!bdo->fits(nq * eb) || !bq->fits(nq * eb) || !bk->fits(nkv * eb) || !bv->fits(nkv * eb) || !bo->fits(nq * 4) ||
        !bl->fits(nr * 4) || !bdq->fits(nq * 4) || !bdk->fits(nkv * 4) || !bdv->fits(nkv * 4) || !bd->fits(nr * 4))
        return MFA_ERROR_INVALID_ARGS;
    if (nq == 0 || nkv == 0) return MFA_SUCCESS;
    if (D > 128) return MFA_ERROR_INVALID_ARGS;  // backward is built for head_dim <= 128 this round

    for (Buffer* b : {bdo, bq, bk, bv, bo, bl})
        if (b->upload(stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
    BwdParams p;
    memset(&p, 0, sizeof(p));
    p.dout = bdo->dev; p.q = bq->dev; p.k = bk->dev; p.v = bv->dev;
    p.o = (const float*)bo->dev; p.lse = (const float*)bl->dev;
    p.dq = (float*)bdq->dev; p.dk = (float*)bdk->dev; p.dv = (float*)bdv->dev; p.dvec = (float*)bd->dev;
    p.B = B; p.H = H; p.Sq = Sq; p.Skv = Skv; p.D = D;
    p.scale = softmax_scale; p.causal = causal ? 1 : 0;
    p.in_prec = prec; p.dout_prec = prec;
    LatencyScope lat(ctx, stream);
    const char* name = "none";
    // 16-bit operands with 16-bit intermediates -> MFMA backward; everything else -> fp32-exact backward
    const bool lowp = dense_prec(intermediate_precision) != P_FP32 && !getenv("UMFA_BWD_EXACT");
    hipError_t e = (lowp && bwd_16_supported(p)) ? launch_bwd_16(p, stream, &name) : launch_bwd(p, stream, &name);
    ctx->last_kernel = name;
    if (e != hipSuccess) return e == hipErrorInvalidValue ? MFA_ERROR_INVALID_ARGS : MFA_ERROR_EXECUTION_FAILED;
    lat.stop();
    for (Buffer* b : {bdq, bdk, bdv, bd})
        if (b->download(stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
    if (hipStreamSynchronize(stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
    lat.publish();
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
    (void)hipSetDevice(ctx->device);
    hipStream_t stream = nullptr;
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
    const int mode = quant_mode == 2 ? 2 : 0;                        // default tensor-wise (:268-272)

    void* ws = ctx->ensure_workspace(quant_workspace_bytes(B, H, Sq, Skv, D, false));
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
    (void)hipSetDevice(ctx->device);
    hipStream_t stream = nullptr;
    const uint32_t B = batch_size, H = num_heads, Sq = seq_len_q, Skv = seq_len_kv, D = head_dim;
    const size_t nq = (size_t)B * H * Sq * D, nkv = (size_t)B * H * Skv * D, nr = (size_t)B * H * Sq;
    const int prec = dense_prec(input_precision);
    const size_t eb = elem_bytes(prec);
    if (!bq->fits(nq * eb) || !bk->fits(nkv * eb) || !bv->fits(nkv * eb) || !bo->fits(nq * 4) || !bdo->fits(nq * eb) ||
        !bl->fits(nr * 4) || !bdq->fits(nq * 4) || !bdk->fits(nkv * 4) || !bdv->fits(nkv * 4))
        return MFA_ERROR_INVALID_ARGS;
    if (bm && !bm->fits(nr * Skv * 4)) return MFA_ERROR_INVALID_ARGS;
    if (nq == 0 || nkv == 0) return MFA_SUCCESS;
    if (!quantized_supported(D) || D > 128) return MFA_ERROR_INVALID_ARGS;
    const int bits = target_precision == MFA_PRECISION_INT4 ? 4 : 8;
    const int mode = quant_mode == 2 ? 2 : 0;

    // workspace: quantiser output + fp32 copies + the D vector the callee owns (MFABridge+Quantized.swift:470-474)
    const size_t wq = quant_workspace_bytes(B, H, Sq, Skv, D, true);
    char* ws = (char*)ctx->ensure_workspace(wq + nr * 4 + 256);
    if (!ws) return MFA_ERROR_MEMORY_ALLOCATION;
    for (Buffer* b : {bq, bk, bv, bo, bdo, bl})
        if (b->upload(stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
    if (bm && bm->upload(stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
    LatencyScope lat(ctx, stream);
    QuantViews views;
    hipError_t e = launch_quantize(bq->dev, bk->dev, bv->dev, prec, B, H, Sq, Skv, D, bits, mode, ws, true, &views, stream);
    if (e != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
    BwdParams p;
    memset(&p, 0, sizeof(p));
    p.dout = bdo->dev; p.q = views.qf; p.k = views.kf; p.v = views.vf;
    p.o = (const float*)bo->dev; p.lse = (const float*)bl->dev;
    p.dq = (float*)bdq->dev; p.dk = (float*)bdk->dev; p.dv = (float*)bdv->dev;
    p.dvec = (float*)(ws + ((wq + 255) & ~(size_t)255));
    p.mask = bm ? (const float*)bm->dev : nullptr;
    p.B = B; p.H = H; p.Sq = Sq; p.Skv = Skv; p.D = D;
    p.scale = softmax_scale; p.causal = causal ? 1 : 0;
    p.in_prec = P_FP32; p.dout_prec = prec;
    const char* name = "none";
    e = launch_bwd(p, stream, &name);
    ctx->last_kernel = name;
    if (e != hipSuccess) return e == hipErrorInvalidValue ? MFA_ERROR_INVALID_ARGS : MFA_ERROR_EXECUTION_FAILED;
    lat.stop();
    for (Buffer* b : {bdq, bdk, bdv})
        if (b->download(stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
    if (hipStreamSynchronize(stream) != hipSuccess) return MFA_ERROR_EXECUTION_FAILED;
    lat.publish();
    return MFA_SUCCESS;
}

}  // extern "C"
