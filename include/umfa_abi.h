/*
 * umfa_abi.h -- the C ABI of libMFAFFI.so for AMD Instinct MI355X (gfx950).
 *
 * Drop-in boundary: every entry point below keeps the name, argument order,
 * argument widths and return convention of the symbol the reference exports
 * from its Swift bridge, so the existing ctypes / bindgen / C++ callers bind
 * unchanged.  Citations are to the reference tree
 * (bghira/universal-metal-flash-attention):
 *   [H]  Sources/MFAFFI/include/mfa_ffi.h            (29 declared symbols)
 *   [B]  Sources/MFABridge/MFABridge.swift           (@_cdecl bodies)
 *   [Q]  Sources/MFABridge/MFABridge+Quantized.swift
 *   [L]  Sources/MFABridge/QuantizedLayoutManifest+FFI.swift
 *   [T]  examples/pytorch-custom-op-ffi/include/metal_sdpa_backend.h (13 undeclared symbols)
 *   [M]  examples/pytorch-custom-op-ffi/src/mps_utils.mm
 *
 * ROCm reading of the opaque pointers (ABI-preserving):
 *   "command_buffer"              -> hipStream_t the work is enqueued on
 *   "*_buffer" in *_encode_mtl    -> raw device pointer
 *   "metal_buffer" in *_from_mtl_buffer* -> raw device pointer (borrowed)
 *   data_ptr in mfa_buffer_from_ptr*     -> host OR device pointer (detected);
 *       host memory is staged to HBM before and copied back after each
 *       synchronous op so results are visible in the caller's memory on return.
 *
 * Tensor layout is contiguous [batch, heads, seq, head_dim] ("BHSD",
 * metal_sdpa_backend.cpp:188-193); attention outputs and gradients are always
 * fp32 ([B] 1074-1433 ignores output_precision; MultiHeadFFITests.swift:171-173).
 * The library never throws across the boundary and never falls back to the CPU:
 * without a usable gfx950 device mfa_create_context returns 3.
 */
#ifndef UMFA_ABI_H
#define UMFA_ABI_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: the 54 entry points declared between this push and its pop are the ONLY
 * dynamic symbols of libMFAFFI.so (tests/test_abi_symbols.py checks `nm -D`).  Harmless for callers. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

/* ---- status codes, [H]:17-26 -------------------------------------------- */
typedef int mfa_error_t;
enum {
    MFA_SUCCESS = 0,
    MFA_ERROR_INVALID_ARGS = 1,
    MFA_ERROR_MEMORY_ALLOCATION = 2,
    MFA_ERROR_DEVICE_NOT_SUPPORTED = 3,
    MFA_ERROR_KERNEL_COMPILATION = 4,
    MFA_ERROR_EXECUTION_FAILED = 5
};

/* ---- element encodings, [H]:33-41 ---------------------------------------- */
typedef int mfa_precision_t;
enum {
    MFA_PRECISION_FP16 = 0,
    MFA_PRECISION_BF16 = 1,
    MFA_PRECISION_FP32 = 2,
    MFA_PRECISION_INT8 = 3,
    MFA_PRECISION_INT4 = 4
};

/* ---- attention masks, [H]:46-64; semantics [B]:157-242 -------------------- */
typedef int mfa_mask_type_t;
enum { MFA_MASK_TYPE_NONE = 0, MFA_MASK_TYPE_BOOL = 1, MFA_MASK_TYPE_ADDITIVE = 2 };
typedef int mfa_mask_scalar_t;
enum {
    MFA_MASK_SCALAR_BYTE = 0,
    MFA_MASK_SCALAR_FP16 = 1,
    MFA_MASK_SCALAR_BF16 = 2,
    MFA_MASK_SCALAR_FP32 = 3
};

/* ---- opaque handles, [H]:69,74,633 ---------------------------------------- */
typedef void* mfa_context_t;
typedef void* mfa_buffer_t;
typedef void* mfa_mla_context_t;

/* ---- quantised-kernel introspection, [H]:76-135, [L]:1-156 ---------------- */
typedef enum {
    MFA_QUANT_KERNEL_FORWARD = 0,
    MFA_QUANT_KERNEL_BACKWARD_QUERY = 1,
    MFA_QUANT_KERNEL_BACKWARD_KEY_VALUE = 2
} mfa_quantized_kernel_t;

/* 38 x int32 binding slots; the reference fills every slot with -1 ([L]:47-49). */
typedef struct {
    int32_t qData, kData, vData, output, gradOutput, logsumexp, gradQuery, dValues, gradKey,
        gradValue;
    int32_t qScale, qZeroPoint, kScale, kZeroPoint, vScale, vZeroPoint;
    int32_t dims, steClipRange;
    int32_t qBlockScales, qBlockZeroPoints, kBlockScales, kBlockZeroPoints, vBlockScales,
        vBlockZeroPoints;
    int32_t qPrecomputedSums, kPrecomputedSums, vPrecomputedSums;
    int32_t qStrides, kStrides, vStrides, oStrides;
    int32_t maskBuffer, numHeads, numKeyValueHeads, headDimension, sequenceLength;
    int32_t scratch0, scratch1;
} mfa_quantized_layout_t;

typedef struct {
    bool supports_multi_head_backward;
    bool supports_blockwise_backward;
    uint32_t max_heads;
    uint32_t max_block_size;
} mfa_quantized_capabilities_t;

void mfa_get_quantized_layout(mfa_quantized_kernel_t kernel, mfa_quantized_layout_t* out_layout); /* [H]:123 */
void mfa_get_quantized_capabilities(void* out_capabilities); /* [H]:135 -> {1,1,128,256}, [L]:126-131 */

/* ---- context: process-wide singleton, +1 per create, [B]:782-805 ---------- */
mfa_error_t mfa_create_context(mfa_context_t* context);   /* [H]:147 */
void mfa_destroy_context(mfa_context_t context);          /* [H]:154 */

/* ---- buffers, [B]:850-1070.  Wrapped memory is never freed by the library. - */
mfa_error_t mfa_create_buffer(mfa_context_t context, size_t size_bytes, mfa_buffer_t* buffer); /* [H]:168 */
mfa_error_t mfa_buffer_from_ptr(mfa_context_t context, void* data_ptr, size_t size_bytes,
                                mfa_buffer_t* buffer); /* [H]:183 */
mfa_error_t mfa_buffer_from_ptr_with_strides(mfa_context_t context, void* data_ptr,
                                             size_t size_bytes, const int64_t* shape,
                                             const int64_t* strides, uint32_t ndim,
                                             mfa_buffer_t* buffer); /* [H]:193 */
mfa_error_t mfa_buffer_from_mtl_buffer(mfa_context_t context, void* metal_buffer,
                                       size_t size_bytes, mfa_buffer_t* buffer); /* [H]:206 */
mfa_error_t mfa_buffer_from_mtl_buffer_with_strides(mfa_context_t context, void* metal_buffer,
                                                    size_t size_bytes, const int64_t* shape,
                                                    const int64_t* strides, uint32_t ndim,
                                                    mfa_buffer_t* buffer); /* [H]:216 */
void* mfa_buffer_contents(mfa_buffer_t buffer); /* [H]:232: host-visible pointer */
void mfa_destroy_buffer(mfa_buffer_t buffer);   /* [H]:239 */

/* ---- dense forward, synchronous, [H]:273-300, [B]:1074-1433 ---------------- */
mfa_error_t mfa_attention_forward(
    mfa_context_t context, mfa_buffer_t q, mfa_buffer_t k, mfa_buffer_t v, mfa_buffer_t out,
    uint32_t batch_size, uint32_t seq_len_q, uint32_t seq_len_kv, uint32_t num_heads,
    uint16_t head_dim, float softmax_scale, bool causal, mfa_precision_t input_precision,
    mfa_precision_t intermediate_precision, mfa_precision_t output_precision, bool transpose_q,
    bool transpose_k, bool transpose_v, bool transpose_o, const void* mask_ptr,
    size_t mask_size_bytes, const int64_t* mask_shape, const int64_t* mask_strides,
    uint32_t mask_ndim, mfa_mask_type_t mask_type, mfa_mask_scalar_t mask_scalar_type);

/* Same, precisions as strings ("fp16","float16","bf16","bfloat16","fp32","float32","int8","int4";
 * NULL/unknown -> fp32).  [T]:311-327, [B]:1438-1522. */
mfa_error_t mfa_attention_forward_str(
    mfa_context_t context, mfa_buffer_t q, mfa_buffer_t k, mfa_buffer_t v, mfa_buffer_t out,
    uint32_t batch_size, uint32_t seq_len_q, uint32_t seq_len_kv, uint32_t num_heads,
    uint16_t head_dim, float softmax_scale, bool causal, const char* input_precision,
    const char* intermediate_precision, const char* output_precision, bool transpose_q,
    bool transpose_k, bool transpose_v, bool transpose_o, const void* mask_ptr,
    size_t mask_size_bytes, const int64_t* mask_shape, const int64_t* mask_strides,
    uint32_t mask_ndim, mfa_mask_type_t mask_type, mfa_mask_scalar_t mask_scalar_type);

/* ---- dense forward, asynchronous on the caller's stream, [H]:312-334, [B]:2377-2543.
 * command_buffer = hipStream_t; *_buffer = device pointers; offsets in BYTES;
 * q/k/v_strides = 4 x int64 ELEMENT strides in BHSD order (last must be 1) or NULL = dense;
 * out is dense fp32 [B,H,Sq,D].  Never synchronises. */
mfa_error_t mfa_attention_encode_mtl(
    mfa_context_t context, void* command_buffer, void* q_buffer, int64_t q_offset,
    const int64_t* q_strides, void* k_buffer, int64_t k_offset, const int64_t* k_strides,
    void* v_buffer, int64_t v_offset, const int64_t* v_strides, void* out_buffer,
    int64_t out_offset, void* mask_buffer, int64_t mask_offset, const int64_t* mask_shape,
    const int64_t* mask_strides, uint32_t mask_ndim, mfa_mask_type_t mask_type,
    mfa_mask_scalar_t mask_scalar_type, uint32_t batch_size, uint32_t seq_len_q,
    uint32_t seq_len_kv, uint32_t num_heads, uint16_t head_dim, float softmax_scale, bool causal,
    const char* input_precision, const char* intermediate_precision);

/* ---- forward + log-sum-exp (fp32 [B*H*Sq], natural log), [T]:471-480, [B]:3078-3166 */
int32_t mfa_attention_forward_with_lse(
    mfa_context_t context, mfa_buffer_t q, mfa_buffer_t k, mfa_buffer_t v, mfa_buffer_t out,
    mfa_buffer_t lse, uint32_t batch_size, uint32_t seq_len_q, uint32_t seq_len_kv,
    uint32_t num_heads, uint16_t head_dim, float softmax_scale, bool causal,
    int32_t input_precision, int32_t intermediate_precision, bool transpose_q, bool transpose_k,
    bool transpose_v, bool transpose_o);

/* ---- dense backward, [H]:407-438, [B]:3171-3282.  dq/dk/dv fp32; d_buffer fp32 [B*H*Sq]. */
mfa_error_t mfa_attention_backward(
    mfa_context_t context, mfa_buffer_t dout, mfa_buffer_t q, mfa_buffer_t k, mfa_buffer_t v,
    mfa_buffer_t out, mfa_buffer_t softmax_lse, mfa_buffer_t dq, mfa_buffer_t dk, mfa_buffer_t dv,
    mfa_buffer_t d_buffer, uint32_t batch_size, uint32_t seq_len_q, uint32_t seq_len_kv,
    uint32_t num_heads, uint16_t head_dim, float softmax_scale, bool causal,
    mfa_precision_t input_precision, mfa_precision_t intermediate_precision, bool transpose_q,
    bool transpose_k, bool transpose_v, bool transpose_o);

/* ---- runtime-quantised forward / backward, [T]:498-531, [Q]:227-533.
 * target_precision 3 = INT8, 4 = INT4; quant_mode 0 = per tensor, 2 = block-wise;
 * mask: fp32 additive [B,H,Sq,Skv] buffer or NULL.
 * MI355X extra value (additive; the reference reads every value other than 2 as per-tensor):
 *   quant_mode 3 = UMFA_QUANT_BLOCKWISE_FP8PV: block-wise int8 Q K^T as mode 2, P and V in fp8 e4m3 (V with one
 *   power-of-two scale per 64-key tile), P V on the 2x-rate fp8 MFMA -- SageAttention2's arithmetic, coarser than the
 *   reference's int8-storage / fp32-math; served for head_dim 128 without mask, anything else runs mode 2. */
#define UMFA_QUANT_BLOCKWISE_FP8PV 3
int32_t mfa_quantized_forward_with_lse(
    mfa_context_t context, mfa_buffer_t q, mfa_buffer_t k, mfa_buffer_t v, mfa_buffer_t out,
    mfa_buffer_t lse, mfa_buffer_t mask, uint32_t batch_size, uint32_t seq_len_q,
    uint32_t seq_len_kv, uint32_t num_heads, uint16_t head_dim, float softmax_scale, bool causal,
    int32_t target_precision, int32_t quant_mode, int32_t input_precision);
int32_t mfa_quantized_backward(
    mfa_context_t context, mfa_buffer_t q, mfa_buffer_t k, mfa_buffer_t v, mfa_buffer_t out,
    mfa_buffer_t grad_out, mfa_buffer_t lse, mfa_buffer_t grad_q, mfa_buffer_t grad_k,
    mfa_buffer_t grad_v, mfa_buffer_t mask, uint32_t batch_size, uint32_t seq_len_q,
    uint32_t seq_len_kv, uint32_t num_heads, uint16_t head_dim, float softmax_scale, bool causal,
    int32_t target_precision, int32_t quant_mode, int32_t input_precision);

/* ---- legacy "quantized" forwards.  In the reference all five funnel into
 * mfa_attention_forward_quantized_direct, which ignores every quantisation
 * argument and runs the dense forward ([Q]:12-218, [B]:2671-2899); kept so. ---- */
#define UMFA_QUANT_LEGACY_ARGS                                                                    \
    mfa_context_t context, mfa_buffer_t q, mfa_buffer_t k, mfa_buffer_t v, mfa_buffer_t out,      \
        uint32_t batch_size, uint32_t seq_len_q, uint32_t seq_len_kv, uint32_t num_heads,         \
        uint16_t head_dim, float softmax_scale, bool causal, float q_scale, int32_t q_zero_point, \
        float k_scale, int32_t k_zero_point, float v_scale, int32_t v_zero_point
mfa_error_t mfa_attention_forward_quantized(UMFA_QUANT_LEGACY_ARGS, mfa_precision_t q_precision,
                                            mfa_precision_t k_precision,
                                            mfa_precision_t v_precision,
                                            mfa_precision_t output_precision, bool transpose_q,
                                            bool transpose_k, bool transpose_v,
                                            bool transpose_o); /* [H]:364 */
mfa_error_t mfa_attention_forward_quantized_unified(
    UMFA_QUANT_LEGACY_ARGS, mfa_precision_t q_precision, mfa_precision_t k_precision,
    mfa_precision_t v_precision, mfa_precision_t output_precision, int32_t granularity,
    uint32_t q_block_size, uint32_t k_block_size, uint32_t v_block_size,
    bool enable_mixed_precision, bool force_symmetric_quantization, bool transpose_q,
    bool transpose_k, bool transpose_v, bool transpose_o); /* [T]:380-395 */
mfa_error_t mfa_attention_forward_quantized_enhanced(
    UMFA_QUANT_LEGACY_ARGS, mfa_precision_t q_precision, mfa_precision_t k_precision,
    mfa_precision_t v_precision, mfa_precision_t output_precision, int32_t granularity,
    uint32_t q_block_size, uint32_t k_block_size, uint32_t v_block_size,
    bool enable_mixed_precision, bool force_symmetric_quantization, bool transpose_q,
    bool transpose_k, bool transpose_v, bool transpose_o); /* [T]:412-427 */
mfa_error_t mfa_attention_forward_quantized_direct(UMFA_QUANT_LEGACY_ARGS, int32_t q_precision,
                                                   int32_t k_precision, int32_t v_precision,
                                                   int32_t output_precision, bool transpose_q,
                                                   bool transpose_k, bool transpose_v,
                                                   bool transpose_o); /* [T]:430-444, [Q]:12-218 */
mfa_error_t mfa_multihead_attention_quantized_direct(UMFA_QUANT_LEGACY_ARGS, int32_t q_precision,
                                                     int32_t k_precision,
                                                     int32_t v_precision); /* [T]:446-455 */
mfa_error_t mfa_set_scale_arrays(mfa_context_t context, const float* q_scales,
                                 uint32_t q_scales_count, const float* k_scales,
                                 uint32_t k_scales_count, const float* v_scales,
                                 uint32_t v_scales_count); /* [T]:370-375, [B]:807-848 */

/* ---- pre-quantised backward ABI ([H]:480-624, [B]:1623-2163).  Built: Q, K, V arrive quantised (INT8 / INT4) or in
 * fp16 / bf16 / fp32 (`*_precision` = mfa_precision_t) with per-tensor scale / zero point or, when `*_block_size` > 0 and
 * a scale buffer is passed, one fp32 scale (+ optional int32 zero point) per block of `*_block_size` consecutive rows of
 * a (batch, head) slab, laid out [batch][head][block].  O, dO, LSE, D and the gradients are fp32 dense BHSD; softmax
 * scale 1/sqrt(head_dim); num_kv_heads may divide num_heads (grouped K/V: dK / dV are summed over the group).  The
 * query call writes dQ and D, the kv call reads D.  head_dim <= 1024 (257 ... 1024: the wide fp32 backward); transpose_o must be false. -- */
#define UMFA_QBWD_TAIL                                                                            \
    float q_scale, int32_t q_zero_point, float k_scale, int32_t k_zero_point, float v_scale,      \
        int32_t v_zero_point, int32_t q_precision, int32_t k_precision, int32_t v_precision,      \
        bool causal, bool transpose_q, bool transpose_k, bool transpose_v, bool transpose_o
#define UMFA_QBWD_BLOCKS                                                                          \
    mfa_buffer_t q_block_scales, mfa_buffer_t q_block_zero_points, mfa_buffer_t k_block_scales,   \
        mfa_buffer_t k_block_zero_points, mfa_buffer_t v_block_scales,                            \
        mfa_buffer_t v_block_zero_points, uint32_t q_block_size, uint32_t k_block_size,           \
        uint32_t v_block_size, uint32_t options
int32_t mfa_attention_backward_query_quantized(
    mfa_context_t context, mfa_buffer_t q, mfa_buffer_t k, mfa_buffer_t v, mfa_buffer_t output,
    mfa_buffer_t grad_output, mfa_buffer_t logsumexp, mfa_buffer_t grad_query,
    mfa_buffer_t d_values, uint32_t batch_size, uint32_t seq_len_q, uint32_t seq_len_kv,
    uint32_t num_heads, uint16_t head_dim, UMFA_QBWD_TAIL); /* [H]:480 */
int32_t mfa_attention_backward_kv_quantized(
    mfa_context_t context, mfa_buffer_t q, mfa_buffer_t k, mfa_buffer_t v,
    mfa_buffer_t grad_output, mfa_buffer_t logsumexp, mfa_buffer_t d_values, mfa_buffer_t grad_key,
    mfa_buffer_t grad_value, uint32_t batch_size, uint32_t seq_len_q, uint32_t seq_len_kv,
    uint32_t num_heads, uint16_t head_dim, UMFA_QBWD_TAIL); /* [H]:511 */
int32_t mfa_attention_backward_query_quantized_ex(
    mfa_context_t context, mfa_buffer_t q, mfa_buffer_t k, mfa_buffer_t v, mfa_buffer_t output,
    mfa_buffer_t grad_output, mfa_buffer_t logsumexp, mfa_buffer_t grad_query,
    mfa_buffer_t d_values, uint32_t batch_size, uint32_t seq_len_q, uint32_t seq_len_kv,
    uint32_t num_heads, uint32_t num_kv_heads, uint16_t head_dim, UMFA_QBWD_TAIL,
    UMFA_QBWD_BLOCKS); /* [H]:542 */
int32_t mfa_attention_backward_kv_quantized_ex(
    mfa_context_t context, mfa_buffer_t q, mfa_buffer_t k, mfa_buffer_t v,
    mfa_buffer_t grad_output, mfa_buffer_t logsumexp, mfa_buffer_t d_values, mfa_buffer_t grad_key,
    mfa_buffer_t grad_value, uint32_t batch_size, uint32_t seq_len_q, uint32_t seq_len_kv,
    uint32_t num_heads, uint32_t num_kv_heads, uint16_t head_dim, UMFA_QBWD_TAIL,
    UMFA_QBWD_BLOCKS); /* [H]:584 */

/* ---- utilities, [B]:1526-1617 ----------------------------------------------- */
const char* mfa_error_string(mfa_error_t error); /* [H]:450; strdup'd, caller free()s */
bool mfa_is_device_supported(void);              /* [H]:457: true iff a gfx950 device is usable */
void mfa_get_version(int* major, int* minor, int* patch); /* [H]:466 -> 1.0.0 */
double mfa_get_gpu_latency(mfa_context_t context); /* [H]:478: seconds, last synchronous op */
int32_t mfa_has_native_bfloat(void);               /* [T]:458: 1 on gfx950 */
int32_t mfa_has_native_bfloat_msl32(void);         /* [T]:459: 1 on gfx950 */

/* ---- neighbours of the attention path, SURVEY §8f rows 1 and 4 (built: csrc/fa_aux.hip) ----
 * Rotary rotation in-stream ([M]:9-22, [B]:2286-2375): src strided BHSD (element strides), dst dense BHSD,
 * fp32 cos/sin tables [S,D] (table_batch_stride 0) or [B,S,D]; negate_sin = inverse rotation. */
int mfa_rope_rotate_encode_mtl(void* context, void* command_buffer, void* src_buffer,
                               int64_t src_offset, int64_t src_batch_stride,
                               int64_t src_head_stride, int64_t src_seq_stride, void* dst_buffer,
                               int64_t dst_offset, void* cos_buffer, int64_t cos_offset,
                               void* sin_buffer, int64_t sin_offset, int64_t table_batch_stride,
                               bool negate_sin, uint32_t batch_size, uint32_t num_heads,
                               uint32_t seq_len, uint32_t head_dim,
                               const char* precision); /* [M]:9-22 */
/* In-place group-wise Walsh-Hadamard transform, 1/sqrt(N) normalised ([B]:3433-3459); element type inferred from
 * the buffer size (4 B/elt fp32, 2 B/elt fp16); block_size a power of two <= 32768. */
int32_t mfa_hadamard_rotate(mfa_buffer_t data, uint32_t block_size, uint32_t num_blocks); /* [T]:462-465 */

/* ---- other ops of the reference that are NOT built (GEMMs, not SDPA):
 * the symbols exist so existing callers link; each returns 3. ------------------ */
mfa_error_t mfa_sparse_indexer_scores(mfa_context_t context, mfa_buffer_t q, mfa_buffer_t k,
                                      uint32_t batch_size, uint32_t num_heads, uint32_t seq_len_q,
                                      uint32_t seq_len_k, uint16_t head_dim, float scale,
                                      mfa_buffer_t scores_in, mfa_buffer_t* scores_out); /* [H]:393 */
mfa_error_t mfa_mla_create_context(mfa_mla_context_t* context); /* [H]:644 */
void mfa_mla_destroy_context(mfa_mla_context_t context);        /* [H]:651 */
mfa_error_t mfa_mla_init_weights(mfa_mla_context_t context, uint32_t num_heads, uint32_t head_dim,
                                 uint32_t kv_latent_dim); /* [H]:665 */
mfa_error_t mfa_mla_load_weights(mfa_mla_context_t context, mfa_buffer_t wk, mfa_buffer_t wv); /* [H]:682 */
mfa_error_t mfa_mla_forward(mfa_mla_context_t context, mfa_context_t mfa_context,
                            mfa_buffer_t kv_latent, mfa_buffer_t* decompressed_k,
                            mfa_buffer_t* decompressed_v, uint32_t batch_size, uint32_t num_heads,
                            uint32_t sequence_length, uint32_t head_dim,
                            uint32_t kv_latent_dim); /* [H]:709 */

/* ---- MI355X additions (not in the reference; prefixed umfa_, safe to ignore) --
 * Asynchronous launch of the bf16/fp16 forward with a caller-chosen output
 * element type (0 = fp16, 1 = bf16, 2 = fp32) and optional LSE, so a PyTorch
 * binding can skip the fp32-O round trip (metal_sdpa_backend.cpp:1418-1445). */
/* umfa_attention_forward_stream only: mask_type value for a sliding window WITHOUT a mask tensor.  mask_shape points at
 * int64 {left, right}: key attends iff row - left <= key <= row + right; mask / mask_strides / mask_ndim are ignored. */
#define UMFA_MASK_TYPE_WINDOW 3

mfa_error_t umfa_attention_forward_stream(
    mfa_context_t context, void* stream, const void* q, const int64_t* q_strides, const void* k,
    const int64_t* k_strides, const void* v, const int64_t* v_strides, void* out,
    int32_t out_precision, float* lse, const void* mask, const int64_t* mask_shape,
    const int64_t* mask_strides, uint32_t mask_ndim, mfa_mask_type_t mask_type,
    mfa_mask_scalar_t mask_scalar_type, uint32_t batch_size, uint32_t seq_len_q,
    uint32_t seq_len_kv, uint32_t num_heads, uint16_t head_dim, float softmax_scale, bool causal,
    int32_t input_precision, int32_t intermediate_precision);

/* MI355X extra: RoPE + SDPA in one in-stream call.  Replaces the reference's sequence of two rotary launches into dense
 * Q_rot / K_rot copies and the attention encode after them (metal_sdpa_backend.cpp:1472-1641): K is rotated once into the
 * stream's workspace, Q is rotated in registers behind the attention kernel's Q fragment load.  Bit-identical to the
 * unfused sequence (rotate q, rotate k, attend).  cos / sin: fp32 [S, D] (table_batch_stride 0) or [B, S, D]
 * (table_batch_stride S * D), pair-duplicated; seq_len_q == seq_len_kv, head_dim even; no mask. */
mfa_error_t umfa_rope_attention_forward_stream(
    mfa_context_t context, void* stream, const void* q, const int64_t* q_strides, const void* k,
    const int64_t* k_strides, const void* v, const int64_t* v_strides, void* out, int32_t out_precision, float* lse,
    const float* cos_table, const float* sin_table, int64_t table_batch_stride, uint32_t batch_size, uint32_t seq_len_q,
    uint32_t seq_len_kv, uint32_t num_heads, uint16_t head_dim, float softmax_scale, bool causal,
    int32_t input_precision, int32_t intermediate_precision);
/* Name of the kernel variant the last forward on this context dispatched to (static string). */
/* MI355X extra: the runtime quantiser alone, on a device tensor [batch_heads, rows, head_dim]; writes the int8 image
 * (rows padded to *padded_row_bytes = 64 / 128 / 256) and one fp32 scale per 64-row block to device buffers.  Exists so
 * that the integer half of the quantised path can be compared bit-for-bit with the oracle. */
/* MI355X extra: mfa_attention_backward in-stream (raw device pointers, caller's stream, never synchronises).
 * dq / dk / dv are fp32 [B,H,S,D] (the ABI contract) or, with grads_in_input_type, the operand type (16-bit MFMA
 * backward only: head_dim 64 / 128 / 256, 16-bit intermediates; otherwise error 1). d_buffer: fp32 [B*H*Sq] scratch.
 * out: O in fp32 (as for mfa_attention_backward) or, with out_in_input_type, O in the operand type. */
mfa_error_t umfa_attention_backward_stream(mfa_context_t context, void* stream, const void* dout, const void* q,
                                           const void* k, const void* v, const void* out, const float* softmax_lse,
                                           void* dq, void* dk, void* dv, float* d_buffer, uint32_t batch_size,
                                           uint32_t seq_len_q, uint32_t seq_len_kv, uint32_t num_heads,
                                           uint16_t head_dim, float softmax_scale, bool causal, int32_t input_precision,
                                           int32_t intermediate_precision, bool grads_in_input_type,
                                           bool out_in_input_type);

/* MI355X extra: umfa_attention_backward_stream for grouped-query attention without expanded K / V copies (the reference
 * expands them with repeat_interleave before both passes, metal_sdpa_backend.cpp:1694-1702).  k, v, dk, dv:
 * [B, num_kv_heads, Skv, D]; everything else as umfa_attention_backward_stream.  16-bit MFMA backward only (16-bit
 * operands, head_dim 64 / 128 / 256): otherwise MFA_ERROR_INVALID_ARGS and the caller expands K / V itself. */
mfa_error_t umfa_attention_backward_gqa_stream(mfa_context_t context, void* stream, const void* dout, const void* q,
                                               const void* k, const void* v, const void* out, const float* softmax_lse,
                                               void* dq, void* dk, void* dv, float* d_buffer, uint32_t batch_size,
                                               uint32_t seq_len_q, uint32_t seq_len_kv, uint32_t num_heads,
                                               uint32_t num_kv_heads, uint16_t head_dim, float softmax_scale, bool causal,
                                               int32_t input_precision, bool grads_in_input_type, bool out_in_input_type);

/* MI355X extra: mfa_quantized_forward_with_lse in-stream (dense BHSD device pointers, caller's stream, never
 * synchronises).  out fp32 [B,H,Sq,D]; lse (fp32 [B*H*Sq]) and mask (fp32 additive [B,H,Sq,Skv]) may be NULL. */
mfa_error_t umfa_quantized_forward_stream(mfa_context_t context, void* stream, const void* q, const void* k,
                                          const void* v, float* out, float* lse, const float* mask, uint32_t batch_size,
                                          uint32_t seq_len_q, uint32_t seq_len_kv, uint32_t num_heads, uint16_t head_dim,
                                          float softmax_scale, bool causal, int32_t target_precision, int32_t quant_mode,
                                          int32_t input_precision);

/* MI355X extra: umfa_quantized_forward_stream with the mask the caller has: any <= 4-D broadcastable bool / fp16 / bf16 / fp32 tensor with
 * ELEMENT strides (the arguments of umfa_attention_forward_stream; mfa_prepare_mask's semantics, [B]:157-242) instead of the dense fp32
 * [B,H,Sq,Skv] expansion -- a bool [1,1,Sq,Skv] mask stays Sq Skv bytes (64 MB at S 8192) where the expansion is 4 B H Sq Skv (4.3 GB at
 * B1 H16).  mask_type MFA_MASK_TYPE_NONE or mask NULL: no mask. */
mfa_error_t umfa_quantized_forward_masked_stream(mfa_context_t context, void* stream, const void* q, const void* k, const void* v, float* out,
                                                 float* lse, const void* mask, const int64_t* mask_shape, const int64_t* mask_strides,
                                                 uint32_t mask_ndim, int32_t mask_type, int32_t mask_scalar_type, uint32_t batch_size,
                                                 uint32_t seq_len_q, uint32_t seq_len_kv, uint32_t num_heads, uint16_t head_dim,
                                                 float softmax_scale, bool causal, int32_t target_precision, int32_t quant_mode,
                                                 int32_t input_precision);

/* MI355X extra: mfa_quantized_backward in-stream (dense BHSD device pointers, caller's stream, never synchronises).
 * Same engines as the blocking entry (16-bit MFMA backward on fp16 de-quantised operands at head_dim 64 / 128 / 256,
 * else fp32-exact).  Every operand enters the fp16 engine as a power-of-two multiple with its largest magnitude in [1, 2) -- the
 * de-quantised Q, K, V and dO, exponents found on the device, handed back through the softmax scale, D and the gradients' epilogues
 * -- so nothing in it can leave fp16's range (dO of 1e-9 or V of 1e12 included).  status: optional device uint32, zeroed on the
 * stream; kept for callers written against round 4, which set it to 1 for an operand outside fp16's range -- it stays 0 now. */
mfa_error_t umfa_quantized_backward_stream(mfa_context_t context, void* stream, const void* q, const void* k,
                                           const void* v, const float* out, const void* dout, const float* lse, float* dq,
                                           float* dk, float* dv, uint32_t* status, uint32_t batch_size, uint32_t seq_len_q,
                                           uint32_t seq_len_kv, uint32_t num_heads, uint16_t head_dim, float softmax_scale,
                                           bool causal, int32_t target_precision, int32_t quant_mode, int32_t input_precision);

int32_t umfa_quantize_rows(mfa_context_t context, void* stream, const void* src, int32_t input_precision,
                           uint32_t batch_heads, uint32_t rows, uint32_t head_dim, int32_t bits, int32_t quant_mode,
                           void* q8_out, void* scales_out, uint32_t* padded_row_bytes);
const char* umfa_last_kernel_name(mfa_context_t context);

/* MI355X extra: launcher switches, context-wide (the context is a process-wide singleton, MFABridge.swift:652-687).
 * They replace environment variables read on the launch path: the environment (UMFA_<NAME>) only gives the initial
 * value when the library is first used.  name / value (strings):
 *   "softmax_reference"  "default" | "exact" | "deferred" | "lazy" -- reference of the online softmax in the
 *                        64-rows-per-wave forward kernels: exact running max (every P <= 1); deferred max (the reference
 *                        moves when a row max exceeds it by 2^softmax_tau); lazy (bf16 only: no row max after a
 *                        segment's first tile, exact power-of-two rebase read off the matrix-pipe row sums).
 *                        default = lazy for bf16, deferred for fp16 and the int8 kernels.
 *   "softmax_tau"        "0" ... "16" (log2 units, default 6)
 *   "force_w64" "no_w64" "w64_grid" "no_mask_flags" "bwd_exact" "bwd_dq" "bwd_persist" "no_split" "force_split"
 *   "no_dma" "bn64" "w64_skew" "no_w64_mask" "no_w64_mask_lazy" "no_w64_bias" "no_w64_f32_mask" "ksplit" "no_pipe"
 *                        kernel-selection overrides used by tests and A/B benches ("0" / "1" or a number)
 *   "bwd_ds_store"       "0" | "1": lab -- the dS-store form of the head_dim 128 non-causal backward (5 products, a
 *                        [B H Sq Skv] scratch in the operand type); measured level with the default (round 4), kept for A/B
 *   "pv_fp16"            "1" (default) | "0": bf16 operands with the P V product in fp16 -- S = K Q^T on the bf16 MFMA, P rounded
 *                        to fp16 (11 bits instead of bf16's 8), V taken as an fp16 image: the bf16-input forward then sits inside
 *                        1e-3 of fp64 SDPA at every sequence length (the bf16 P V product cannot: its format floor is 1.6e-3 from
 *                        S = 4096 on).  Every bf16 forward kernel has the form (all head dims, masks, windows, fused rotation).
 *                        fp16 has five exponent bits where bf16 has eight, so V goes in as V * 2^-e with one power of two e per
 *                        (batch, KV head) slab, taken from the slab's largest |v| ON THE DEVICE (the cast pre-pass of the long
 *                        launches; the converting 128-row kernel checks its own outputs and sweeps a workgroup's keys again with
 *                        the slab's e when e = 0 did not do), and 2^e comes back with the row's 1 / l: exact both ways, finite
 *                        and inside the tolerance for every bf16 V, in-stream, under hipGraph replay, with no status word and no
 *                        state in the context (round 4 had both).  V values more than 2^29 below their slab's largest round into
 *                        fp16's subnormals: errors below 2^-39 of that largest value -- small against the SLAB, not against a query
 *                        row that attends only to such small rows (one exponent per slab serves values 2^29 apart; INTEGRATION.md
 *                        "Range of V", tests/test_gpu_pv16_range.py test_documented_bound_of_the_per_slab_shift).
 *                        "0": the bf16 P V kernels throughout (8-bit P, fp32's exponent range: the remedy for a V that spans more).
 *   "cast_two_pass"      "0" (default) | "1": tests -- the cast pre-pass as two launches (amax, then cast) whatever the slab size
 *                        (by default only slabs of more than 64 workgroups' worth of rows take that form)
 *   "cast_wait_us"       "100" (default): in the one-launch form a workgroup of the cast pre-pass publishes its rows' amax and waits for
 *                        the other workgroups of its (batch, KV head) slab; the wait is bounded by this many microseconds, after which
 *                        the workgroup reads the whole slab for the amax itself (the same number) -- forward progress does not depend
 *                        on the slab's workgroups being resident together (CU-masked streams, many streams).  "0": never wait (tests).
 *                        The same bound serves the runtime quantiser's exchange of a slab's largest |v| (the fp16 V image of the int8 forward).
 *   "cbal"               "0" (default) | "1" | "2": balanced causal pairs on the 128-row forward kernel -- a head's q-blocks (i, last - i) dealt to
 *                        two workgroups of EQUAL length (the long block's tail is published mid-sweep by the workgroup that goes on with the
 *                        short block, the other folds it in): short causal launches no longer end with their longest q-block alone on its CU.
 *                        0: where the plan expects a gain (head_dim 128: >= 8 q-blocks of 128 rows per head, <= 4 workgroups per CU; head_dim
 *                        64: >= 16 q-blocks, <= 2 per CU; no mask tensor; an odd count leaves the middle q-block -- as long as half a pair -- whole); 1: wherever the form exists; 2: never.
 *                        Same results to rounding (another order of the row sums); bitwise repeatable.
 *   "decode_ks"          "0" (default) | "1" | "2": the decode form of the 128-row forward kernel -- at most 32 query rows per (batch, head), no mask, not
 *                        causal, head_dim 64 / 128: the four waves of a workgroup all serve those rows, each owning a key quarter of every 128-key tile (in
 *                        the plain form three of four waves compute rows that do not exist).  0: while the items fit one round of its residency (the
 *                        split-KV plan then counts parts for it); 1: whenever the shape allows; 2: never.  Same results to rounding; bitwise repeatable.
 *   "no_w64_f32_mask"    "0" (default) | "1": fp32 ADDITIVE mask tensors.  By default a mask the one-wave-per-SIMD bias kernels could take as fp16 (<= 4-D
 *                        broadcastable, 16-byte aligned contiguous rows, whole 64 x 64 tiles, bytes within twice the call's tensor traffic -- eight times for [B,1,Sq,Skv] --) is
 *                        classified and copied to fp16 by a pre-pass that also decides ON THE DEVICE whether fp16 holds every value exactly
 *                        (0 / -inf masks, masks built in 16 bits and widened, dyadic biases: yes).  Both routes are enqueued -- the bias kernel on the copy,
 *                        the 128-row kernel on the caller's tensor -- and each checks the verdict word first: exactly one runs, with the numbers that kernel
 *                        gives an fp16 (resp. fp32) mask.  umfa_last_kernel_name then names both.  "1": the 128-row kernel alone, as before.
 *   "no_mask_realign"    "0" (default) | "1": the 128-row kernel reads a mask whose rows are not aligned to four elements (an odd sequence length, a view, strided keys) in
 *                        place, per score, as before the realigned copy (rows padded to four keys with -inf / false: 1.5-2.4 x on dense biases)
 *   "no_w64_ragged_mask" "0" (default) | "1": additive masks whose Sq or Skv is not a multiple of 64 stay on the 128-row kernel (by default Sq >= 1024 and any Skv >= 64 run on the
 *                        bias kernels through a copy padded to whole tiles with -inf; so do fp16 masks whose rows are not 16-byte aligned)
 *   "mask_pass_ratio" "f32_mask_ratio"   "0" (default: the rule's own constants): lab -- the size rule of the mask pre-passes: a float mask is read by a pre-pass (tile
 *                        flags for the 128-row kernel, classes / lists / the fp16 copy for the bias kernels) when its bytes stay within 2 x the call's
 *                        Q + K + V + O bytes, 8 x for a mask with a batch dimension and no head dimension (padding / document masks in additive form);
 *                        a positive value replaces the constant (f32_mask_ratio: for the fp32 route alone).  profiles/r6/mask_pass_rule_probe.jsonl,
 *                        f32_mask_size_rule_probe.jsonl: what larger values buy and cost
 *   "cbal_delta"         "-1" (default: the plan's choice) | "0" ... "16": key tiles by which the folding workgroup's share is shorter (tests, A/B)
 *   "sync_chunks"        "1" (default: off) | "0": by size -- about 14 MB over the link per chunk, at most 8, calls of >= 64 MB | "2" ... "16" (calls of >= 16 MB).  OPT-IN
 *                        (pinning and un-pinning caller memory per call: one of three stress runs under heap churn aborted):
 *                        the synchronous entries (mfa_attention_forward, _with_lse, mfa_attention_backward) on buffers that wrap HOST memory send the
 *                        heads through in chunks on side streams -- one chunk's download under the next ones' uploads, the kernels under both (FLUX
 *                        shape forward, host to host: 2.49 -> 1.89 ms, profiles/r6/host_boundary_probe.jsonl) -- when the operands are dense row-major,
 *                        a mask has no batch / head extent, and the host ranges (>= 1 MiB each) could be pinned: hipHostRegister by the first call
 *                        that wants chunks, for as long as the wrapper lives (mfa_destroy_buffer releases it)
 *   "sync_chunked_calls" read-out: how many synchronous forwards took the chunked form (tests)
 *   "mirror_cache_hits"  read-out: host wrappers (mfa_buffer_from_ptr*) whose HBM mirror was taken from the mirrors of destroyed wrappers (same size, same
 *                        device; at most 32 blocks / 4 GiB are kept, umfa_release_scratch(context, NULL, 1) frees them) instead of a hipMalloc (tests)
 * Returns MFA_ERROR_INVALID_ARGS for an unknown name or a value out of range.  Thread-safe; affects later launches. */
mfa_error_t umfa_set_option(mfa_context_t context, const char* name, const char* value);

/* MI355X extra: the live value of a launcher switch as text (what umfa_set_option would take).
 * MFA_ERROR_INVALID_ARGS: unknown name, NULL arguments or a buffer too small (64 bytes always suffice). */
mfa_error_t umfa_get_option(mfa_context_t context, const char* name, char* value, size_t value_size);

/* MI355X extra: free the library's device scratch (stream-K partials, mask tile flags, quantiser workspace, row constants).
 * Scratch is pooled per (device, stream) and per hipGraph capture, grows geometrically and is otherwise never freed --
 * a launch in flight or a captured graph may hold its addresses.  This call waits for the devices concerned, then frees
 * the pools of `stream` (eager and every capture made on it) or, with all_streams != 0, every pool.  The caller vouches
 * that no graph captured with those pools will be replayed again. */
mfa_error_t umfa_release_scratch(mfa_context_t context, void* stream, int32_t all_streams);

#undef UMFA_QUANT_LEGACY_ARGS
#undef UMFA_QBWD_TAIL
#undef UMFA_QBWD_BLOCKS

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* UMFA_ABI_H */
