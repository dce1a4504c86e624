// kernels.h -- host-callable launchers of the gfx950 kernels (all asynchronous on `stream`).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

#include "fa_common.h"

namespace umfa {

// Process-wide launcher switches (tuning.hip): initial values from the environment once, then umfa_set_option only.
enum SoftmaxRef : int {
    SM_DEFAULT = 0,   // bf16: lazy; fp16 / int8 kernels: deferred max with threshold 2^sm_tau
    SM_EXACT = 1,     // the reference max is the exact running max (every P <= 1)
    SM_DEFERRED = 2,  // deferred max: the reference moves when a row max exceeds it by 2^sm_tau (T13)
    SM_LAZY = 3,      // bf16 only: no row max after a segment's first tile, rebase by exact powers of two off the row sums
};
struct Tuning {
    std::atomic<int> sm_mode{SM_DEFAULT};
    std::atomic<float> sm_tau{6.0f};
    std::atomic<int> force_w64{0}, no_w64{0}, w64_grid{0}, w64_skew{0}, no_mask_flags{0}, bwd_exact{0}, bwd_dq{0}, bwd_persist{0},
        bwd_separate_delta{0}, no_split{0}, force_split{0}, no_dma{0}, bn64{0}, pv_fp16{0}, bwd_ds_store{0}, no_w64_mask{0}, ksplit{0}, no_pipe{0}, no_w64_mask_lazy{0}, no_w64_bias{0}, no_w64_f32_mask{0} /* fp32 additive masks stay on the 128-row kernel (no classification pass, no guarded pair of launches) */,
        no_mask_realign{0} /* the 128-row kernel reads a mask whose rows are not aligned to four elements in place (per score), as before the realigned copy */,
        no_w64_ragged_mask{0} /* additive masks of ragged shapes (Sq or Skv not a multiple of 64) stay on the 128-row kernel */,
        mask_pass_ratio{0} /* lab: the constant of mask_flags_worthwhile (float masks are read by a pre-pass when their bytes stay within this many times the call's Q + K + V + O bytes); 0 = the rule's own */,
        f32_mask_ratio{0} /* ... lab: > 0 = the pair is taken for fp32 masks of up to this many times the call's Q + K + V + O bytes, in place of mask_flags_worthwhile's rule */,
        cast_two_pass{0} /* V cast pre-pass: amax and cast as two launches whatever the slab size (tests) */, bwd_ds_lab{0} /* lab, timing only: BwdParams::ds_lab */,
        cast_u{0} /* lab: 16-byte loads per thread of the V cast pre-pass (4, 16, 32; 0 = the launcher's choice) */,
        quant_block_wg{0} /* tests / A-B: the block-wise quantiser in its one-workgroup-per-block form everywhere */,
        cast_wait_us{100} /* V cast pre-pass: how long a workgroup waits for its slab's other workgroups before it reads the slab's amax itself */,
        decode_ks{0} /* decode form of the 128-row kernel (<= 32 query rows: four key quarters per tile): 0 = where it applies, 2 = never */,
        cbal{0} /* balanced causal pairs on the 128-row kernel: 0 = where the plan wants them, 1 = wherever they exist, 2 = never */,
        cbal_delta{-1} /* ... key tiles by which a pair's folding part is shorter than half; < 0 = the plan's choice */,
        sync_chunks{1} /* synchronous forward / backward on host-wrapping buffers: head chunks whose upload / kernels / download overlap on side streams (pinned host ranges); 1 = never (the default: one upload, the kernels, one download on the null stream), 0 = by size, n = n chunks.  OPT-IN: one of three runs of tools/lab/sync_chunk_stress.py (pin / unpin per call under heap churn, torch in-process) aborted, profiles/r6/lab_notes.md section 24 */,
        sync_chunked_calls{0} /* read-out for tests: synchronous forwards that took the chunked form */,
        mirror_cache_hits{0} /* read-out for tests: host wrappers whose HBM mirror came from the cache of destroyed wrappers' mirrors (runtime_internal.h MirrorCache) */;
};
Tuning& tuning();
bool set_tuning(const char* name, const char* value);  // false: unknown name or value out of range
bool get_tuning(const char* name, char* out, size_t n);  // the live value as text; false: unknown name / buffer too small
// per-device launch state (tuning.hip): CU count of the CURRENT device; MaxDynamicSharedMemorySize of `kernel` on the
// current device raised to >= bytes (once per device and kernel)
int device_cu_count();
hipError_t ensure_dynamic_lds(const void* kernel, size_t bytes);

// fp32-exact forward (any input type, any D <= 256, masks, causal, LSE).
hipError_t launch_fwd_exact(const FwdParams& p, hipStream_t stream, const char** name);

// head dims 257 ... 1024 (fa_fwd_wide.hip): fp32 arithmetic, any operand type / strides / mask; the reference's callers admit them
// (metal_sdpa_backend.cpp:1078-1086), nothing tuned depends on them
hipError_t launch_fwd_wide(const FwdParams& p, hipStream_t stream, const char** name);
// ... and their backward (fa_bwd_wide.hip): dense contiguous BHSD, fp32 gradients, no mask
hipError_t launch_bwd_wide(const BwdParams& p, hipStream_t stream, const char** name);

// bf16 / fp16 MFMA forward.  Requires in_prec in {FP16, BF16}, D % 8 == 0, D <= 256,
// 16-byte aligned operands/strides, scale > 0.  Returns hipErrorNotSupported otherwise
// (the caller then takes the exact path).
hipError_t launch_fwd_16(const FwdParams& p, hipStream_t stream, const char** name);
bool fwd_16_supported(const FwdParams& p);
// Split-KV tail plan: how many items stay whole, parts per split item, scratch sizes (0 = no split).
struct FwdSplitPlan {
    uint32_t n_full, nsplit;
    size_t buf_bytes, cnt_bytes;
    uint32_t decode = 0;                // the decode form of the kernel (FwdParams::decode_form): the part count was chosen for ITS residency
    uint32_t cbal = 0, cbal_delta = 0;  // balanced causal pairs (FwdParams::cbal): nsplit stays 1, the scratch is the pairs'
};
FwdSplitPlan fwd_16_split_plan(const FwdParams& p);

// 64-rows-per-wave persistent forward (fa_fwd16_w64.hip): head_dim 128, no mask, non-causal, Sq % 256 == 0,
// Skv % 64 == 0.  Needs a zeroed ticket array (cnt_bytes) and a partials buffer (buf_bytes).
struct FwdW64Plan {
    size_t buf_bytes, cnt_bytes;
};
bool fwd_w64_supported(const FwdParams& p);
uint32_t fwd_w64_grid(const FwdParams& p);  // workgroups launch_fwd_w64 will start for this call
// the dispatcher's cost model (fa_fwd16_w64.hip): predicted microseconds of this launch on the one-workgroup-per-CU kernel / the 128-row kernel
double fwd_w64_predict_us(const FwdParams& p);
double fwd_16_predict_us(const FwdParams& p);
FwdW64Plan fwd_w64_plan(const FwdParams& p);
hipError_t launch_fwd_w64(const FwdParams& p, float* part_buf, uint32_t* part_cnt, hipStream_t stream, const char** name);

// Backward: D = rowsum(dO o O), then dQ and dK/dV.  launch_bwd: fp32-exact (any input type, head_dim <= 128);
// launch_bwd_16: bf16 / fp16 MFMA (head_dim 128, dO in the input type, no mask).
hipError_t launch_bwd(const BwdParams& p, hipStream_t stream, const char** name);
bool bwd_16_supported(const BwdParams& p);
hipError_t launch_bwd_16(const BwdParams& p, hipStream_t stream, const char** name);

// Neighbours of the attention path (fa_aux.hip): rotary rotation and group-wise Hadamard transform.
struct RopeParams {
    const void* src;
    void* dst;
    const float* cos_table;
    const float* sin_table;
    int64_t src_batch_stride, src_head_stride, src_seq_stride;  // elements; head_dim contiguous
    int64_t table_batch_stride;                                  // 0: one [S,D] table for every batch
    uint32_t B, H, S, D;
    int negate_sin;
};
hipError_t launch_rope(const RopeParams& p, int prec, hipStream_t stream);
hipError_t launch_hadamard(void* data, uint32_t block_size, uint32_t num_blocks, int prec, hipStream_t stream);

// De-quantisation of caller-quantised operands into fp32 [B, H_dst, S, D] (pre-quantised backward ABI) and the
// group sum of per-query-head dK / dV for grouped key/value heads.
struct DequantParams {
    const void* src;
    float* dst;
    void* dst16;                      // optional: fp16 output instead of dst (operands of the 16-bit MFMA backward)
    uint32_t* overflow;               // with dst16: set to 1 when a value does not fit fp16
    const float* block_scales;        // optional, [B * H_src * ceil(S / block_size)]
    const int32_t* block_zero_points; // optional, same shape
    uint32_t B, H_src, H_dst, S, D, block_size;
    float scale;
    int zero_point;
    int prec;        // P_INT8 / P_INT4 / P_FP16 / P_BF16 / P_FP32
    int transposed;  // source slab stored [D, S]
    // the fp16 images as power-of-two multiples (BwdParams::units): a first launch with amax_word set only takes the largest |x| of the
    // de-quantised tensor (fp32 bits, max-ed into the zeroed word; nothing is stored), the second, with unit_amax pointing at that word,
    // stores x * 2^-e (amax 2^-e in [1, 2)): nothing leaves fp16's range, `overflow` stays clear
    uint32_t* amax_word;
    const uint32_t* unit_amax;
};
hipError_t launch_dequant(const DequantParams& p, hipStream_t stream);
// bf16 [B,H,S,D] with element strides (head_dim contiguous) -> dense fp16 [B,H,S,D] of V * 2^-e, one power of two per (batch, head)
// slab chosen from the slab's largest |v| (no value leaves fp16's range); hdr: VSC_HDR_WORDS uint32 per slab, zero on entry except word
// VSC_HDR_SCALE, which the pass leaves holding 2^e as fp32 (FwdParams::vsc reads it)
constexpr uint32_t VSC_HDR_WORDS = 128, VSC_HDR_DEPART = 64, VSC_HDR_SCALE = 65, VSC_HDR_AMAX = 66;
hipError_t launch_cast_rows_bf16_to_f16(const void* src, const int64_t* strides, void* dst, uint32_t B, uint32_t H, uint32_t S, uint32_t D,
                                        uint32_t* hdr, hipStream_t stream);
// rowc[0 .. n) = -lse * log2 e, rowc[n .. 2n) = -dvec: what bwd16_dq leaves for bwd16_dkdv, for a dK / dV-only call
hipError_t launch_bwd16_rowc(const float* lse, const float* dvec, float* rowc, int64_t n, const float* d_mul /* one float onto D, or NULL */, hipStream_t stream);
// dO of the quantised backward entries -> fp16 as dO * 2^-e, one power of two per call from the tensor's largest |dO| (device);
// hdr = 3 words: amax bits, 2^e, 2^-e (the first three words of the 16-word units header, launch_bwd_units)
hipError_t launch_cast_f16_unit(const void* src, int prec, void* dst, int64_t n, uint32_t* hdr, hipStream_t stream, bool amax_done = false /* hdr[0] already holds the amax */);
// the quantised backward with EVERY operand as a power-of-two multiple (BwdParams::units): a 16-word header -- [0 ... 2] as above for dO,
// [4 ... 6] the largest |q|, |k|, |v| as fp32 bits (launch_amax_dense into a zeroed word; the quantiser scales its fp16 copies by them:
// launch_quantize's famax), [8 ... 14] the units table (launch_bwd_units, after all four)
hipError_t launch_amax_dense(const void* src, int prec, int64_t n, uint32_t* word, hipStream_t stream);
hipError_t launch_amax_dense_n(int count /* <= 4 */, const void* const* src, int prec, const int64_t* n, uint32_t* const* word, hipStream_t stream);  // one launch
hipError_t launch_bwd_units(uint32_t* hdr, hipStream_t stream, uint32_t* clean_flag = nullptr);  // leaves hdr[0], hdr[4 ... 6] (and *clean_flag) zero behind it
// dst: [B, Hkv, slab] in out_prec (fp32 default; fp16 / bf16: rounded once after the fp32 sum)
hipError_t launch_group_sum(const float* src, void* dst, uint32_t B, uint32_t H, uint32_t Hkv, int64_t slab, hipStream_t stream,
                            int out_prec = P_FP32);

// Runtime-quantised path (fa_quant.hip): fused symmetric quantiser for Q, K, V + int8-QK^T forward.
struct QuantViews {
    const int8_t* q8;       // [B*H*Sq][dpq] int8 (rows zero-padded to dpq)
    const int8_t* k8;       // [B*H*Skv][dpq]
    const void* v16;        // [B*H*Skv][D] fp16, de-quantised (q_v * s_v)
    const uint8_t* v8;      // quant_mode 3 only: V as fp8 e4m3 [B*H][tile][8192] in MFMA operand order, else NULL
    const uint32_t* v_e8;   // quant_mode 3 only: per-tile power-of-two scale (E8M0 byte x 4)
    const float* q_scale;   // [B*H][nqblk]
    const float* k_scale;   // [B*H][nkblk]
    const float* v_scale;
    const float* qf;        // optional fake-quantised fp32 copies [rows][D] (backward, copies = 1)
    const float* kf;
    const float* vf;
    const void* qh;         // optional fake-quantised fp16 copies [rows][D] (MFMA backward, copies = 2)
    const void* kh;
    const void* vh;
    uint32_t nqblk, nkblk, dpq;
};
bool quantized_supported(uint32_t D);
size_t quant_workspace_bytes(uint32_t B, uint32_t H, uint32_t Sq, uint32_t Skv, uint32_t D, bool want_f32);
// copies: 0 none, 1 fake-quantised fp32 copies (fp32-exact backward), 2 fp16 copies in the same workspace region (16-bit
// MFMA backward; *overflow |= 1 when q * scale does not fit fp16)
hipError_t launch_quantize(const void* q, const void* k, const void* v, int in_prec, uint32_t B, uint32_t H,
                           uint32_t Sq, uint32_t Skv, uint32_t D, int bits, int quant_mode, void* workspace,
                           int copies, QuantViews* views, hipStream_t stream, uint32_t* overflow = nullptr,
                           uint32_t* vhdr = nullptr /* slab headers (VSC_HDR_*): the fp16 V image becomes q * s * 2^-e, 2^e left in word VSC_HDR_SCALE */,
                           const uint32_t* famax = nullptr /* copies == 2: three words, the largest |q|, |k|, |v| (fp32 bits): the fp16 copies
                                                              become q * s * 2^-e_t with amax_t 2^-e_t in [1, 2) */);
// fp.q/k/v: contiguous BHSD in fp.in_prec; fp.o fp32; fp.mask: fp32 additive [B,H,Sq,Skv] or NULL.
hipError_t launch_quantized_fwd(const FwdParams& fp, int bits, int quant_mode, void* workspace, hipStream_t stream,
                                const char** name);
// 64-rows-per-wave variant of the quantised forward (fa_fwd16_w64.hip): head_dim 128, no mask; fp.part_buf / part_cnt as
// for launch_fwd_w64.  launch_quantized_fwd takes it when fp.part_buf is set and the shape qualifies.
bool fwd_w64_i8_supported(const FwdParams& p);
hipError_t launch_fwd_w64_i8(const FwdParams& p, const QuantViews& v, float* part_buf, uint32_t* part_cnt, hipStream_t stream);

// bool mask -> per-lane bit words + visited-tile lists for fa_fwd16_w64's MASKT instantiations (fa_aux.hip); fills p.mk_*
size_t mask_pack_bytes(const FwdParams& p);
hipError_t launch_mask_pack(FwdParams& p, void* scratch, hipStream_t stream);
// additive fp16 / bf16 / fp32 mask -> per-wave tile classes + visited-tile lists for fa_fwd16_w64's MASKA instantiations (same scratch layout as the bool pack, no
// bit image: the kernel reads the caller's tensor itself -- or the fp16 copy this pass writes of a bf16 / fp32 one); fills p.mk_list / mk_cnt / mk_bs / mk_hs / mk_nrb64.
// fp32: also the exactness verdict word (p.guard, guard_want = 0) and, 256 bytes behind it, the 128-row kernel's tile flags (mask_flags_describe)
// cast != NULL: the arguments of the same call's V cast pass (launch_cast_rows_bf16_to_f16) -- it rides in the classification's launch (the cast's workgroups first, as
// with the bool re-pack below) when its one-launch form applies, and is launched in front of the classification otherwise
struct CastRowsCall {
    const void* src;
    int64_t strides[4];
    void* dst;
    uint32_t B, H, S, D;
    uint32_t* hdr;
};
hipError_t launch_mask_classify(FwdParams& p, void* scratch, hipStream_t stream, const CastRowsCall* cast = nullptr);
bool mask_needs_copy(const FwdParams& p);  // the bias route reads the classification pass's padded fp16 copy of this additive mask (bf16 / fp32: always; fp16: ragged shape or unaligned rows)
size_t mask_copy_bytes(const FwdParams& p);  // bf16 / fp32 masks: the dense fp16 copy the kernel reads (fp32: + exactness bytes, verdict word, 128-row tile flags), behind the pack area (256-byte aligned) in the same scratch block
// the V cast pass and the mask re-pack as ONE launch (the pack's workgroups behind the cast's), then the list kernel
hipError_t launch_cast_rows_and_mask_pack(const void* src, const int64_t* strides, void* dst, uint32_t B, uint32_t H, uint32_t S, uint32_t D,
                                          uint32_t* hdr, FwdParams& p, void* mask_scratch, hipStream_t stream);
// mask tile flags for fa_fwd16's tile early-exit (fa_aux.hip); launch_mask_flags fills p.mask_flags / mf_*
size_t mask_flags_bytes(const FwdParams& p);
void mask_flags_describe(FwdParams& p, const uint8_t* flags);  // fills FwdParams::mf_* / mask_flags for a flag array written elsewhere (fp32 masks: by the classification pass)
bool mask_flags_worthwhile(const FwdParams& p);
// a mask fa_fwd16 would read per score with scalar loads (rows not aligned to four elements, an Skv that is not a multiple of four, strided keys) -> a copy with rows padded to four keys
bool mask_rows_scalar(const FwdParams& p);
size_t mask_realign_bytes(const FwdParams& p);
hipError_t launch_mask_realign(FwdParams& p, void* dst, hipStream_t stream);
hipError_t launch_mask_flags(FwdParams& p, uint8_t* flags, hipStream_t stream);

}  // namespace umfa
