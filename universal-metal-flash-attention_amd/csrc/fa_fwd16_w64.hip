// fa_fwd16_w64.hip -- bf16 / fp16 forward at head_dim 128, "one wave per SIMD" structure (no mask; causal or not;
// Sq >= 256, Skv >= 64); every other case stays on fa_fwd16 (fa_fwd_16.hip).
//
// Replaces the same Metal `attention` forward dispatch as fa_fwd_16.hip (MFABridge.swift:2240-2248, 2525-2541).
//
// Why a second structure: in fa_fwd16 a wave owns 32 query rows, so every 16 MFMAs it re-reads a whole K and V tile
// from LDS and its softmax VALU work runs back to back with its MFMAs (measured: time ~ MFMA + VALU, ~45 % MFMA
// busy).  Here (cdna_hip_programming.md, "4-wave, one-wave-per-SIMD, persistent structure"):
//   * workgroup = 4 waves = 256 query rows; a wave owns 64 rows (two 32-row q-blocks) and the whole 512-register
//     file: O^T (64 x 128 fp32), Q^T and the current K tile's fragments live in accumulator registers that only
//     inline-asm MFMAs / ds_reads touch; the compiler allocates the arch VGPRs (scores, P, V^T fragments).
//   * per 64-key tile one pass of 64 MFMAs: S(i) = K(i) Q^T, then O^T += V(i-1)^T P(i-1)^T; the softmax of tile
//     i-1, the transposed V reads, the K fragment reads of tile i+1 and the row max of tile i are placed in the
//     issue gaps between those MFMAs by tools/gen_w64_body.py (fa_fwd16_w64_body.inc).
//   * deferred max (T13): the reference max moves only when a row max exceeds it by > 2^6; O (in AGPRs) is then
//     rescaled by a rare v_accvgpr_read/mul/write pass after the pending tile's PV.
//   * K/V tiles arrive by LDS-DMA into 2-slot rings (K three tiles ahead, V one), one barrier per tile.
//   * persistent grid (one workgroup per CU): whole rounds of items first (lockstep per XCD, K/V read into its L2
//     once), then the remaining items are shared "stream-K" style, so 384 items on 256 CUs (the FLUX shape) cost
//     1.5 item-times instead of 2; an item cut by a slice boundary is folded by the last of its parts to arrive,
//     in index order (bitwise reproducible), through part_buf.
#include <cstdlib>
#include <type_traits>

#include "fa_common.h"
#include "fa_fwd_16_kernel.h"
#include "kernels.h"

namespace umfa {

struct W64Params {
    const void* q;
    const void* k;
    const void* v;
    void* o;
    float* lse;
    int64_t qs[3], ks[3], vs[3];  // batch, head, seq strides in elements (head_dim contiguous)
    uint32_t B, H, Sq, Skv;
    float scale;
    uint32_t n_items, T;  // items = B*H*(Sq/256) blocks of 256 query rows; T = Skv/64 key tiles per item
    float* part_buf;      // [2 * grid slots][wave 4][q-block 2][chunk 17][lane 64] x 16 bytes (see the kernel)
    uint32_t* part_cnt;   // [n_items % grid] arrival tickets, zero between launches (the folding part resets its own)
    float tau;            // deferred-max threshold (log2 units)
    uint32_t lazy;        // bf16 kernels: lazy reference mode (no row max after a segment's first tile; see the kernel)
    uint32_t skew;        // tiles moved from the folding part of a two-way cut item to the publishing part (see the kernel)
    uint32_t Tw;          // WINDOW kernels: key tiles per item (the band of a 256-row block), <= T
    int32_t win_left, win_right;  // WINDOW kernels: key attends iff row - win_left <= key <= row + win_right
    const float* rope_cos;  // fused rotary embedding of Q (FwdParams::rope_*), NULL = none
    const float* rope_sin;
    int64_t rope_tb;
    const uint32_t* mk_bits;  // MASKT kernels: the packed bool mask (FwdParams::mk_*, fa_aux.hip mask_pack_kernel)
    const uint32_t* mk_list;
    const uint32_t* mk_cnt;
    uint32_t mk_bs, mk_hs, mk_nrb64;
    const uint32_t* mk_prefix;  // [n_items % grid + 1] running sums of the shared blocks' list lengths (fa_aux.hip mask_prefix_kernel)
    const float* vsc;         // bf16pv16 kernels: 2^e of the V image's slabs (FwdParams::vsc), slab (b, h) at vsc[128 (b vsc_bs + h vsc_hs) + 65]
    uint32_t vsc_bs, vsc_hs;
};

// ---- asm-owned accumulator registers: helpers with literal register numbers (generated)
#include "fa_fwd16_w64_regs.inc"

// int8 K tile image: rows of 128 bytes, 16-byte chunks XOR-swizzled for conflict-free ds_read_b128 (same rule as
// fa_quant.hip k8_off<128>)
__device__ __forceinline__ constexpr int k8_off_128(int row, int ch) { return row * 128 + 16 * (ch ^ ((row >> 1) & 7)); }

struct W64I8Params {
    const int8_t* q8;       // [B*H*Sq][128] int8 (quantiser workspace)
    const int8_t* k8;       // [B*H*Skv][128]
    const _Float16* v16;    // [B*H*Skv][128] fp16, de-quantised
    const uint8_t* v8;      // fp8 variant: [B*H][tile][8192] e4m3 in MFMA operand order (fa_quant.hip)
    const uint32_t* v_e8;   // fp8 variant: [B*H][nkblk] E8M0 scale byte of the tile, replicated in the four bytes
    const float* q_scale;   // [B*H][nqblk], one per 64 rows
    const float* k_scale;   // [B*H][nkblk]
    void* o;                // fp32 [B,H,Sq,128]
    float* lse;
    uint32_t B, H, Sq, Skv, nqblk, nkblk;
    float scale;
    uint32_t n_items, T;
    float* part_buf;
    uint32_t* part_cnt;
    float tau;
    uint32_t lazy;          // lazy reference mode (fp16 P thresholds; the fp8 variant has no lazy bodies and ignores it)
    uint32_t skew;
};

#define W64_I8 0
#define W64_BODY_INC "fa_fwd16_w64_body.inc"
#define W64_T __bf16
#define W64_MFMA "v_mfma_f32_32x32x16_bf16"
#define W64_MFMA_QK "v_mfma_f32_32x32x16_bf16"
#define W64_MSUM "v_mfma_f32_4x4x4_16b_bf16"   /* row sums: lane-local sum of four P values against an all-ones operand */
#define W64_ONES_BITS 0x3f803f80u              /* bf16 1.0 twice */
#define W64_LAZY_PARTS 1                        /* lazy reference mode, bf16 P: fp32's exponent range (2 = fp16 P: tighter thresholds) */
#define W64_CVT "v_cvt_pk_bf16_f32"
#define W64_KERNEL fa_fwd16_w64_bf16
#include "fa_fwd16_w64_kernel.inc"
#undef W64_T
#undef W64_MFMA
#undef W64_MFMA_QK
#undef W64_MSUM
#undef W64_ONES_BITS
#undef W64_LAZY_PARTS
#undef W64_CVT
#undef W64_KERNEL

// bf16 Q / K / V with the P V product in fp16 -- the DEFAULT bf16 forward (option pv_fp16): S = K Q^T on the bf16 MFMA, P rounded to fp16 (v_cvt_pk_f16_f32: 11 bits instead of bf16's 8, which
// is what puts the bf16-input forward inside the north-star's 1e-3), O^T += V^T P^T on the fp16 MFMA against an fp16 image of V
// (the runtime's cast pre-pass, fa_aux.hip: V * 2^-e with one power of two per (batch, head) slab, so that no bf16 value leaves fp16's
// range; 2^e comes back in the epilogue's 1 / l: W64_VSC); the lazy reference with fp16's thresholds.
// (Round 4 also built the conversion INSIDE this kernel -- register-staged V tiles, 48 vector instructions + 4 ds_write per tile
// per wave: +23 % cycles per tile, +15 % loop time at the FLUX shape (in-kernel stamps 167.6 vs 145.3 us,
// profiles/r4/lab_notes.md): every workgroup converts every V tile again, 16 x redundantly at FLUX; the pre-pass converts once
// at HBM speed.  An epilogue check of the outputs was dropped too: two more live registers there cost the TILE LOOP 76
// accumulator-register moves and the kernel 8-13 %.)
#define W64_T __bf16
#define W64_MFMA "v_mfma_f32_32x32x16_f16"
#define W64_MFMA_QK "v_mfma_f32_32x32x16_bf16"
#define W64_MSUM "v_mfma_f32_4x4x4_16b_f16"
#define W64_ONES_BITS 0x3c003c00u
#define W64_LAZY_PARTS 2
#define W64_CVT "v_cvt_pk_f16_f32"
#define W64_KERNEL fa_fwd16_w64_bf16pv16
#undef W64_VSC
#define W64_VSC 1
#include "fa_fwd16_w64_kernel.inc"
#undef W64_VSC
#define W64_VSC 0
#undef W64_T
#undef W64_MFMA
#undef W64_MFMA_QK
#undef W64_MSUM
#undef W64_ONES_BITS
#undef W64_LAZY_PARTS
#undef W64_CVT
#undef W64_KERNEL

#define W64_T _Float16
#define W64_MFMA "v_mfma_f32_32x32x16_f16"
#define W64_MFMA_QK "v_mfma_f32_32x32x16_f16"
#define W64_MSUM "v_mfma_f32_4x4x4_16b_f16"
#define W64_ONES_BITS 0x3c003c00u              /* fp16 1.0 twice */
#define W64_LAZY_PARTS 2                        /* fp16 P: the lazy mode with fp16's thresholds (see the kernel) */
#define W64_CVT "v_cvt_pk_f16_f32"
#define W64_KERNEL fa_fwd16_w64_f16
#include "fa_fwd16_w64_kernel.inc"
#undef W64_T
#undef W64_MFMA
#undef W64_CVT
#undef W64_KERNEL
#undef W64_MFMA_QK
#undef W64_BODY_INC

// head_dim 64: the same structure (tools/gen_w64_body.py Cfg(d64=True)): 32 MFMAs per 64-key tile, rows of 128 bytes in the
// tile images, d-blocks 0 and 1 of head_dim 128's O^T register map
#undef W64_DP  /* (the kernel text defaults it to 128) */
#define W64_DP 64
#define W64_BODY_INC "fa_fwd16_w64d64_body.inc"
#undef W64_MSUM
#undef W64_ONES_BITS
#undef W64_LAZY_PARTS
#define W64_T __bf16
#define W64_MFMA "v_mfma_f32_32x32x16_bf16"
#define W64_MFMA_QK "v_mfma_f32_32x32x16_bf16"
#define W64_MSUM "v_mfma_f32_4x4x4_16b_bf16"
#define W64_ONES_BITS 0x3f803f80u
#define W64_LAZY_PARTS 1
#define W64_CVT "v_cvt_pk_bf16_f32"
#define W64_KERNEL fa_fwd16_w64d64_bf16
#include "fa_fwd16_w64_kernel.inc"
#undef W64_T
#undef W64_MFMA
#undef W64_MFMA_QK
#undef W64_MSUM
#undef W64_ONES_BITS
#undef W64_LAZY_PARTS
#undef W64_CVT
#undef W64_KERNEL
/* bf16 operands, fp16 P V against the fp16 image of V: the default bf16 forward at head_dim 64 */
#define W64_T __bf16
#define W64_MFMA "v_mfma_f32_32x32x16_f16"
#define W64_MFMA_QK "v_mfma_f32_32x32x16_bf16"
#define W64_MSUM "v_mfma_f32_4x4x4_16b_f16"
#define W64_ONES_BITS 0x3c003c00u
#define W64_LAZY_PARTS 2
#define W64_CVT "v_cvt_pk_f16_f32"
#define W64_KERNEL fa_fwd16_w64d64_bf16pv16
#undef W64_VSC
#define W64_VSC 1
#include "fa_fwd16_w64_kernel.inc"
#undef W64_VSC
#define W64_VSC 0
#undef W64_T
#undef W64_MFMA
#undef W64_MFMA_QK
#undef W64_MSUM
#undef W64_ONES_BITS
#undef W64_LAZY_PARTS
#undef W64_CVT
#undef W64_KERNEL
#define W64_T _Float16
#define W64_MFMA "v_mfma_f32_32x32x16_f16"
#define W64_MFMA_QK "v_mfma_f32_32x32x16_f16"
#define W64_MSUM "v_mfma_f32_4x4x4_16b_f16"
#define W64_ONES_BITS 0x3c003c00u
#define W64_LAZY_PARTS 2
#define W64_CVT "v_cvt_pk_f16_f32"
#define W64_KERNEL fa_fwd16_w64d64_f16
#include "fa_fwd16_w64_kernel.inc"
#undef W64_T
#undef W64_MFMA
#undef W64_MFMA_QK
#undef W64_CVT
#undef W64_KERNEL
#undef W64_BODY_INC
#undef W64_DP
#undef W64_I8
// (W64_MSUM / W64_ONES_BITS / W64_LAZY_PARTS of the fp16 family stay defined for the int8 kernels below: fp16 P there too)

// runtime-quantised variant: int8 QK^T, fp16 PV
#define W64_I8 1
#define W64_BODY_INC "fa_fwd_w64_i8_body.inc"
#define W64_T _Float16
#define W64_MFMA "v_mfma_f32_32x32x16_f16"
#define W64_MFMA_QK "v_mfma_i32_32x32x32_i8"
#define W64_CVT "v_cvt_pk_f16_f32"
#define W64_KERNEL fa_fwd_w64_i8
#include "fa_fwd16_w64_kernel.inc"
#undef W64_T
#undef W64_MFMA
#undef W64_MFMA_QK
#undef W64_CVT
#undef W64_KERNEL
#undef W64_BODY_INC

// runtime-quantised, fp8 P V (opt-in fast mode, quant_mode 3): int8 QK^T, fp8 e4m3 P and V
#undef W64_F8
#define W64_F8 1
#undef W64_LAZY_PARTS
#define W64_LAZY_PARTS 0                        /* fp8 P tops out at 448: deferred max only */
#define W64_BODY_INC "fa_fwd_w64_i8f8_body.inc"
#define W64_T _Float16
#define W64_MFMA "v_mfma_f32_32x32x16_f16"
#define W64_MFMA_QK "v_mfma_i32_32x32x32_i8"
#define W64_CVT "v_cvt_pk_f16_f32"
#define W64_KERNEL fa_fwd_w64_i8f8
#include "fa_fwd16_w64_kernel.inc"
#undef W64_T
#undef W64_MFMA
#undef W64_MFMA_QK
#undef W64_CVT
#undef W64_KERNEL
#undef W64_F8
#undef W64_I8
#undef W64_BODY_INC


// Softmax reference policy of a launch (kernels.h SoftmaxRef, set by umfa_set_option): tau of the max-chain tile bodies and
// whether the bf16 kernels run their lazy bodies.  Measured accuracy / time of the regimes: DESIGN.md §3.2.
#ifndef W64_LAZY_FP16_DEFAULT
#define W64_LAZY_FP16_DEFAULT 1
#endif
static void w64_softmax_policy(int in_prec, float* tau, uint32_t* lazy) {
    const int mode = tuning().sm_mode.load(std::memory_order_relaxed);
    const float t = tuning().sm_tau.load(std::memory_order_relaxed);
    *tau = mode == SM_EXACT ? 0.0f : t;
    // bf16 P: lazy by default.  fp16 P (fp16 kernels, the int8 kernel): the lazy bodies exist with fp16's thresholds
    *lazy = (mode == SM_LAZY || (mode == SM_DEFAULT && (in_prec == P_BF16 || W64_LAZY_FP16_DEFAULT))) ? 1u : 0u;
}

static int w64_cu_count() { return device_cu_count(); }

// Sliding window without a mask tensor (MK_WINDOW; with `causal` the right edge is the diagonal): the WINDOW instantiations
// sweep, per 256-row block, only the key tiles of its band -- w64_window_tiles of them, the same count for every block.
static bool w64_is_window(const FwdParams& p) { return p.mask_kind == MK_WINDOW; }
// (clamped to values that change nothing: key >= row - left holds for every row once left >= Sq, key <= row + right once
// right >= Skv; the kernel adds them to 32-bit row / key differences)
static uint32_t w64_win_left(const FwdParams& p) { return p.win_left < p.Sq ? p.win_left : p.Sq; }
static uint32_t w64_win_right(const FwdParams& p) { return p.causal ? 0u : (p.win_right < p.Skv ? p.win_right : p.Skv); }
static uint32_t w64_tiles_per_item(const FwdParams& p) {
    const uint32_t T = (p.Skv + 63) / 64;
    if (!w64_is_window(p)) return T;
    const uint64_t band = 256ull + w64_win_left(p) + w64_win_right(p);  // keys a 256-row block can see
    const uint64_t tw = (band + 63) / 64 + 1;                          // + 1: the band need not start on a tile boundary
    return tw < T ? (uint32_t)tw : T;
}

bool fwd_w64_supported(const FwdParams& p) {
    if (tuning().no_w64.load(std::memory_order_relaxed) || !fwd_16_supported(p)) return false;
    if ((p.D != 128 && p.D != 64) || (p.mask_kind != MK_NONE && p.mask_kind != MK_WINDOW && p.mask_kind != MK_BOOL)) return false;
    if (p.mask_kind == MK_BOOL) {
        // bool mask tensors (MASKT instantiations): head_dim 128, the fp16-P-V families (bf16 operands by default, fp16 operands), no
        // causal flag / rotation on top; whole items per workgroup, so at least one item per CU
        if (tuning().no_w64_mask.load(std::memory_order_relaxed) || !p.mask || p.D != 128 || p.causal || p.rope_cos) return false;
        if (p.in_prec == P_BF16 && !p.pv16) return false;
        if (p.Skv < 64 || p.Sq < 256 || (p.Sq % 256 != 0 && p.Sq < 1024) || ((p.Skv + 63) / 64) > 1024u) return false;  // (a block's tile list sits in 4 KiB of LDS)
        if (p.out_prec != P_FP32 && p.out_prec != p.in_prec) return false;
        // at least one block per CU; or -- fewer blocks, every block then shared between workgroups (w64_grid) -- a mask WITHOUT a row
        // dimension (key padding, [B, 1 | H, 1, Skv]) and the unmasked kernel's 10 tile steps per CU.  Measured with fewer blocks than CUs
        // (profiles/r4/mask_w64_few_blocks.jsonl, this kernel / 128-row kernel): padding 1.15-1.49 x, dense random per-head 1.7-2.4 x, but
        // block-diagonal and window TENSORS 0.75-0.96 x (short lists: two parts + a fold per block) -- and how dense a [Sq, Skv] mask is
        // the host cannot know without reading it back
        const uint64_t blocks = (uint64_t)p.B * p.H * ((p.Sq + 255) / 256), cus = (uint64_t)w64_cu_count();
        if (tuning().force_w64.load(std::memory_order_relaxed)) return true;
        if (p.in_prec == P_BF16 && p.Sq < 1024) return false;  // (the fp16 image of V is re-read by too few q-blocks: see below)
        // (1024 <= Sq < 2048, masks shared by every (batch, head) or without a row dimension: B4 H12 S1536 window 78 us against 62, B4 H8 S1536 padding 72 / 65,
        // B1 H64 S1024 window 48 / 44 -- the pass and the per-block prologues against short lists)
        if (p.in_prec == P_BF16 && p.Sq < 2048 && ((p.ms[0] == 0 && p.ms[1] == 0) || p.ms[2] == 0)) return false;
        // (random-size audit, routing_random_masks_*.jsonl: with WHOLE blocks on `blocks` workgroups -- nothing cut, some CUs idle -- DENSE masks with a row
        // dimension win here too from three eighths of a block per CU (random per-head masks B4 H6 S1536 68 us against 112, B2 H4 S4096 139 / 162), sparse
        // structured ones lose as much (block-diagonal, 8 documents, B4 H6 S1536 47 / 30).  The host cannot see the density; it can see the shape: a mask with
        // a batch or head dimension of its own is not a window or a document mask, one shared by every (batch, head) usually is)
        return blocks >= cus || (p.ms[2] == 0 && blocks * ((p.Skv + 63) / 64) >= cus * 10) || ((p.ms[0] != 0 || p.ms[1] != 0) && blocks * 8 >= cus * 3);
    }
    if (w64_is_window(p) && p.rope_cos) return false;  // window instantiations: no fused rotation
    if (p.D == 64 && p.rope_cos) return false;  // the fused Q rotation exists at head_dim 128 only
    // rows are processed in blocks of 256: a ragged last block wastes its empty waves, so small ragged Sq stay on
    // the 128-row kernel; any Skv >= 64 works (a partial last key tile runs the masking variant of the tile body)
    if (p.Skv < 64 || p.Sq < 256 || (p.Sq % 256 != 0 && p.Sq < 1024)) return false;
    if (p.out_prec != P_FP32 && p.out_prec != p.in_prec) return false;
    // Enough parallel work for one workgroup per CU, else the 128-row kernel (2-3 resident workgroups, finer items)
    // is faster.  Same-box medians, us, w64 / 128-row: causal B4 H16 S1024 (128 jobs) 53 / 46, B1 H16 S2048 (64) 70 / 53,
    // B1 H8 S4096 (64) 112 / 96, B1 H8 S8192 (128) 213 / 214, B2 H24 S2048 (192) 85 / 98, B8 H16 S1024 (256) 61 / 85;
    // non-causal B1 H24 S1024 (6 tile steps per CU) 38 / 32, B1 H4 S4096 (16 per CU) 54 / 75.  UMFA_FORCE_W64=1 lifts it
    // (parity tests on small shapes).
    if (!tuning().force_w64.load(std::memory_order_relaxed)) {
        const uint64_t cus = (uint64_t)w64_cu_count();
        const uint64_t nqb = (p.Sq + 255) / 256;
        // bf16 operands with the fp16 P V product: this kernel needs the fp16 image of V (a cast pre-pass: two more passes over V), and with
        // nqb q-blocks per head the pass costs ~1 / nqb of the kernel's own time.  The 128-row kernel converts V in place in LDS for ~12 %.
        // Graph-replayed us, this kernel + pass / 128-row kernel (profiles/r4/small_nqb_probe.jsonl): Sq 256: 444 / 353, 30 / 22, 254 / 201 (D 64);
        // Sq 512: 331 / 327, 66 / 54, 40 / 31; Sq 768: 131 / 125, causal 54 / 51; Sq 1024: 49 / 51.
        // Sq 1024 ... 2048 (gate_probe.jsonl): head_dim 128, whole blocks 1.04-1.22 x the 128-row kernel; ragged Sq 1100 0.81 x; causal with an odd
        // number of q-blocks (Sq 1280: the middle block has no mirror) 0.89 x; head_dim 64 at Sq 1024 0.96-0.99 x
        // Long key ranges shift it (random-size audit, routing_random_bf16_before.jsonl): with 64 or more key tiles per item this kernel's lead over the
        // 128-row kernel is 1.4 x and pays for the pass from three q-blocks on -- B8 H5 Sq768 Skv8192 148 against 182 us, head_dim 64 B2 H12 Sq1536
        // Skv8192 104 against 147
        const bool long_keys = !p.causal && (p.Skv + 63) / 64 >= 64;
        if (p.in_prec == P_BF16 && p.pv16) {
            if (p.Sq < 768 || (p.Sq < 1024 && !(long_keys && p.D == 128))) return false;
            if (p.Sq < 2048 && (p.Sq % 256 != 0 || (p.D == 64 && !(long_keys && p.Sq >= 1024)) || (p.causal && (nqb & 1)))) return false;
        }
        // (causal with an odd, small number of q-blocks -- the middle block has no mirror, its job is half as long as the others -- for every operand type:
        // fp16 B4 H32 S1280 104 us against 90 on the 128-row kernel)
        if (p.causal && !w64_is_window(p) && (nqb & 1) && nqb < 8) return false;
        // (key ranges of a few tiles: an item is mostly prologue and output -- fp16 B2 H128 Sq3072 Skv256, 12 whole rounds: 205 us against 186; head_dim 64
        // B8 H12 Sq2048 Skv77 29.0 / 23.4)
        if (!w64_is_window(p) && (p.Skv + 63) / 64 < (p.D == 64 ? 16u : 8u)) return false;  // (head_dim 64, 8 tiles, nine whole rounds: 123 / 112)
        if (w64_is_window(p)) {
            // the band's tile steps are what there is to share (thresholds of the unmasked kernel: cut items need 10 steps per CU)
            const uint64_t items = (uint64_t)p.B * p.H * nqb, steps = items * w64_tiles_per_item(p);
            if (items % cus != 0 && steps < cus * 10) return false;
        } else if (p.D == 64) {
            // half the MFMA time per tile step, the same prologue / drain / fold: the break-even sits higher.  us, w64 / 128-row
            // (profiles/r3/d64_w64_vs_128row.jsonl): causal 160 jobs 31.1 / 27.6, 192 jobs 45.9 / 46.6, 256 jobs 32.4 / 36.4,
            // 512 jobs 91.7 / 107; non-causal cut items 10 steps per CU 27.2 / 23.4, 12: 27.0 / 23.6, 16: 30.4 / 30.0,
            // 96: 109 / 122; whole rounds B8 H16 S1024 42.1 / 46.2
            const uint64_t items = (uint64_t)p.B * p.H * nqb, steps = items * ((p.Skv + 63) / 64);
            // (round 4, with the V cast pass in the launch: 24 steps per CU -- B1 H16 S2048: 34.5 against 31.4 us on the 128-row kernel)
            // (causal, bf16 with the cast pass: 1.5 jobs per CU -- few_items_probe_causal.jsonl: 192 jobs 61.0 / 49.9 us at S 2048, 90.5 / 90.2 at S 4096)
            // (causal jobs are whole -- no cut: a last round that is mostly empty is paid in full.  More than a quarter of the rounds' slots empty and
            // the 128-row kernel wins at head_dim 64: B1 H34 S8192 (2.1 rounds) 442 against 407 us, fp16 B8 H12 S3072 (2.25) 196 / 166, B1 H32 S6144 (1.5) 235 / 223)
            if (p.causal) {
                const uint64_t jobs = (uint64_t)p.B * p.H * ((nqb + 1) / 2), rounds = (jobs + cus - 1) / cus;
                if (rounds >= 2 && rounds * cus * 4 > jobs * 5) return false;  // (a single partly filled round is fine: B4 H3 S8192, 192 jobs, 151 us here / 200)
                if ((nqb & 1) && nqb < 16) return false;                        // (odd q-block counts: the unpaired middle block -- B8 H16 S2304 176 / 148)
            }
            // (second pass: long jobs amortise the pass and the prologues -- 240 jobs of 24 + 24 q-blocks (S 6144) 127 us here against 168; so: 1.5 jobs
            // per CU, or 0.9 per CU, or 0.75 per CU with twenty or more q-blocks per head)
            const uint64_t cjobs = (uint64_t)p.B * p.H * ((nqb + 1) / 2);
            const bool c_ok = !(p.in_prec == P_BF16 && p.pv16) ? cjobs * 4 >= cus * 3 : (cjobs * 4 >= cus * 6 || cjobs * 10 >= cus * 9 || (nqb >= 20 && cjobs * 4 >= cus * 3));
            if (p.causal ? !c_ok : (items % cus != 0 && (steps < cus * 20 || (p.Skv + 63) / 64 < 32 || (items > cus && (p.Skv + 63) / 64 < 40)))) return false;  // (20: fp16 B2 H10 S2048 37.6 here / 39.7; a remainder of SHORT items behind whole rounds: fp16 B4 H10 S2304 (36 tiles) 78 / 70 -- long ones win: B4 H6 Sq3072 Skv8192 183 / 221)  // (fp16 operands too: routing_sweep_fp16.jsonl, B1 H16 S2048 30.1 against 26.9 us; cut items of fewer than 32 tiles: fp16 B8 H8 S1280 47.4 / 42.1)
        } else if (p.causal) {
            if ((uint64_t)p.B * p.H * ((nqb + 1) / 2) * 8 < cus * 5) return false;  // 160 jobs: 78 / 85, 126 / 150, 217 / 258 us
            // (whole jobs, no cut: from the second round on a mostly empty last round is paid in full -- B1 H64 S2304, 320 jobs = 1.25 rounds: 171 us
            // against 140 on the 128-row kernel)
            const uint64_t jobs = (uint64_t)p.B * p.H * ((nqb + 1) / 2), rounds = (jobs + cus - 1) / cus;
            if (rounds >= 2 && rounds * cus * 2 > jobs * 3) return false;
        } else {
            // whole rounds (every workgroup one or more complete items, nothing to fold) win at any size: B1 H256 S256
            // 19 / 22 us, B1 H128 S512 27 / 32; cut items need 10 tile steps per CU, 8 with long key ranges
            // (B1 H32 S1024: 35 / 33, B1 H8 S2048: 39 / 44, B1 H40 S1024: 44 / 46, B1 H96 S512: 35 / 30)
            const uint64_t items = (uint64_t)p.B * p.H * nqb, steps = items * ((p.Skv + 63) / 64);
            // (round 4, bf16 operands with the V cast pass in the launch: 12 steps per CU -- B1 H8 S2048, 8 per CU: 41.8 against 35.3 us)
            // (fp16 operands, no pass, lose the same launches by 7 %: routing_sweep_fp16.jsonl -- one threshold for both)
            // (second pass, random-size audit: 10, not 12 -- B8 H1 S2304, 10.1 steps per CU: 41 us here against 50 on the 128-row kernel (fp16), 45 / 48 (bf16))
            // (not when the launch runs WHOLE items on `items` workgroups (w64_grid): nothing is cut then -- fp16 B8 H6 S768, 144 items of 12 tiles: 26.5 us
            // here against 30.3)
            const bool whole = items < cus && items * 2 > cus && 2 * ((p.Skv + 63) / 64) * (cus - items) < 35 * cus;  // (as w64_grid decides)
            if (!whole && items % cus != 0 && steps < cus * 10) return false;

            // short key ranges (fewer than 16 tiles per item): a cut item is a few tiles and a fold -- B1 H24 Sq4096 Skv512 (8 tiles, 12 steps per CU)
            // 40.0 us fp16 / 43.0 bf16 against 36.5 / 41.6 on the 128-row kernel
            if (!whole && items % cus != 0 && (p.Skv + 63) / 64 < 16 && steps < cus * 24) return false;
        }
    }
    return true;
}

static uint32_t w64_grid(const FwdParams& p) {
    if (p.mask_kind == MK_BOOL) {  // mask tensors: one workgroup per CU at most; whole blocks in rounds, the last n % grid blocks cut along their tile lists
        const uint64_t items = (uint64_t)p.B * p.H * ((p.Sq + 255) / 256);
        const uint64_t cus = (uint64_t)w64_cu_count();
        if (const int gi = tuning().w64_grid.load(std::memory_order_relaxed)) {  // lab / tests: force the number of workgroups
            if (gi > 0 && (uint64_t)gi <= cus && (uint64_t)gi <= items && gi <= 512) return (uint32_t)gi;
        }
        const uint64_t gmax = cus < 512 ? cus : 512;  // (the shared blocks' running sums sit in 2 KiB of LDS)
        if (items < cus) return (uint32_t)((p.ms[2] == 0 && items * ((p.Skv + 63) / 64) >= cus * 10) ? gmax : items);  // few blocks, key-padding mask: every block shared
        return (uint32_t)gmax;
    }
    const bool pairs = p.causal && !w64_is_window(p);  // (a causal window is a window with right = 0: linear schedule)
    uint64_t total = (uint64_t)p.B * p.H * ((p.Sq + 255) / 256) * w64_tiles_per_item(p);  // (item, key tile) steps
    if (pairs) total = (uint64_t)p.B * p.H * (((p.Sq + 255) / 256 + 1) / 2);  // jobs = mirrored pairs of q-blocks
    const uint32_t cus = (uint32_t)w64_cu_count();
    if (const int gi = tuning().w64_grid.load(std::memory_order_relaxed)) {  // lab: force the number of workgroups
        const uint32_t g = (uint32_t)gi;
        if (gi > 0 && g <= cus && g <= total) return g;
    }
    if (!pairs) {
        // few items (the strong-scaling shards of a problem: B1 H3 S4096 = 48 items): a whole number of EQUAL parts per item
        // (grid = items x floor(CUs / items): the slices total * w / G then start and end on part boundaries, every
        // workgroup has one segment and one prologue) instead of CUs ragged slices.  Graph-replayed us, aligned / ragged:
        // H3 47.3 / 51.7, H6 71.5 / 72.7, H12 (one part: whole items) 109.1 / 110.0.
        const uint64_t items = (uint64_t)p.B * p.H * ((p.Sq + 255) / 256);
        if (items < cus && cus / items >= 2 && items * (cus / items) <= total) return (uint32_t)(items * (cus / items));
        // more than half an item per CU: whole items on `items` workgroups (some CUs idle) against cutting every item for all CUs and folding
        // it -- the cut costs ~28 us of prologues and folds, the idle CUs T (1 - items / CUs) tile steps of ~1.6 us.  Graph-replayed us, whole /
        // cut / 128-row kernel (profiles/r4/grid_probe.jsonl, grid_probe_more.jsonl): S 2048 (T = 32): 144 items 56 / 59 / 70, 176: 58 / 66 / 72, 216
        // (S 2304): 73 / 86 / 85; S 4096 (T = 64): 176 items 100 / 102 / 133, 192: 106 / 106 / 137, 240: 123 / 135 / 149; head_dim 64, 216 items 48 / 54 / 51
        if (items < cus && 2 * (uint64_t)w64_tiles_per_item(p) * (cus - items) < 35 * cus) return (uint32_t)items;
    }
    return total < cus ? (uint32_t)total : cus;  // never more workgroups than steps: every slice is non-empty
}

uint32_t fwd_w64_grid(const FwdParams& p) { return w64_grid(p); }

FwdW64Plan fwd_w64_plan(const FwdParams& p) {
    FwdW64Plan plan;
    const uint32_t items = p.B * p.H * ((p.Sq + 255) / 256);
    (void)items;
    plan.cnt_bytes = ((size_t)w64_grid(p) * sizeof(uint32_t) + 255) & ~(size_t)255;  // < grid shared items (causal: none)
    plan.buf_bytes = (size_t)2 * w64_grid(p) * (4 * 2 * 17 * 1024);
    return plan;
}

template <typename KFN>
static hipError_t launch_w64_kernel(KFN kfn, const FwdParams& p, const W64Params& wp, hipStream_t stream) {
    const uint32_t grid = w64_grid(p);
    // K/V rings + per-wave output staging + flag / ticket words + the mask kernels' bit-word ring, tile list and shared-block running sums
    const size_t lds = 65536 + 4 * 32 * (512 + 16) + 64 + 4096 + 8192 + 2048 + 64;
    if (hipError_t e = ensure_dynamic_lds((const void*)kfn, lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds, stream, wp);
    return hipGetLastError();
}

// part_cnt must be zero on entry (the runtime zeroes it when it allocates; the kernel leaves it zero).
hipError_t launch_fwd_w64(const FwdParams& p, float* part_buf, uint32_t* part_cnt, hipStream_t stream, const char** name) {
    if (!fwd_w64_supported(p) || !part_buf || !part_cnt) return hipErrorNotSupported;
    W64Params wp;
    wp.q = p.q; wp.k = p.k; wp.v = p.v; wp.o = p.o; wp.lse = p.lse;
    for (int i = 0; i < 3; ++i) { wp.qs[i] = p.qs[i]; wp.ks[i] = p.ks[i]; wp.vs[i] = p.vs[i]; }
    wp.B = p.B; wp.H = p.H; wp.Sq = p.Sq; wp.Skv = p.Skv;
    wp.scale = p.scale;
    wp.n_items = p.B * p.H * ((p.Sq + 255) / 256);
    wp.T = (p.Skv + 63) / 64;
    wp.part_buf = part_buf;
    wp.part_cnt = part_cnt;
    w64_softmax_policy(p.in_prec, &wp.tau, &wp.lazy);
    wp.skew = (uint32_t)tuning().w64_skew.load(std::memory_order_relaxed);
    // (skew = 0xffffffff would run the remainder of a multi-round launch as one more round of whole items (kernel: whole_rem).  Measured,
    // whole / cut: 464 items S 4096 242 / 235, 480 items S 2048 138 / 142, 448 items 134 / 130 -- a wash, so the launcher never asks for it;
    // lab option w64_skew = -1)
    if (tuning().w64_skew.load(std::memory_order_relaxed) == -1) wp.skew = 0xffffffffu;
    wp.rope_cos = p.rope_cos; wp.rope_sin = p.rope_sin; wp.rope_tb = p.rope_tb;
    wp.vsc = p.vsc; wp.vsc_bs = p.vsc_bs; wp.vsc_hs = p.vsc_hs;
    if (p.in_prec == P_BF16 && p.pv16 && !p.vsc) return hipErrorInvalidValue;  // the fp16 image of V comes with its slab exponents (runtime.hip)
    wp.Tw = wp.T; wp.win_left = wp.win_right = 0;
    const bool rope = p.rope_cos != nullptr, window = w64_is_window(p), fp32o = p.out_prec == P_FP32, maskt = p.mask_kind == MK_BOOL;
    if (maskt) {
        if (!p.mk_bits || !p.mk_list || !p.mk_cnt) return hipErrorInvalidValue;  // runtime.hip packs the mask first (launch_mask_pack)
        wp.mk_bits = p.mk_bits; wp.mk_list = p.mk_list; wp.mk_cnt = p.mk_cnt;
        wp.mk_bs = p.mk_bs; wp.mk_hs = p.mk_hs; wp.mk_nrb64 = p.mk_nrb64;
        wp.mk_prefix = p.mk_prefix;
        if (wp.n_items % w64_grid(p) != 0 && !p.mk_prefix) return hipErrorInvalidValue;
        // the max chain, unless the mask has no row dimension (key padding: a listed tile holds a key for every row, the lazy bodies are as safe as
        // without a mask); with one, which rows have keys in a segment is not arithmetic
        if (p.ms[2] != 0 || tuning().no_w64_mask_lazy.load(std::memory_order_relaxed)) wp.lazy = 0;
    }
    if (rope && p.out_prec != p.in_prec) return hipErrorNotSupported;  // fused-RoPE instantiations: O in the operand type only (runtime.hip asks first)
    if (window) {
        wp.Tw = w64_tiles_per_item(p);
        wp.win_left = (int32_t)w64_win_left(p);
        wp.win_right = (int32_t)w64_win_right(p);
    }
    // one kernel family = one (operand type, P V type, head_dim); its instantiations: <OUT, causal>, <OUT, false, false, window>,
    // and at head_dim 128 <operand type, causal, rope>
#define W64_FAMILY(FAM, T16, HAS_ROPE)                                                                                         \
    do {                                                                                                                       \
        if constexpr (HAS_ROPE == 2) {  /* families with mask-tensor instantiations */                                        \
            if (maskt) return fp32o ? launch_w64_kernel(FAM<float, false, false, false, true>, p, wp, stream) : launch_w64_kernel(FAM<T16, false, false, false, true>, p, wp, stream); \
        }                                                                                                                      \
        if constexpr (HAS_ROPE != 0) {                                                                                         \
            if (rope) return p.causal ? launch_w64_kernel(FAM<T16, true, true>, p, wp, stream) : launch_w64_kernel(FAM<T16, false, true>, p, wp, stream); \
        }                                                                                                                      \
        if (window) return fp32o ? launch_w64_kernel(FAM<float, false, false, true>, p, wp, stream) : launch_w64_kernel(FAM<T16, false, false, true>, p, wp, stream); \
        if (p.causal) return fp32o ? launch_w64_kernel(FAM<float, true>, p, wp, stream) : launch_w64_kernel(FAM<T16, true>, p, wp, stream); \
        return fp32o ? launch_w64_kernel(FAM<float, false>, p, wp, stream) : launch_w64_kernel(FAM<T16, false>, p, wp, stream); \
    } while (0)
    static const char* const names[3][2][3] = {
        {{"fa_fwd16_w64<bf16,128>", "fa_fwd16_w64<bf16,128,rope>", "fa_fwd16_w64<bf16,128,window>"},
         {"fa_fwd16_w64<bf16,64>", "fa_fwd16_w64<bf16,64,rope>", "fa_fwd16_w64<bf16,64,window>"}},
        {{"fa_fwd16_w64<bf16,128,pv16>", "fa_fwd16_w64<bf16,128,pv16,rope>", "fa_fwd16_w64<bf16,128,pv16,window>"},
         {"fa_fwd16_w64<bf16,64,pv16>", "fa_fwd16_w64<bf16,64,pv16,rope>", "fa_fwd16_w64<bf16,64,pv16,window>"}},
        {{"fa_fwd16_w64<fp16,128>", "fa_fwd16_w64<fp16,128,rope>", "fa_fwd16_w64<fp16,128,window>"},
         {"fa_fwd16_w64<fp16,64>", "fa_fwd16_w64<fp16,64,rope>", "fa_fwd16_w64<fp16,64,window>"}}};
    if (p.D == 64 && rope) return hipErrorNotSupported;
    const int fam = p.in_prec == P_BF16 ? (p.pv16 ? 1 : 0) : 2;
    if (maskt && (fam == 0 || p.D != 128)) return hipErrorNotSupported;
    *name = maskt ? (fam == 1 ? "fa_fwd16_w64<bf16,128,pv16,mask>" : "fa_fwd16_w64<fp16,128,mask>") : names[fam][p.D == 64 ? 1 : 0][rope ? 1 : window ? 2 : 0];
    if (p.D == 64) {
        if (fam == 0) W64_FAMILY(fa_fwd16_w64d64_bf16, __bf16, 0);
        if (fam == 1) W64_FAMILY(fa_fwd16_w64d64_bf16pv16, __bf16, 0);
        W64_FAMILY(fa_fwd16_w64d64_f16, _Float16, 0);
    }
    if (fam == 0) W64_FAMILY(fa_fwd16_w64_bf16, __bf16, 1);
    if (fam == 1) W64_FAMILY(fa_fwd16_w64_bf16pv16, __bf16, 2);
    W64_FAMILY(fa_fwd16_w64_f16, _Float16, 2);
#undef W64_FAMILY
}

// ---- runtime-quantised variant ---------------------------------------------------------------------------------
bool fwd_w64_i8_supported(const FwdParams& p) {
    if (tuning().no_w64.load(std::memory_order_relaxed) || p.D != 128 || p.mask_kind != MK_NONE || p.mask) return false;
    if (!(p.scale > 0.0f)) return false;
    if (p.Skv < 64 || p.Sq < 256 || (p.Sq % 256 != 0 && p.Sq < 1024)) return false;
    return true;
}

hipError_t launch_fwd_w64_i8(const FwdParams& p, const QuantViews& v, float* part_buf, uint32_t* part_cnt, hipStream_t stream) {
    if (!fwd_w64_i8_supported(p) || !part_buf || !part_cnt || v.dpq != 128) return hipErrorNotSupported;
    W64I8Params wp;
    wp.q8 = v.q8; wp.k8 = v.k8; wp.v16 = (const _Float16*)v.v16;
    wp.v8 = v.v8; wp.v_e8 = v.v_e8;
    wp.q_scale = v.q_scale; wp.k_scale = v.k_scale;
    wp.o = p.o; wp.lse = p.lse;
    wp.B = p.B; wp.H = p.H; wp.Sq = p.Sq; wp.Skv = p.Skv; wp.nqblk = v.nqblk; wp.nkblk = v.nkblk;
    wp.scale = p.scale;
    wp.n_items = p.B * p.H * ((p.Sq + 255) / 256);
    wp.T = (p.Skv + 63) / 64;
    wp.part_buf = part_buf;
    wp.part_cnt = part_cnt;
    w64_softmax_policy(P_FP16, &wp.tau, &wp.lazy);  // fp16 P
    wp.skew = (uint32_t)tuning().w64_skew.load(std::memory_order_relaxed);
    const uint32_t grid = w64_grid(p);
    const size_t lds = 65536 + 4 * 32 * (512 + 16) + 16;
    const bool f8 = v.v8 != nullptr;
    auto kfn = f8 ? (p.causal ? fa_fwd_w64_i8f8<float, true> : fa_fwd_w64_i8f8<float, false>)
                  : (p.causal ? fa_fwd_w64_i8<float, true> : fa_fwd_w64_i8<float, false>);
    if (hipError_t e = ensure_dynamic_lds((const void*)kfn, lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds, stream, wp);
    return hipGetLastError();
}

}  // namespace umfa
