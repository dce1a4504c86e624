// fa_fwd16_w64_params.h -- what the translation units of the one-wave-per-SIMD forward share: the kernels' parameter blocks and the generated helpers that
// address the asm-owned registers (fa_fwd16_w64.hip: every family but the additive-mask ones; fa_fwd16_w64_bias.hip: those, with a body file of their own)
#pragma once
#include <cstdint>

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
    const void* mask;         // MASKA kernels: the caller's additive fp16 mask tensor, read in place (tile classes / lists: mk_list, mk_cnt)
    int64_t mask_s[3];        // its batch, head and row strides in elements (0 = broadcast; keys contiguous)
    const float* vsc;         // bf16pv16 kernels: 2^e of the V image's slabs (FwdParams::vsc), slab (b, h) at vsc[128 (b vsc_bs + h vsc_hs) + 65]
    uint32_t vsc_bs, vsc_hs;
    const uint32_t* guard;    // MASKA kernels: FwdParams::guard (the launch runs iff NULL or *guard == 0: an fp32 mask whose fp16 copy is exact)
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
    const uint32_t* mk_bits;  // MASKT instantiation: the packed bool mask, as in W64Params (FwdParams::mk_*, fa_aux.hip mask_pack_kernel)
    const uint32_t* mk_list;
    const uint32_t* mk_cnt;
    uint32_t mk_bs, mk_hs, mk_nrb64;
    const float* vsc;       // slab headers of the fp16 V image q * s * 2^-e (fa_quant.hip QuantParams::vhdr): 2^e in word 65 of slab bh; NULL = 1
};

// the additive-mask (MASKA) instantiations live in fa_fwd16_w64_bias.hip: family 1 = bf16 operands with fp16 P V, 2 = fp16 operands
hipError_t launch_fwd_w64_bias(const W64Params& wp, int family, bool fp32_out, uint32_t grid, size_t lds, hipStream_t stream, int head_dim);

}  // namespace umfa
