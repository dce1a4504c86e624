// fa_bwd_16.hip -- bf16 / fp16 MFMA backward for gfx950, head_dim 128 (the FLUX / config-3 shape), 64 and 256.
//
// Same contract as fa_bwd.hip (mfa_attention_backward, MFABridge.swift:3171-3282: D = rowsum(dO o O), dQ, dK, dV in
// fp32, two dispatches "backward query" then "backward key-value", no atomics) with the five products on
// v_mfma_f32_32x32x16_{bf16,f16} instead of the fp32 MFMA (1/16 of the rate).  Orientations follow
// cdna_hip_programming.md "Attention backward": every second product takes the first one's accumulator,
// rounded to the input type, as its B operand with no cross-lane movement.
//
//   bwd16_dq    workgroup = 4 waves x 32 query rows; sweeps 32-key tiles of K and V (LDS, LDS-DMA staged).
//               lane <-> query:   S^T = K Q^T,  dP^T = V dO^T,  P^T = exp2(c S^T - L2[q]),  dS^T = P^T o (dP^T - D[q]),
//               dQ^T += K^T dS^T  (K^T fragments by transposed reads of the SAME K image).
//   bwd16_dkdv  workgroup = 4 waves x 32 keys (K, V fragments in registers); sweeps 32-row tiles of Q and dO (LDS).
//               lane <-> key:     S = Q K^T,  dP = dO V^T,  P, dS as above (row constants from LDS),
//               dV^T += dO^T P,  dK^T += Q^T dS  (Q^T / dO^T fragments by transposed reads of the same images).
// All four tile kinds use ONE dual-use LDS image (2*D-byte rows, 16-byte chunks XOR-swizzled with
// ((row&3)<<2 | (row>>2)&3) at D = 128, ((row>>2)&3 | ((row>>1)&1)<<2) at D = 64): conflict-free for ds_read_b128 row
// reads AND ds_read_b64_tr_b16 transposed reads (tools/lds_bank_check.py).  Tiles arrive by LDS-DMA with the swizzle
// on the source chunk.
// P and dS are rounded to the input type before their second product, like P in the forward.
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "fa_common.h"
#include "fa_fwd_16_kernel.h"  // Mma16<T>, xcd_remap
#include "kernels.h"

namespace umfa {

namespace {

// geometry of one head_dim: 16-key MFMA steps, 32-column d-blocks, 32-row tiles of 2*DP-byte rows
#define BWD16_GEO(DP)                                                                    \
    constexpr int ROW_B = 2 * DP, NKS = DP / 16, NDB = DP / 32, TILE_BYTES = 32 * ROW_B; \
    constexpr int TILE_PIECES = TILE_BYTES / 1024;                                       \
    constexpr int PD = DP >= 128 ? 4 : 2; /* k-steps of LDS row fragments in flight ahead of their MFMAs */ \
    constexpr int NH = DP == 256 ? 2 : 1; /* dkdv: passes over the query range, each owning NDB / NH d-blocks of dK, dV */ \
    [[maybe_unused]] constexpr int NDBH = NDB / NH

template <int DP>
__device__ __forceinline__ constexpr int d_off(int row, int ch) {
    static_assert(DP == 256 || DP == 128 || DP == 64, "swizzles exist for 512-, 256- and 128-byte rows");
    // rows of 256 and 512 bytes all start at bank 0, so they share one swizzle (on the low four chunk-index bits)
    const int f = DP >= 128 ? (((row & 3) << 2) | ((row >> 2) & 3)) : (((row >> 2) & 3) | (((row >> 1) & 1) << 2));
    return 2 * DP * row + 16 * (ch ^ f);
}

__device__ __forceinline__ i32x4 make_srd(const void* base, uint32_t bytes) {
    const unsigned long long a = (unsigned long long)base;
    i32x4 d;
    d[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
    d[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
    d[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    d[3] = 0x00020000;
    return d;
}

// LDS-DMA of `npieces` 1-KiB pieces (4 or 8 rows each) of a [rows][2*DP B] slab image starting at global row `row0`.
// Piece n goes to lds_dst + n KiB; wave w issues pieces w, w+4, ...  Rows past the slab are range-checked away.
template <int NPIECES, int DP>
__device__ __forceinline__ void dma_rows(const i32x4& srd, unsigned lds_dst, uint32_t row0, int uw, int lane) {
    constexpr int ROW_B = 2 * DP, NCH = DP / 8, RPP = 1024 / ROW_B;  // chunks per row, rows per piece
    const int r = lane / NCH, c = lane % NCH;
#pragma unroll
    for (int n0 = 0; n0 < NPIECES; n0 += 4) {
        const int n = n0 + uw;
        if (n < NPIECES) {
            const int row = RPP * n + r;
            const int voff = (int)(row0 + row) * ROW_B + (d_off<DP>(row, c) - row * ROW_B);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                         ::"s"(lds_dst + n * 1024), "v"(voff), "s"(srd) : "memory");
        }
    }
}

// The same with the lane-dependent part of every piece's source offset computed ONCE (dma_lane_offsets, before the tile
// loop): per tile one scalar multiply and one vector add per piece are left.  As dma_rows inside the loop the row / chunk /
// swizzle arithmetic of all pieces was redone for every tile: ~540 of ~3600 cycles per tile of bwd16_dkdv (phase stamps,
// tools/lab/bwd_stamps.py).
template <int NPIECES, int DP>
__device__ __forceinline__ void dma_lane_offsets(int (&off)[(NPIECES + 3) / 4], int uw, int lane) {
    constexpr int NCH = DP / 8, RPP = 1024 / (2 * DP);
    const int r = lane / NCH, c = lane % NCH;
#pragma unroll
    for (int i = 0; i < (NPIECES + 3) / 4; ++i) {
        const int row = RPP * (4 * i + uw) + r;
        off[i] = d_off<DP>(row, c);  // = row * ROW_B + 16 * (c ^ swizzle(row))
    }
}
template <int NPIECES, int DP>
__device__ __forceinline__ void dma_rows_pre(const i32x4& srd, unsigned lds_dst, uint32_t row0, int uw, const int (&off)[(NPIECES + 3) / 4]) {
    constexpr int ROW_B = 2 * DP;
    const int base = (int)row0 * ROW_B;
#pragma unroll
    for (int i = 0; i < (NPIECES + 3) / 4; ++i) {
        const int n = 4 * i + uw;
        if (n < NPIECES) {
            const int voff = base + off[i];
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                         ::"s"(lds_dst + n * 1024), "v"(voff), "s"(srd) : "memory");
        }
    }
}

// one 1-KiB piece of an LDS-DMA tile (see dma_rows_pre)
__device__ __forceinline__ void dma_piece(const i32x4& srd, unsigned lds_dst, int voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_dst), "v"(voff), "s"(srd) : "memory");
}

// transposed-read fragment: rows (row0 .. +3) and (row0+8 .. +11) x 16 columns of d-block i, as the A operand whose
// element j is image row 16 s + 8 (j>>2) + 4 hi + (j&3) (the k order of an accumulator used as B operand)
template <typename M, int DP>
__device__ __forceinline__ typename M::V8 tr_frag(const char* img, int i, int s, int hi, int tr_qq, int tr_pp, int tr_g1) {
    const int ch = 4 * i + 2 * tr_g1 + (tr_pp >> 1);
    const int r0 = 16 * s + 4 * hi + tr_qq;
    const typename M::V4 lo = M::tr_read(img + d_off<DP>(r0, ch) + 8 * (tr_pp & 1));
    const typename M::V4 hi4 = M::tr_read(img + d_off<DP>(r0 + 8, ch) + 8 * (tr_pp & 1));
    return __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
}

// four consecutive gradient elements: fp32 (ABI contract) or rounded once to the input type T
template <typename T>
__device__ __forceinline__ void store_grad4(void* base, int64_t elem, f32x4 val, bool in_type) {
    if (in_type) {
        typedef T T4 __attribute__((ext_vector_type(4)));
        *(T4*)((T*)base + elem) = T4{(T)val[0], (T)val[1], (T)val[2], (T)val[3]};
    } else {
        *(f32x4*)((float*)base + elem) = val;
    }
}

// Causal work items differ in length; with two workgroups per CU they finish together only if the lengths on a CU add
// up alike.  Same order as the forward (fa_fwd_16_kernel.h, where it was measured): consecutive items are a mirrored
// pair of blocks, and the pair 32 items (= CUs per XCD) further on has its long and short member swapped.
// Returns the block's rank by length (0 = longest) and its (batch, head).  Only when every workgroup of the launch is
// resident at once (two per CU): with more workgroups than slots the dispatcher refills slots as they free up and
// plain longest-first order is the better schedule (B1 H16 S8192 causal backward: 1.52 ms vs 1.92 ms paired).
__device__ __forceinline__ uint32_t causal_rank(uint32_t item, uint32_t nblk, uint32_t& bh, bool two_per_cu) {
    if ((nblk & 1) || !two_per_cu || gridDim.x > 512) { bh = item / nblk; return item % nblk; }
    const uint32_t pi = item >> 1, h2 = nblk >> 1, j = pi % h2;
    bh = pi / h2;
    return (((item & 1) ^ (item >> 5)) & 1) ? nblk - 1 - j : j;
}

}  // namespace

template <int DP>
__global__ __launch_bounds__(256) void bwd16_delta_kernel(BwdParams p) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= (int64_t)p.B * p.H * p.Sq) return;
    // DP / 64 columns per lane
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < DP / 64; ++j) {
        const int64_t at = row * DP + (DP / 64) * lane + j;
        s += load_as_float(p.dout, at, p.dout_prec) * (p.o_in_type ? load_as_float(p.o, at, p.in_prec) : p.o[at]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) p.dvec[row] = s * unit_dvec(p);
}

// ------------------------------------------------------------------------------------------------ dQ
// (batch, key/value head) slab that query head `bh` attends to: grouped-query attention without expanded K / V copies
__device__ __forceinline__ uint32_t bwd16_kv_slab(const BwdParams& p, uint32_t bh) {
    if (p.Hkv == 0 || p.Hkv == p.H) return bh;
    return (bh / p.H) * p.Hkv + (bh % p.H) / (p.H / p.Hkv);
}

template <typename T, bool CAUSAL, int DP>
__global__ __launch_bounds__(256, DP == 256 ? 1 : 2) void bwd16_dq_kernel(BwdParams p) {
    BWD16_GEO(DP);
    typedef Mma16<T> M;
    typedef typename M::V8 V8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [K buf0][K buf1][V buf0][V buf1], one 32-key tile (8 KiB at D = 128) each
    const int tid = threadIdx.x, lane = tid & 63, ql = lane & 31, hi = lane >> 5;
    const int wave = tid >> 6, uw = __builtin_amdgcn_readfirstlane(wave);
    const uint32_t nqb = (p.Sq + 127) / 128;
    const uint32_t vid = xcd_remap(blockIdx.x, nqb * p.B * p.H);
    uint32_t bh = vid / nqb;
    uint32_t qb = vid % nqb;
    if (CAUSAL) qb = nqb - 1 - causal_rank(vid, nqb, bh, DP != 256);  // the last query block sees the most keys
    const uint32_t q_row = qb * 128 + wave * 32 + ql, wave_q0 = qb * 128 + wave * 32;
    (void)wave_q0;  // (causal instantiations only)
    const bool qok = q_row < p.Sq;
    const T* qp = (const T*)p.q + (int64_t)bh * p.Sq * DP;
    const T* dop = (const T*)p.dout + (int64_t)bh * p.Sq * DP;
    const uint32_t kvbh = bwd16_kv_slab(p, bh);  // grouped K / V heads are read in place
    const T* kp = (const T*)p.k + (int64_t)kvbh * p.Skv * DP;
    const T* vp = (const T*)p.v + (int64_t)kvbh * p.Skv * DP;

    // B operands: lane (q, hi) holds Q[q][16 ks + 8 hi ..], dO[q][...]
    V8 qf[NKS], dof[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        if (qok) {
            qf[ks] = *(const V8*)(qp + (int64_t)q_row * DP + 16 * ks + 8 * hi);
            dof[ks] = *(const V8*)(dop + (int64_t)q_row * DP + 16 * ks + 8 * hi);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { qf[ks][j] = (T)0.0f; dof[ks][j] = (T)0.0f; }
        }
    }
    const float c = p.scale * UMFA_LOG2E * unit_c(p);  // (unit_c: operands that arrive as power-of-two multiples, BwdParams::units)
    const float L2 = qok ? p.lse[(int64_t)bh * p.Sq + q_row] * UMFA_LOG2E : INFINITY;  // +inf -> P = 0
    // D[q] = rowsum(dO o O), fused here (the reference zeroes and fills a D scratch in its own pass): this lane already
    // holds dO[q][16 ks + 8 hi .. +7] for every k-step -- half of the row -- so it reads the same half of O (fp32, or the
    // operand type on the in-stream entry), multiplies, and the two lanes of a row add their halves.  The dK / dV
    // kernel, launched behind this one, reads D from the scratch.
    float delta = 0.0f;
    if (qok) {
        const int64_t orow = ((int64_t)bh * p.Sq + q_row) * DP;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int64_t at = orow + 16 * ks + 8 * hi;
            if (p.o_in_type) {
                const V8 ov = *(const V8*)((const T*)p.o + at);
#pragma unroll
                for (int j = 0; j < 8; ++j) delta = __builtin_fmaf((float)dof[ks][j], (float)ov[j], delta);
            } else {
                const f32x4 o0 = *(const f32x4*)(p.o + at), o1 = *(const f32x4*)(p.o + at + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    delta = __builtin_fmaf((float)dof[ks][j], o0[j], delta);
                    delta = __builtin_fmaf((float)dof[ks][4 + j], o1[j], delta);
                }
            }
        }
    }
    delta += __shfl_xor(delta, 32, 64);
    const float delta_out = delta * unit_dvec(p);  // D in true units for the caller ...
    delta *= unit_rowd(p);                         // ... and in dP's units (dO and V arrive as power-of-two multiples) for this kernel and bwd16_dkdv
    if (qok && hi == 0) {
        const int64_t ri = (int64_t)bh * p.Sq + q_row;
        p.dvec[ri] = delta_out;
        // ... and, for bwd16_dkdv, the two row constants in the form it consumes them: the addend of the exponent FMA and
        // the initial value of the dP accumulator (dP - D comes out of the MFMA chain) -- 64 vector instructions per tile
        // (32 multiplies, 32 subtractions) that its loop no longer issues
        p.rowc[ri] = -L2;
        p.rowc[(int64_t)p.B * p.H * p.Sq + ri] = -delta;
    }

    const i32x4 k_srd = make_srd(kp, p.Skv * (uint32_t)ROW_B), v_srd = make_srd(vp, p.Skv * (uint32_t)ROW_B);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((LDS_AS char*)smem));
#pragma unroll
    for (int i = 0; i < 4 * TILE_BYTES / 4096; ++i) *(i32x4*)(smem + i * 4096 + tid * 16) = i32x4{0, 0, 0, 0};
    __syncthreads();

    uint32_t ntiles = (p.Skv + 31) / 32;
    if (CAUSAL) {
        const uint32_t lim = (qb * 128 + 128 + 31) / 32;
        ntiles = ntiles < lim ? ntiles : lim;
    }
    f32x16 acc[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    const int tr_qq = (lane >> 2) & 3, tr_pp = lane & 3, tr_g1 = (lane >> 4) & 1;

    auto stage = [&](uint32_t t) __attribute__((always_inline)) {
        dma_rows<TILE_PIECES, DP>(k_srd, lds0 + (t & 1) * TILE_BYTES, t * 32, uw, lane);
        dma_rows<TILE_PIECES, DP>(v_srd, lds0 + 2 * TILE_BYTES + (t & 1) * TILE_BYTES, t * 32, uw, lane);
    };
    stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ... and once more as a builtin, for hipcc's own scoreboard: Q / dO fragments (loaded above, first used inside the
    // loop) otherwise keep "s_waitcnt vmcnt(1) / vmcnt(0)" in front of the first MFMAs of EVERY tile -- right behind
    // stage(t + 1), whose LDS-DMA loads the compiler does not see, so each tile waited for its successor's prefetch
    // (tools/trace_waits.py; same mechanism as in fa_fwd16_w64)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();

    // One 32-key tile.  EDGE: the tile crosses the end of the key range or this wave's causal diagonal, scores get a
    // per-key test; every other tile runs without it (as a run-time flag inside ONE body the test became 16 v_cmp + 32
    // v_cndmask on every tile of every launch, in a loop that is VALU-issue bound: two waves per SIMD, ~170 vector
    // instructions per 24 MFMAs).
    // PAR: which half of the double buffer, as a compile-time constant in the steady-state loop (unrolled by two): every
    // LDS address is then a loop-invariant lane register + an immediate, where a run-time (t & 1) cost ~40 address adds
    // per tile; -1 = run-time parity (the few edge tiles).
    auto tile_body = [&](uint32_t t, auto EDGE_C, auto PAR_C) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(EDGE_C)::value;
        constexpr int PAR = decltype(PAR_C)::value;
        const int par = PAR >= 0 ? PAR : (int)(t & 1);
        const char* Kt = smem + par * TILE_BYTES;
        const char* Vt = smem + 2 * TILE_BYTES + par * TILE_BYTES;
        const uint32_t key_base = t * 32;
        f32x16 s, dp;
        // explicit software pipeline: the row fragments of k-step ks + PD are in flight while the MFMAs of ks
        // run (left alone, hipcc reuses ONE 4-register buffer: ds_read -> s_waitcnt lgkmcnt(0) -> MFMA, 16 times)
        V8 ak[NKS], av[NKS];
#pragma unroll
        for (int ks = 0; ks < PD; ++ks) {
            ak[ks] = *(const V8*)(Kt + d_off<DP>(ql, 2 * ks + hi));
            av[ks] = *(const V8*)(Vt + d_off<DP>(ql, 2 * ks + hi));
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * PD, 0);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if (ks + PD < NKS) {
                ak[ks + PD] = *(const V8*)(Kt + d_off<DP>(ql, 2 * (ks + PD) + hi));
                av[ks + PD] = *(const V8*)(Vt + d_off<DP>(ql, 2 * (ks + PD) + hi));
            }
            s = M::mma(ak[ks], qf[ks], ks ? s : f32x16{});      // S^T[key][q]   (first k-step: C = 0 inline)
            dp = M::mma(av[ks], dof[ks], ks ? dp : f32x16{});   // dP^T[key][q]
            if (ks + PD < NKS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
        V8 ds[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c, -L2));
            if constexpr (EDGE) {
                const uint32_t key = key_base + acc_row(r, hi);
                if (key >= p.Skv || (CAUSAL && key > q_row)) pr = 0.0f;
            }
            ds[r >> 3][r & 7] = (T)(pr * (dp[r] - delta));
        }
        // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
                acc[i] = M::mma(tr_frag<M, DP>(Kt, i, s2, hi, tr_qq, tr_pp, tr_g1), ds[s2], acc[i]);
    };
    // tile ranges of this wave (wave-uniform): [0, t1) plain, [t1, t2) edge, [t2, ntiles) nothing to do but keep the
    // staging and the barriers of the workgroup going
    const uint32_t wq0 = __builtin_amdgcn_readfirstlane(qb * 128 + (uint32_t)uw * 32);
    uint32_t t2 = ntiles, t1 = p.Skv / 32;
    if (CAUSAL) {
        t2 = wq0 / 32 + 1 < ntiles ? wq0 / 32 + 1 : ntiles;  // key_base <= wave_q0 + 31
        t1 = wq0 / 32 < t1 ? wq0 / 32 : t1;                  // key_base + 31 <= wave_q0
    }
    t1 = t1 < t2 ? t1 : t2;
    uint32_t t = 0;
    for (; t + 1 < t1; t += 2) {  // t even
        stage(t + 1);  // other buffer: its last readers passed the previous barrier
        tile_body(t, std::false_type{}, std::integral_constant<int, 0>{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        stage(t + 2);
        tile_body(t + 1, std::false_type{}, std::integral_constant<int, 1>{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    for (; t < t2; ++t) {  // an odd plain tile (computed with the test: same values) and the edge tiles
        stage(t + 1);
        tile_body(t, std::true_type{}, std::integral_constant<int, -1>{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    for (; t < ntiles; ++t) {
        stage(t + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const float osc = p.scale * unit_dq(p);
    if (qok) {
        const int64_t orow = ((int64_t)bh * p.Sq + q_row) * DP;
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 val = {acc[i][4 * g] * osc, acc[i][4 * g + 1] * osc, acc[i][4 * g + 2] * osc,
                             acc[i][4 * g + 3] * osc};
                store_grad4<T>(p.dq, orow + 32 * i + 8 * g + 4 * hi, val, p.grad_in_type != 0);
            }
    }
}

// ------------------------------------------------------------------------------------------------ dQ, head_dim 128, v2
// The same products as bwd16_dq with the structure bwd16_dkdv got in round 2: ONE workgroup per CU (one wave per SIMD, the
// whole register file: FLUX = 768 workgroups = 3 full rounds, where two-per-CU left every CU's third workgroup running
// alone at a third of the matrix pipe), 64-key tiles = two 32-key sub-tiles per barrier, and the tile as four phases of
// back-to-back MFMAs with the vector work of the next phase's operands between them (source order pinned):
//   P1a  S^T, dP^T of sub-tile 0 (K / V row fragments PF ahead)      P1b  S^T, dP^T of sub-tile 1 | P, dS of sub-tile 0
//   P2a  dQ^T += K0^T dS0 (transposed fragments PT ahead)            | P, dS of sub-tile 1          P2b  dQ^T += K1^T dS1
// S^T / dP^T are VGPR-destination MFMAs (Mma16::mma_v) with the Q / dO fragments homed in AGPRs.
template <typename T, bool CAUSAL>
__global__ __launch_bounds__(256, 1) void bwd16_dq2_kernel(BwdParams p) {
    constexpr int DP = 128;
    BWD16_GEO(DP);
    (void)PD;  // (this kernel keeps PF = 8 row fragments in flight instead)
    typedef Mma16<T> M;
    typedef typename M::V8 V8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [K buf0 16 KiB][K buf1][V buf0][V buf1]: 64-key tiles = two 32-row sub-images each
    constexpr int KROWS = 64, KTILE_B = KROWS * ROW_B, KT = 0, VT = 2 * KTILE_B;
    const int tid = threadIdx.x, lane = tid & 63, ql = lane & 31, hi = lane >> 5;
    const int wave = tid >> 6, uw = __builtin_amdgcn_readfirstlane(wave);
    const uint32_t nqb = (p.Sq + 127) / 128;
    const uint32_t vid = xcd_remap(blockIdx.x, nqb * p.B * p.H);
    uint32_t bh = vid / nqb;
    uint32_t qb = vid % nqb;
    if (CAUSAL) { bh = vid / nqb; qb = nqb - 1 - vid % nqb; }  // longest first (one workgroup per CU: slots refill as they free up)
    const uint32_t q_row = qb * 128 + wave * 32 + ql;
    const bool qok = q_row < p.Sq;
    const T* qp = (const T*)p.q + (int64_t)bh * p.Sq * DP;
    const T* dop = (const T*)p.dout + (int64_t)bh * p.Sq * DP;
    const uint32_t kvbh = bwd16_kv_slab(p, bh);  // grouped K / V heads are read in place
    const T* kp = (const T*)p.k + (int64_t)kvbh * p.Skv * DP;
    const T* vp = (const T*)p.v + (int64_t)kvbh * p.Skv * DP;
    const i32x4 k_srd = make_srd(kp, p.Skv * (uint32_t)ROW_B), v_srd = make_srd(vp, p.Skv * (uint32_t)ROW_B);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((LDS_AS char*)smem));
#pragma unroll
    for (int i = 0; i < 4 * KTILE_B / 4096; ++i) *(i32x4*)(smem + i * 4096 + tid * 16) = i32x4{0, 0, 0, 0};
    __syncthreads();
    int dma_off[(2 * TILE_PIECES + 3) / 4];
    dma_lane_offsets<2 * TILE_PIECES, DP>(dma_off, uw, lane);
    auto stage = [&](uint32_t t) __attribute__((always_inline)) {
        dma_rows_pre<2 * TILE_PIECES, DP>(k_srd, lds0 + KT + (t & 1) * KTILE_B, t * KROWS, uw, dma_off);
        dma_rows_pre<2 * TILE_PIECES, DP>(v_srd, lds0 + VT + (t & 1) * KTILE_B, t * KROWS, uw, dma_off);
    };
    stage(0);  // first: its latency runs under the Q / dO / O loads and the D row sums below

    // B operands: lane (q, hi) holds Q[q][16 ks + 8 hi ..], dO[q][...]
    V8 qf[NKS], dof[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        if (qok) {
            qf[ks] = *(const V8*)(qp + (int64_t)q_row * DP + 16 * ks + 8 * hi);
            dof[ks] = *(const V8*)(dop + (int64_t)q_row * DP + 16 * ks + 8 * hi);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { qf[ks][j] = (T)0.0f; dof[ks][j] = (T)0.0f; }
        }
    }
    const float c = p.scale * UMFA_LOG2E * unit_c(p);  // (unit_c: operands that arrive as power-of-two multiples, BwdParams::units)
    const float L2 = qok ? p.lse[(int64_t)bh * p.Sq + q_row] * UMFA_LOG2E : INFINITY;  // +inf -> P = 0
    // D[q] = rowsum(dO o O), fused here as in bwd16_dq (see there)
    float delta = 0.0f;
    if (qok) {
        const int64_t orow = ((int64_t)bh * p.Sq + q_row) * DP;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int64_t at = orow + 16 * ks + 8 * hi;
            if (p.o_in_type) {
                const V8 ov = *(const V8*)((const T*)p.o + at);
#pragma unroll
                for (int j = 0; j < 8; ++j) delta = __builtin_fmaf((float)dof[ks][j], (float)ov[j], delta);
            } else {
                const f32x4 o0 = *(const f32x4*)(p.o + at), o1 = *(const f32x4*)(p.o + at + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    delta = __builtin_fmaf((float)dof[ks][j], o0[j], delta);
                    delta = __builtin_fmaf((float)dof[ks][4 + j], o1[j], delta);
                }
            }
        }
    }
    delta += __shfl_xor(delta, 32, 64);
    const float delta_out = delta * unit_dvec(p);  // D in true units for the caller ...
    delta *= unit_rowd(p);                         // ... and in dP's units (dO and V arrive as power-of-two multiples) for this kernel and bwd16_dkdv
    if (qok && hi == 0) {
        const int64_t ri = (int64_t)bh * p.Sq + q_row;
        p.dvec[ri] = delta_out;
        p.rowc[ri] = -L2;                                        // row constants for bwd16_dkdv (see bwd16_dq)
        p.rowc[(int64_t)p.B * p.H * p.Sq + ri] = -delta;
    }
    // home the B operands in the accumulator half of the register file (Mma16::mma_v takes them from there)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+a"(qf[ks]), "+a"(dof[ks]));


    uint32_t ntiles = (p.Skv + KROWS - 1) / KROWS;
    if (CAUSAL) {
        const uint32_t lim = (qb * 128 + 128 + KROWS - 1) / KROWS;
        ntiles = ntiles < lim ? ntiles : lim;
    }
    f32x16 acc[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    const int tr_qq = (lane >> 2) & 3, tr_pp = lane & 3, tr_g1 = (lane >> 4) & 1;


    constexpr int NPW = (2 * TILE_PIECES + 3) / 4;
    auto stage_piece = [&](uint32_t t, int i) __attribute__((always_inline)) {  // one piece at a time from inside the pinned MFMA stream (see bwd16_dkdv)
        const int j = i % NPW, n = 4 * j + uw;
        const int voff = (int)(t * KROWS) * ROW_B + dma_off[j];
        dma_piece(i < NPW ? k_srd : v_srd, lds0 + (i < NPW ? KT : VT) + (t & 1) * KTILE_B + n * 1024, voff);
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tile 0 (issued at the top of the kernel) and everything loaded since
    __builtin_amdgcn_s_waitcnt(0x0F70);  // and for hipcc's scoreboard (Q / dO fragment loads; see bwd16_dq)
    __syncthreads();

    auto tile_body = [&](uint32_t t, auto EDGE_C, auto PAR_C) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(EDGE_C)::value;
        constexpr int PAR = decltype(PAR_C)::value;
        const int par = PAR >= 0 ? PAR : (int)(t & 1);
        const char* Kt = smem + KT + par * KTILE_B;
        const char* Vt = smem + VT + par * KTILE_B;
        const uint32_t key_base = t * KROWS;
        f32x16 s[2], dp[2];
        // row fragments as one flat sequence f: sub-tile u = f / 2NKS, NKS of K (S^T = K Q^T), then NKS of V (dP^T = V dO^T)
        // (alternating the two chains k-step by k-step measured slower, same box: the hardware chains back-to-back MFMAs
        // onto one accumulator without a bubble)
        V8 rf[4 * NKS];
        auto rd = [&](int f) {
            const int u = f / (2 * NKS), ks = f % NKS;
            rf[f] = *(const V8*)((((f / NKS) & 1) ? Vt : Kt) + u * TILE_BYTES + d_off<DP>(ql, 2 * ks + hi));
        };
        V8 ds[2][2];
        auto softmax_pair = [&](int u, int r0) {  // dS of scores r0, r0 + 1 of sub-tile u
#pragma unroll
            for (int r = r0; r < r0 + 2; ++r) {
                float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(s[u][r], c, -L2));
                if constexpr (EDGE) {
                    const uint32_t key = key_base + 32 * u + acc_row(r, hi);
                    if (key >= p.Skv || (CAUSAL && key > q_row)) pr = 0.0f;
                }
                ds[u][r >> 3][r & 7] = (T)(pr * (dp[u][r] - delta));
            }
        };
        constexpr int NF = 2 * NDB, PT = 4, PF = 8;  // dQ MFMAs of one sub-tile; transposed / row fragments in flight
        V8 tf[2 * NF];
        // dQ MFMA j: sub-tile u = j / NF; sub-tile 0 d-block-major, sub-tile 1 k-step-major (its first four MFMAs need only
        // the first half of dS(1): the second half of that softmax runs under them)
        auto trd = [&](int j) {
            const int u = j / NF, r = j % NF, i = u ? (r & 3) : (r >> 1), s2 = u ? (r >> 2) : (r & 1);
            tf[j] = tr_frag<M, DP>(Kt + u * TILE_BYTES, i, s2, hi, tr_qq, tr_pp, tr_g1);
        };
#pragma unroll
        for (int f = 0; f < PF; ++f) rd(f);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < 4 * NKS; ++f) {
            const int u = f / (2 * NKS), ks = f % NKS;
            if (f % 4 == 1 && f / 4 < 2 * NPW) stage_piece(t + 1, f / 4);
            if (f + PF < 4 * NKS) rd(f + PF);
            if (f >= 4 * NKS - PT) trd(f - (4 * NKS - PT));
            if ((f / NKS) & 1) { if (ks) M::mma_v(dp[u], rf[f], dof[ks]); else M::mma_v_first(dp[u], rf[f], dof[ks]); }
            else { if (ks) M::mma_v(s[u], rf[f], qf[ks]); else M::mma_v_first(s[u], rf[f], qf[ks]); }
            __builtin_amdgcn_sched_barrier(0);  // the vector work below must not move ahead of this MFMA (see Mma16::mma_v)
            // P1b: the 8 score pairs of sub-tile 0 behind every second MFMA (>= 2 issues after dP's last); ONE call site
            // per step: eight guarded copies per step put the loop over hipcc's pragma-unroll cost threshold, the loop
            // then stayed rolled, the accumulators went through scratch and were copied right behind the asm MFMAs
            if (f >= 2 * NKS + 1 && (f - (2 * NKS + 1)) % 2 == 0) softmax_pair(0, f - (2 * NKS + 1));
            __builtin_amdgcn_sched_barrier(0);
        }
        // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
#pragma unroll
        for (int j = 0; j < 2 * NF; ++j) {
            const int u = j / NF, r = j % NF, i = u ? (r & 3) : (r >> 1), s2 = u ? (r >> 2) : (r & 1);
            if (j + PT < 2 * NF) trd(j + PT);
            acc[i] = M::mma(tf[j], ds[u][s2], acc[i]);
            __builtin_amdgcn_sched_barrier(0);
            // the score pairs of sub-tile 1: 0..3 (dS(1)[0]) behind MFMAs 1, 3, 5, 7 of P2a (first: 2 issues after dP's last),
            // 4..7 (dS(1)[1]) behind MFMAs 8..11 = the k-step-0 MFMAs of P2b
            if (j < NF && (j & 1)) softmax_pair(1, j - 1);
            if (j >= NF && j < NF + 4) softmax_pair(1, 8 + 2 * (j - NF));
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto step = [&](uint32_t tt, auto EDGE_C, auto PAR_C) __attribute__((always_inline)) {
        tile_body(tt, EDGE_C, PAR_C);  // issues the next tile's LDS-DMA pieces itself (other buffer: its last readers passed the previous barrier)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    // tile ranges of this wave (wave-uniform): [0, t1) plain, [t1, t2) with the per-key test, [t2, ntiles) idle
    const uint32_t wq0 = __builtin_amdgcn_readfirstlane(qb * 128 + (uint32_t)uw * 32);
    uint32_t t2 = ntiles, t1 = p.Skv / KROWS;
    if (CAUSAL) {
        t2 = (wq0 + 31) / KROWS + 1 < ntiles ? (wq0 + 31) / KROWS + 1 : ntiles;  // key_base <= wave_q0 + 31
        t1 = (wq0 + 1) / KROWS < t1 ? (wq0 + 1) / KROWS : t1;                    // key_base + 63 <= wave_q0
    }
    t1 = t1 < t2 ? t1 : t2;
    uint32_t t = 0;
    for (; t + 1 < t1; t += 2) {
        step(t, std::false_type{}, std::integral_constant<int, 0>{});
        step(t + 1, std::false_type{}, std::integral_constant<int, 1>{});
    }
    for (; t < t2; ++t) step(t, std::true_type{}, std::integral_constant<int, -1>{});
    for (; t < ntiles; ++t) {
        stage(t + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const float osc = p.scale * unit_dq(p);
    if (p.grad_in_type) {  // operand-type dQ: through per-wave LDS rows, whole 256-byte rows out (see bwd16_dkdv's epilogue)
        char* stg = smem + wave * 8192;
        typedef T T4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = (4 * i + g) ^ (ql & 15);
                *(T4*)(stg + ql * 256 + 16 * ch + 8 * hi) = T4{(T)(acc[i][4 * g] * osc), (T)(acc[i][4 * g + 1] * osc),
                                                                (T)(acc[i][4 * g + 2] * osc), (T)(acc[i][4 * g + 3] * osc)};
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t wave_q0 = qb * 128 + wave * 32;
        T* out = (T*)p.dq + ((int64_t)bh * p.Sq + wave_q0) * DP;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int r = 4 * it + (lane >> 4), c = lane & 15;
            const i32x4 v16 = *(const i32x4*)(stg + r * 256 + 16 * (c ^ (r & 15)));
            if (wave_q0 + r < p.Sq) *(i32x4*)(out + r * DP + 8 * c) = v16;
        }
    } else if (qok) {
        const int64_t orow = ((int64_t)bh * p.Sq + q_row) * DP;
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 val = {acc[i][4 * g] * osc, acc[i][4 * g + 1] * osc, acc[i][4 * g + 2] * osc,
                             acc[i][4 * g + 3] * osc};
                store_grad4<T>(p.dq, orow + 32 * i + 8 * g + 4 * hi, val, false);
            }
    }
}

// ------------------------------------------------------------------------------------------------ dQ = scale dS K (dS-store form)
// The second half of bwd16_dq2 alone: dQ^T[d][q] += K^T[d][key] dS^T[key][q] with dS^T read back from the scratch bwd16_dkdv wrote
// (tiles [key tile 64][q-block 128][64][128], operand type): no S, no dP, no exponentials.  HBM-bound -- the dS matrix is read
// exactly once, 805 MB at the FLUX shape -- so what matters is bytes in flight: one workgroup per CU with a FOUR-slot LDS-DMA ring of
// (K tile, dS^T tile) pairs (3 x 16 KiB of dS per CU on their way = latency x the CU's share of the bandwidth), both operands as
// transposed-read fragments of the same dual-use image (a dS^T tile IS a [64][128] image with queries for columns).
// Non-causal, head_dim 128, Sq % 128 == 0, Skv % 64 == 0 (the launcher checks).
template <typename T>
__global__ __launch_bounds__(256, 1) void bwd16_dq_gemm_kernel(BwdParams p) {
    constexpr int DP = 128;
    BWD16_GEO(DP);
    (void)PD; (void)NKS;
    typedef Mma16<T> M;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KROWS = 64, KTILE_B = KROWS * ROW_B, NS = 4, SLOT_B = 2 * KTILE_B;  // slot = [K tile][dS^T tile]
    constexpr int NPW = (2 * TILE_PIECES + 3) / 4;                                  // LDS-DMA instructions per wave and image
    const int tid = threadIdx.x, lane = tid & 63, ql = lane & 31, hi = lane >> 5;
    const int wave = tid >> 6, uw = __builtin_amdgcn_readfirstlane(wave);
    const uint32_t nqb = p.Sq / 128;
    const uint32_t vid = xcd_remap(blockIdx.x, nqb * p.B * p.H);
    const uint32_t bh = vid / nqb, qb = vid % nqb;
    const uint32_t q_row = qb * 128 + wave * 32 + ql;
    const uint32_t kvbh = bwd16_kv_slab(p, bh);
    const T* kp = (const T*)p.k + (int64_t)kvbh * p.Skv * DP;
    const i32x4 k_srd = make_srd(kp, p.Skv * (uint32_t)ROW_B);
    // the slab as one image of 256-byte rows: tile (kt, qb) = rows (kt * nqb + qb) * 64 ... + 63
    const i32x4 ds_srd = make_srd((const char*)p.ds + (int64_t)bh * p.Sq * p.Skv * 2, p.Sq * p.Skv * 2u);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((LDS_AS char*)smem));
    int dma_off[NPW], ds_off[NPW];
    dma_lane_offsets<2 * TILE_PIECES, DP>(dma_off, uw, lane);
    // a dS^T tile in memory: 8-byte unit (query group gq of 32, key k of 64) at (gq * 64 + k) * 8 (bwd16_dkdv).  In LDS the same, with
    // the 16-byte chunk index XORed with (gq & 7) << 1: the 32 lanes of a transposed read touch 8 query groups x 4 keys, and unswizzled
    // every group would start on the same bank (512-byte rows).  The swizzle goes on the SOURCE chunk (LDS-DMA writes lanes linearly).
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int sig = 64 * (4 * i + uw) + lane;  // LDS chunk slot of this lane in piece 4 i + uw
        ds_off[i] = (sig ^ (((sig >> 5) & 7) << 1)) * 16;
    }
    const uint32_t ntiles = p.Skv / KROWS;
    auto stage = [&](uint32_t t) __attribute__((always_inline)) {  // (tiles past the end: range-checked away, same instruction count)
        dma_rows_pre<2 * TILE_PIECES, DP>(k_srd, lds0 + (t % NS) * SLOT_B, t * KROWS, uw, dma_off);
        dma_rows_pre<2 * TILE_PIECES, DP>(ds_srd, lds0 + (t % NS) * SLOT_B + KTILE_B, t < ntiles ? (t * nqb + qb) * KROWS : 0x7fffffu, uw, ds_off);
    };
#pragma unroll
    for (int i = 0; i < NS - 1; ++i) stage(i);
    f32x16 acc[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    const int tr_qq = (lane >> 2) & 3, tr_pp = lane & 3, tr_g1 = (lane >> 4) & 1;
    // this lane's unit of a B fragment: query group 8 wave + 4 g1 + pp, keys 16 (2 u + s2) + 4 hi + qq (and + 8): byte offsets in the tile
    int dsu[2][2][2];
    {
        const int gq = 8 * wave + 4 * tr_g1 + tr_pp;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int h8 = 0; h8 < 2; ++h8) {
                    const int k = 32 * u + 16 * s2 + 8 * h8 + 4 * hi + tr_qq;
                    dsu[u][s2][h8] = gq * 512 + ((k * 8) ^ ((gq & 7) << 5));
                }
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * 2 * NPW) : "memory");  // tile 0 has landed; the two younger ones stay in flight
    __syncthreads();
    for (uint32_t t = 0; t < ntiles; ++t) {
        const char* Kt = smem + (t % NS) * SLOT_B;
        const char* Dt = Kt + KTILE_B;
        stage(t + NS - 1);  // into the slot tile t - 1 was read from (every wave is past the barrier that ended it)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                // B operand: lane (q, hi) holds dS^T[16-key step][q] in the accumulator-row key order the transposed K fragments use
                const typename M::V4 blo = M::tr_read(Dt + dsu[u][s2][0]), bhi = M::tr_read(Dt + dsu[u][s2][1]);
                const typename M::V8 b = __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int i = 0; i < NDB; ++i) acc[i] = M::mma(tr_frag<M, DP>(Kt + u * TILE_BYTES, i, s2, hi, tr_qq, tr_pp, tr_g1), b, acc[i]);
            }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * 2 * NPW) : "memory");  // tile t + 1 has landed
        __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ring's last (out-of-range) requests: the epilogue below reuses the tile area
    __syncthreads();
    const float osc = p.scale * unit_dq(p);
    if (p.grad_in_type) {  // operand-type dQ through per-wave LDS rows (see bwd16_dq2)
        char* stg = smem + wave * 8192;
        typedef T T4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = (4 * i + g) ^ (ql & 15);
                *(T4*)(stg + ql * 256 + 16 * ch + 8 * hi) = T4{(T)(acc[i][4 * g] * osc), (T)(acc[i][4 * g + 1] * osc),
                                                                (T)(acc[i][4 * g + 2] * osc), (T)(acc[i][4 * g + 3] * osc)};
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t wave_q0 = qb * 128 + wave * 32;
        T* out = (T*)p.dq + ((int64_t)bh * p.Sq + wave_q0) * DP;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int r = 4 * it + (lane >> 4), c = lane & 15;
            const i32x4 v16 = *(const i32x4*)(stg + r * 256 + 16 * (c ^ (r & 15)));
            *(i32x4*)(out + r * DP + 8 * c) = v16;
        }
    } else {
        const int64_t orow = ((int64_t)bh * p.Sq + q_row) * DP;
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 val = {acc[i][4 * g] * osc, acc[i][4 * g + 1] * osc, acc[i][4 * g + 2] * osc,
                             acc[i][4 * g + 3] * osc};
                store_grad4<T>(p.dq, orow + 32 * i + 8 * g + 4 * hi, val, false);
            }
    }
}

// ------------------------------------------------------------------------------------------------ dK, dV
// STORE_DS (head_dim 128, non-causal): every dS = P (dP - D) the kernel forms is also written, in the operand type, to
// p.ds[bh][q][key] -- bwd16_dq_gemm then needs no S, dP or exponentials of its own (the dS-store form of the backward)
template <typename T, bool CAUSAL, int DP, bool STORE_DS = false>
__global__ __launch_bounds__(256, DP == 64 ? 2 : 1) void bwd16_dkdv_kernel(BwdParams p) {
    static_assert(!STORE_DS || (DP == 128 && !CAUSAL), "the dS-store form exists at head_dim 128, non-causal");
    BWD16_GEO(DP);
#ifdef BWD16_LAB_STAMP
    const unsigned long long rt_entry = __builtin_amdgcn_s_memrealtime();
    unsigned long long rt_loop0 = 0, rt_loop1 = 0;
#endif
    typedef Mma16<T> M;
    typedef typename M::V8 V8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [Q buf0 8 KiB][Q buf1][dO buf0][dO buf1][L2 2x32 f32][D 2x32 f32]; this wave's K / V fragments live in registers
    // (one wave per SIMD: 512-register budget, accumulators in AGPRs)
    constexpr int QT = 0, DOT = QT + 4 * TILE_BYTES, VEC = DOT + 4 * TILE_BYTES;  // 64-row tiles, double-buffered
    const int tid = threadIdx.x, lane = tid & 63, kl = lane & 31, hi = lane >> 5;
    const int wave = tid >> 6, uw = __builtin_amdgcn_readfirstlane(wave);
    const uint32_t nkb = (p.Skv + 127) / 128;
    // Non-causal launches may come as a PERSISTENT grid (one workgroup per CU looping over its key blocks, item += gridDim.x:
    // no dispatch latency and no LDS clearing between the ~3 blocks a CU gets at the FLUX shape); with gridDim.x = number
    // of items the loop runs once.  Causal launches keep one workgroup per item (unequal items: dynamic dispatch balances).
    const uint32_t n_items = nkb * p.B * p.H;
    for (uint32_t item = blockIdx.x; item < n_items; item += gridDim.x) {
    const uint32_t vid = xcd_remap(item, n_items);
    uint32_t bh = vid / nkb, kb = vid % nkb;
    if (CAUSAL) kb = causal_rank(vid, nkb, bh, DP == 64);  // the first key block is seen by the most queries
    const uint32_t key = kb * 128 + wave * 32 + kl, wave_k0 = kb * 128 + wave * 32;
    const bool kok = key < p.Skv;
    const T* qp = (const T*)p.q + (int64_t)bh * p.Sq * DP;
    const T* dop = (const T*)p.dout + (int64_t)bh * p.Sq * DP;
    const uint32_t kvbh = bwd16_kv_slab(p, bh);  // grouped K / V heads are read in place
    const T* kp = (const T*)p.k + (int64_t)kvbh * p.Skv * DP;
    const T* vp = (const T*)p.v + (int64_t)kvbh * p.Skv * DP;
    const i32x4 q_srd = make_srd(qp, p.Sq * (uint32_t)ROW_B), do_srd = make_srd(dop, p.Sq * (uint32_t)ROW_B);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((LDS_AS char*)smem));
    float* const vec = (float*)(smem + VEC);
    // dS-store form: this (batch, head)'s slab of the scratch holds dS^T as 16-KiB tiles [key tile of 64][q-block of 128], inside a tile
    // [query group of 4: 32][key: 64][4 queries] in the operand type.  A lane (key, hi) holds 4 consecutive queries per register group =
    // one 8-byte unit, and the 32 lanes of a half are 32 consecutive keys of ONE query group: every store instruction writes two
    // contiguous 256-byte runs (lanes index keys -- in a [key][query] layout each lane's 16 bytes were a memory request of their own,
    // 50 M requests per FLUX backward: +90 us on this kernel; in [query][key] they are 32 two-byte stores per tile, +111 us).
    // bwd16_dq_gemm reads the units back with transposed LDS reads (any arrangement of 8-byte units serves those).
    const uint32_t ds_nqb = p.Sq / 128u;
    const auto ds_rsrc = __builtin_amdgcn_make_buffer_rsrc(STORE_DS ? (void*)((char*)p.ds + (int64_t)bh * p.Sq * p.Skv * 2) : (void*)nullptr, 0,
                                                           STORE_DS ? (int)(p.Sq * p.Skv * 2u) : 0, 0x00020000);
    const int ds_voff = ((wave & 1) * 32 + kl) * 8 + hi * 512;         // key row of the 64-key tile; the odd query group of a register group
    const uint32_t ds_kt = 2u * kb + (uint32_t)(uw >> 1);              // this wave's key tile
    (void)ds_rsrc; (void)ds_voff; (void)ds_nqb; (void)ds_kt;
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    // sub-tile (32 queries from q0_sub) of this lane's key: register groups 0, 1 in a, 2, 3 in b (two dwords each)
    auto ds_store = [&](uint32_t q0_sub, const u32x4_t& a, const u32x4_t& b) __attribute__((always_inline)) {
        if constexpr (STORE_DS) {
            const uint32_t tile = (p.ds_lab & 1) ? 0u : ds_kt * ds_nqb + q0_sub / 128u;
            const int soff = (int)(tile * 16384u + ((q0_sub % 128u) / 4u) * 512u);  // query group 8 u' of the tile; register group g: + 2 g groups
#ifndef BWD16_DS_AUX
#define BWD16_DS_AUX 0
#endif
            __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{a[0], a[1]}, ds_rsrc, ds_voff, soff, BWD16_DS_AUX);
            __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{a[2], a[3]}, ds_rsrc, ds_voff, soff + 1024, BWD16_DS_AUX);
            __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{b[0], b[1]}, ds_rsrc, ds_voff, soff + 2048, BWD16_DS_AUX);
            __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{b[2], b[3]}, ds_rsrc, ds_voff, soff + 3072, BWD16_DS_AUX);
        }
    };
    // the second sub-tile's units are stored at the top of the NEXT tile (issued late in a tile they would still be on their way at its
    // end-of-tile vmcnt(0), which the LDS-DMA ring needs)
    u32x4_t ds_carry[2] = {u32x4_t{0, 0, 0, 0}, u32x4_t{0, 0, 0, 0}};
    uint32_t ds_carry_q0 = 0xffffffffu;

    if (item == blockIdx.x) {
#pragma unroll
        for (int i = 0; i < VEC / 4096; ++i) *(i32x4*)(smem + i * 4096 + tid * 16) = i32x4{0, 0, 0, 0};
        __syncthreads();
    }
    // B operands of S = Q K^T and dP = dO V^T: lane (key, hi) holds K[key][16 ks + 8 hi ..], V[key][...]
    V8 kf[NKS], vf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        if (kok) {
            kf[ks] = *(const V8*)(kp + (int64_t)key * DP + 16 * ks + 8 * hi);
            vf[ks] = *(const V8*)(vp + (int64_t)key * DP + 16 * ks + 8 * hi);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { kf[ks][j] = (T)0.0f; vf[ks][j] = (T)0.0f; }
        }
    }

    if constexpr (DP == 128) {
        // home these 64 registers in the accumulator half of the file: the S / dP MFMAs of the pinned pipeline below take
        // them as AGPR B operands (Mma16::mma_v); left in VGPRs, hipcc copies each one to a scratch AGPR before every use
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+a"(kf[ks]), "+a"(vf[ks]));
    }
    const float c = p.scale * UMFA_LOG2E * unit_c(p);  // (unit_c: operands that arrive as power-of-two multiples, BwdParams::units)
    // query tiles of 64 rows = two 32-row sub-tiles per barrier: both S/dP products are issued before the first
    // sub-tile's exp/convert work, so the wave has matrix work in flight while its VALU runs
    constexpr int QROWS = 64, QTILE_B = QROWS * ROW_B;
    const uint32_t ntiles = (p.Sq + QROWS - 1) / QROWS;
    const uint32_t t0 = CAUSAL ? (kb * 128) / QROWS : 0;  // query tiles entirely before this key block see nothing
    constexpr int NPW = DP == 128 ? (2 * TILE_PIECES + 3) / 4 : 1;  // pieces per wave and tile image (hoisted offsets: head_dim 128 only)
    int dma_off[NPW];
    if constexpr (DP == 128) dma_lane_offsets<2 * TILE_PIECES, DP>(dma_off, uw, lane);
    auto stage = [&](uint32_t t) __attribute__((always_inline)) {
        if constexpr (DP == 128) {
            dma_rows_pre<2 * TILE_PIECES, DP>(q_srd, lds0 + QT + (t & 1) * QTILE_B, t * QROWS, uw, dma_off);
            dma_rows_pre<2 * TILE_PIECES, DP>(do_srd, lds0 + DOT + (t & 1) * QTILE_B, t * QROWS, uw, dma_off);
        } else {
            dma_rows<2 * TILE_PIECES, DP>(q_srd, lds0 + QT + (t & 1) * QTILE_B, t * QROWS, uw, lane);
            dma_rows<2 * TILE_PIECES, DP>(do_srd, lds0 + DOT + (t & 1) * QTILE_B, t * QROWS, uw, lane);
        }
    };
    // head_dim 128: the next tile's eight pieces one at a time from inside the pinned MFMA stream (a wave that issues them
    // back to back sits ~520 cycles in the issue of these nine instructions: phase stamps, tools/lab/bwd_stamps.py --
    // the same finding as in fa_fwd16_w64, where one LDS-DMA per four MFMA gaps was +6.5 %)
    auto stage_piece = [&](uint32_t t, int i) __attribute__((always_inline)) {  // i = 0 .. 2 NPW - 1: Q pieces, then dO pieces
        const int j = i % NPW, n = 4 * j + uw;
        const int voff = (int)(t * QROWS) * ROW_B + dma_off[j];
        dma_piece(i < NPW ? q_srd : do_srd, lds0 + (i < NPW ? QT : DOT) + (t & 1) * QTILE_B + n * 1024, voff);
    };
    // Row constants of a tile (LSE and D of its 64 rows) ride with the tile: one LDS-DMA dword load each (wave 0: LSE,
    // wave 1: D), waited for by the tile's own "s_waitcnt vmcnt(0)" + barrier.  As compiler-visible loads hipcc put
    // "s_waitcnt vmcnt(0)" in front of their ds_write -- right behind the LDS-DMA issue, which it cannot see -- so wave 0
    // waited for the whole prefetch at the top of every tile and the other waves for wave 0 at the barrier.
    // Rows past Sq read 0 (descriptor range check): their Q and dO rows are zero too, so S = dP = 0, P = 1, dS = 0 and
    // nothing reaches dK or dV.  LSE arrives in natural-log units; the factor log2(e) is applied where it is used.
    const i32x4 lse_srd = make_srd(p.rowc + (int64_t)bh * p.Sq, p.Sq * 4u), dv_srd = make_srd(p.rowc + ((int64_t)p.B * p.H + bh) * p.Sq, p.Sq * 4u);
    auto stage_consts = [&](uint32_t t) __attribute__((always_inline)) {
        const int voff = (int)(t * QROWS + (uint32_t)lane) * 4;
        if (uw == 0)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds"
                         ::"s"(lds0 + VEC + (t & 1) * QROWS * 4), "v"(voff), "s"(lse_srd) : "memory");
        else if (uw == 1)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds"
                         ::"s"(lds0 + VEC + 2 * QROWS * 4 + (t & 1) * QROWS * 4), "v"(voff), "s"(dv_srd) : "memory");
    };
    const int tr_qq = (lane >> 2) & 3, tr_pp = lane & 3, tr_g1 = (lane >> 4) & 1;
    // head_dim 256: dK and dV of all 8 d-blocks (256 accumulator registers) do not fit beside the K / V fragments (128),
    // so the query range is swept NH = 2 times, each pass owning 4 d-blocks (S and dP are recomputed: 6 products
    // issued for 4 -- against the fp32-exact path this head_dim had to take before, still > 10x)
#pragma unroll 1
    for (int hpass = 0; hpass < NH; ++hpass) {
    f32x16 dk[NDBH], dv[NDBH];
#pragma unroll
    for (int i = 0; i < NDBH; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[i][r] = 0.0f; dv[i][r] = 0.0f; }

    if (t0 < ntiles) { stage(t0); stage_consts(t0); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0F70);  // for the compiler's scoreboard too (K / V fragments are first used in the loop)
    __syncthreads();

    // Causal: the tiles whose queries all lie before this wave's keys (at most one per wave: wave_k0 - 63 <= q_base from
    // tile wave_k0 / 64 on) only keep the staging and the barriers going, in a loop of their own -- as a branch around
    // the tile body inside ONE loop, hipcc merged the 128 dK / dV accumulators of the two paths with ~330
    // v_accvgpr_read / _write copies per tile.
    uint32_t t = t0;
    if (CAUSAL) {
        const uint32_t t_act = __builtin_amdgcn_readfirstlane((kb * 128 + (uint32_t)uw * 32) / QROWS);  // wave-uniform for the compiler too
        for (; t < ntiles && t < t_act; ++t) {
            stage(t + 1);
            stage_consts(t + 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
#ifdef BWD16_LAB_STAMP  // lab: wait-free clock stamps at the phase boundaries of the pinned tile, summed per workgroup
    unsigned long long lst[7] = {0, 0, 0, 0, 0, 0, 0};
    uint32_t lacc[6] = {0, 0, 0, 0, 0, 0};
#define BWD_STAMP(k) asm volatile("s_memtime %0" : "=s"(lst[k]))
#else
#define BWD_STAMP(k) do { } while (0)
#endif
    // EDGE: some query of the tile lies before some key of this wave (causal diagonal): scores get a per-key test.
    // PAR: half of the double buffer as a compile-time constant (steady loop, unrolled by two: LDS addresses are then lane
    // registers + immediates; a run-time (t & 1) cost ~46 address adds per tile), -1 = run-time parity.
    auto tile_body = [&](uint32_t t, auto EDGE_C, auto PAR_C) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(EDGE_C)::value;
        constexpr int PAR = decltype(PAR_C)::value;
        const int par = PAR >= 0 ? PAR : (int)(t & 1);
        const char* Qt = smem + QT + par * QTILE_B;
        const char* dOt = smem + DOT + par * QTILE_B;
        const float* L2v = vec + par * QROWS;
        const float* Dv = vec + 2 * QROWS + par * QROWS;
        const uint32_t q_base = t * QROWS;
        {
            if constexpr (DP == 128) {
                // One tile = 64 MFMAs in four phases of 16 (at D = 128), every phase's MFMAs back to back and the vector work
                // of the NEXT phase's operands placed between them, in source order pinned by sched_barrier(0):
                //   P1a  S, dP of sub-tile 0                    (row fragments of Q, dO: PD k-steps in flight)
                //   P1b  S, dP of sub-tile 1        | P, dS of sub-tile 0 (exp2, products, rounding to T)
                //   P2a  dV, dK += sub-tile 0       | P, dS of sub-tile 1
                //   P2b  dV, dK += sub-tile 1
                // Left to hipcc the second half was "2 x ds_read_b64_tr_b16 -> s_waitcnt lgkmcnt(0) -> MFMA" 32 times per
                // tile with the exponentials in front of it, nothing overlapped (tools/trace_waits.py; PMC: 38 % of the
                // wave cycles in waits, MFMA pipe 36 % busy).
                // S[q][key] = Q K^T, dP[q][key] = dO V^T: rows = queries (registers), columns = keys (lanes)
                f32x16 s[2], dp[2];
                // row fragments as ONE flat sequence f: sub-tile u = f / 2NKS, then the NKS fragments of Q (S = Q K^T), then the
                // NKS of dO (dP = dO V^T) -- all S MFMAs of a sub-tile before its dP MFMAs, so the dP accumulator can start
                // at -D (row constants read from LDS while the S MFMAs run): dP - D comes out of the MFMA chain
                V8 rf[4 * NKS];
                auto rd = [&](int f) {
                    const int u = f / (2 * NKS), ks = f % NKS;
                    rf[f] = *(const V8*)((((f / NKS) & 1) ? dOt : Qt) + u * TILE_BYTES + d_off<DP>(kl, 2 * ks + hi));
                };
                // row constants of sub-tile u: registers 4g .. 4g+3 are queries 8g + 4hi + 0..3
                f32x4 l2c[2][4];
                auto rd_consts = [&](int u) {
    #pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        l2c[u][g] = *(const f32x4*)(L2v + 32 * u + 8 * g + 4 * hi);   // -LSE log2(e)
                        const f32x4 nd = *(const f32x4*)(Dv + 32 * u + 8 * g + 4 * hi);  // -D
    #pragma unroll
                        for (int e = 0; e < 4; ++e) dp[u][4 * g + e] = nd[e];
                    }
                };
                V8 pb[2][2], sb[2][2];
                // P and dS of scores r0, r0 + 1 of sub-tile u (one packed conversion each)
                auto softmax_pair = [&](int u, int r0) {
                    const uint32_t qb0 = q_base + 32 * u;
                    const bool edge = EDGE && CAUSAL && qb0 < wave_k0 + 31;  // sub-tile straddles the diagonal of this wave's keys
    #pragma unroll
                    for (int r = r0; r < r0 + 2; ++r) {
                        const int g = r >> 2, e = r & 3;
                        float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(s[u][r], c, l2c[u][g][e]));  // l2c = -LSE log2(e)
                        if (edge && key > qb0 + 8 * g + 4 * hi + e) pr = 0.0f;
                        pb[u][r >> 3][r & 7] = (T)pr;
                        const T dsv = (T)(pr * dp[u][r]);                                                // dp = dP - D (accumulator started at -D)
                        sb[u][r >> 3][r & 7] = dsv;
                    }
                };
                constexpr int NF = 4 * NDBH, PT = 4;  // dV / dK MFMAs of one sub-tile; transposed fragments in flight
                V8 tf[2 * NF];
                auto trd = [&](int j) {
                    const int u = j / NF, r = j % NF, i = r >> 2, s2 = (r >> 1) & 1;
                    tf[j] = tr_frag<M, DP>(((r & 1) ? Qt : dOt) + u * TILE_BYTES, i + hpass * NDBH, s2, hi, tr_qq, tr_pp, tr_g1);
                };
                constexpr int PF = 2 * PD;  // row fragments in flight
                BWD_STAMP(1);
                if constexpr (STORE_DS) {
                    if (ds_carry_q0 != 0xffffffffu) ds_store(ds_carry_q0, ds_carry[0], ds_carry[1]);
                }
    #pragma unroll
                for (int f = 0; f < PF; ++f) rd(f);
                __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                for (int f = 0; f < 4 * NKS; ++f) {
                    const int u = f / (2 * NKS), ks = f % NKS;
                    if (f == 2 * NKS) BWD_STAMP(2);
                    if (f == 1) rd_consts(0);  // behind the tile's first MFMA, not in front of it; needed at f = NKS
                    if (f % 4 == 1 && f / 4 < 2 * NPW) stage_piece(t + 1, f / 4);
                    if (f == 3) stage_consts(t + 1);
                    if (f + PF < 4 * NKS) rd(f + PF);
                    if (f == 2 * NKS + 2) rd_consts(1);  // needed from f = 3 NKS on (dP of sub-tile 1 starts at -D)
                    if (f >= 4 * NKS - PT) trd(f - (4 * NKS - PT));
                    // VGPR-destination MFMAs (Mma16::mma_v: the vector unit reads S and dP; as AGPR accumulators they cost
                    // 128 v_accvgpr_read per tile); first k-step of S: C = 0 as the inline constant
                    if ((f / NKS) & 1) M::mma_v(dp[u], rf[f], vf[ks]);
                    else if (ks) M::mma_v(s[u], rf[f], kf[ks]);
                    else M::mma_v_first(s[u], rf[f], kf[ks]);
                    __builtin_amdgcn_sched_barrier(0);  // the vector work below must not move ahead of this MFMA (see Mma16::mma_v)
                    // P1b: the 8 score pairs of sub-tile 0 behind every second MFMA (>= 2 issues after dP's last); one call site
                    // per step (see bwd16_dq2: guarded copies per step can push the loop over the pragma-unroll threshold)
                    if (f >= 2 * NKS + 1 && (f - (2 * NKS + 1)) % 2 == 0) softmax_pair(0, f - (2 * NKS + 1));
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (STORE_DS) {  // sub-tile 0 is complete (its last pair ran behind the P1b MFMAs)
                    ds_store(q_base, __builtin_bit_cast(u32x4_t, sb[0][0]), __builtin_bit_cast(u32x4_t, sb[0][1]));
                    __builtin_amdgcn_sched_barrier(0);
                }
    #pragma unroll
                for (int j = 0; j < 2 * NF; ++j) {
                    const int u = j / NF, r = j % NF, i = r >> 2, s2 = (r >> 1) & 1;
                    if (j == 0) BWD_STAMP(3);
                    if (j == NF) BWD_STAMP(4);
                    if (j + PT < 2 * NF) trd(j + PT);
                    if (r & 1) dk[i] = M::mma(tf[j], sb[u][s2], dk[i]);
                    else dv[i] = M::mma(tf[j], pb[u][s2], dv[i]);
                    __builtin_amdgcn_sched_barrier(0);
                    // P2a: the 8 score pairs of sub-tile 1 behind every second MFMA, from the second one on
                    if (j < NF && j % 2 == 1) softmax_pair(1, j - 1);
                    if constexpr (STORE_DS) {
                        if (j == NF + 1) {  // sub-tile 1 complete: stored at the top of the next tile
                            ds_carry[0] = __builtin_bit_cast(u32x4_t, sb[1][0]);
                            ds_carry[1] = __builtin_bit_cast(u32x4_t, sb[1][1]);
                            ds_carry_q0 = q_base + 32u;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {  // head_dim 64 (two workgroups per CU, 256 registers) and 256 (two passes): hipcc's own order
                // S[q][key] = Q K^T, dP[q][key] = dO V^T: rows = queries (registers), columns = keys (lanes)
                f32x16 s[2], dp[2];
    #pragma unroll
                for (int u = 0; u < 2; ++u)
    #pragma unroll
                    for (int r = 0; r < 16; ++r) { s[u][r] = 0.0f; dp[u][r] = 0.0f; }
                // A operands: row = query kl of sub-tile u, one flat sequence j = u * NKS + ks over both sub-tiles,
                // software-pipelined like the dq kernel's row fragments (PD k-steps in flight ahead of their MFMAs)
                V8 aq[2 * NKS], ado[2 * NKS];
                auto rd = [&](int j) {
                    aq[j] = *(const V8*)(Qt + (j / NKS) * TILE_BYTES + d_off<DP>(kl, 2 * (j % NKS) + hi));
                    ado[j] = *(const V8*)(dOt + (j / NKS) * TILE_BYTES + d_off<DP>(kl, 2 * (j % NKS) + hi));
                };
    #pragma unroll
                for (int j = 0; j < PD; ++j) rd(j);
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * PD, 0);
    #pragma unroll
                for (int j = 0; j < 2 * NKS; ++j) {
                    if (j + PD < 2 * NKS) rd(j + PD);
                    s[j / NKS] = M::mma(aq[j], kf[j % NKS], s[j / NKS]);
                    dp[j / NKS] = M::mma(ado[j], vf[j % NKS], dp[j / NKS]);
                    if (j + PD < 2 * NKS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                }
    #pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const uint32_t qb0 = q_base + 32 * u;
                    const bool edge = EDGE && CAUSAL && qb0 < wave_k0 + 31;  // sub-tile straddles the diagonal of this wave's keys
                    V8 pb[2], sb[2];
    #pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        // registers 4g .. 4g+3 are queries 8g + 4hi + 0..3 of the sub-tile
                        const f32x4 l2 = *(const f32x4*)(L2v + 32 * u + 8 * g + 4 * hi);
                        const f32x4 dl = *(const f32x4*)(Dv + 32 * u + 8 * g + 4 * hi);
    #pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int r = 4 * g + e;
                            float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(s[u][r], c, l2[e]));  // l2 = -LSE log2(e), dl = -D (p.rowc)
                            if (edge && key > qb0 + 8 * g + 4 * hi + e) pr = 0.0f;
                            pb[r >> 3][r & 7] = (T)pr;
                            sb[r >> 3][r & 7] = (T)(pr * (dp[u][r] + dl[e]));
                        }
                    }
                    // dV^T[d][key] += dO^T[d][q] P[q][key];  dK^T[d][key] += Q^T[d][q] dS[q][key]
    #pragma unroll
                    for (int i = 0; i < NDBH; ++i)
    #pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2) {
                            dv[i] = M::mma(tr_frag<M, DP>(dOt + u * TILE_BYTES, i + hpass * NDBH, s2, hi, tr_qq, tr_pp, tr_g1), pb[s2], dv[i]);
                            dk[i] = M::mma(tr_frag<M, DP>(Qt + u * TILE_BYTES, i + hpass * NDBH, s2, hi, tr_qq, tr_pp, tr_g1), sb[s2], dk[i]);
                        }
                }
            }
        }
    };
    auto step = [&](uint32_t tt, auto EDGE_C, auto PAR_C) __attribute__((always_inline)) {
        BWD_STAMP(0);
        if constexpr (DP != 128) {  // (head_dim 128 issues them piece by piece inside the tile body)
            stage(tt + 1);
            stage_consts(tt + 1);
        }
        tile_body(tt, EDGE_C, PAR_C);
        BWD_STAMP(5);
#ifdef BWD16_LAB_NOBAR      // lab, timing only (results are wrong): what the end-of-tile wait + barrier cost
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#endif
#ifdef BWD16_LAB_STAMP
        BWD_STAMP(6);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k_ = 0; k_ < 6; ++k_) lacc[k_] += (uint32_t)lst[k_ + 1] - (uint32_t)lst[k_];
#endif
    };
#ifdef BWD16_LAB_STAMP
    rt_loop0 = __builtin_amdgcn_s_memrealtime();
#endif
    // [t, t_ne): tiles on the diagonal (and one more when that leaves an odd start), run-time parity; then pairs
    uint32_t t_ne = t;
    if (CAUSAL) {
        t_ne = __builtin_amdgcn_readfirstlane((kb * 128 + (uint32_t)uw * 32 + 31 + QROWS - 1) / QROWS);  // q_base >= wave_k0 + 31
        t_ne = t_ne > t ? t_ne : t;
    }
    t_ne += t_ne & 1;
    t_ne = t_ne < ntiles ? t_ne : ntiles;
    for (; t < t_ne; ++t) step(t, std::true_type{}, std::integral_constant<int, -1>{});
    for (; t + 1 < ntiles; t += 2) {
        step(t, std::false_type{}, std::integral_constant<int, 0>{});
        step(t + 1, std::false_type{}, std::integral_constant<int, 1>{});
    }
    for (; t < ntiles; ++t) step(t, std::true_type{}, std::integral_constant<int, -1>{});
    if constexpr (STORE_DS) {
        if (ds_carry_q0 != 0xffffffffu) ds_store(ds_carry_q0, ds_carry[0], ds_carry[1]);
        ds_carry_q0 = 0xffffffffu;
    }
#ifdef BWD16_LAB_STAMP
    rt_loop1 = __builtin_amdgcn_s_memrealtime();
#endif
    const float gsv = unit_dv(p), osc = p.scale * unit_dk(p);
    if (DP == 128 && p.grad_in_type && !p.dkdv_fp32) {
        // Gradients in the operand type (in-stream entry), head_dim 128: a lane holds 4 consecutive d of ONE key per register
        // group, so direct stores touch 32 rows x 8 bytes per instruction (64 scattered store instructions per wave and
        // tensor pair, ~5.8 us per workgroup); instead each wave writes its 32 x 128 block into its own 8 KiB of the (idle)
        // tile buffers -- 16-byte chunks XOR-swizzled with the row -- and streams it out as whole 256-byte rows, 16 bytes
        // per lane, 4 rows per instruction (the forward's T21).  dK first, then dV through the same 8 KiB.
        char* stg = smem + wave * 8192;
        typedef T T4 __attribute__((ext_vector_type(4)));
        auto flush = [&](void* base, auto&& val) {
#pragma unroll
            for (int i = 0; i < NDBH; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 x = val(i, g);
                    const int ch = (4 * i + g) ^ (kl & 15);  // 16-byte chunk holding d = 32 i + 8 g + {0..7}
                    *(T4*)(stg + kl * 256 + 16 * ch + 8 * hi) = T4{(T)x[0], (T)x[1], (T)x[2], (T)x[3]};
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            T* out = (T*)base + ((int64_t)bh * p.Skv + wave_k0) * DP;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int r = 4 * it + (lane >> 4), c = lane & 15;
                const i32x4 v16 = *(const i32x4*)(stg + r * 256 + 16 * (c ^ (r & 15)));
                if (wave_k0 + r < p.Skv) *(i32x4*)(out + r * DP + 8 * c) = v16;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        };
        flush(p.dk, [&](int i, int g) { return f32x4{dk[i][4 * g] * osc, dk[i][4 * g + 1] * osc, dk[i][4 * g + 2] * osc, dk[i][4 * g + 3] * osc}; });
        flush(p.dv, [&](int i, int g) { return f32x4{dv[i][4 * g] * gsv, dv[i][4 * g + 1] * gsv, dv[i][4 * g + 2] * gsv, dv[i][4 * g + 3] * gsv}; });
    } else if (kok) {
        const int64_t krow = ((int64_t)bh * p.Skv + key) * DP;
#pragma unroll
        for (int i = 0; i < NDBH; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = 32 * (i + hpass * NDBH) + 8 * g + 4 * hi;
                f32x4 kv = {dk[i][4 * g] * osc, dk[i][4 * g + 1] * osc, dk[i][4 * g + 2] * osc, dk[i][4 * g + 3] * osc};
                f32x4 vv = {dv[i][4 * g] * gsv, dv[i][4 * g + 1] * gsv, dv[i][4 * g + 2] * gsv, dv[i][4 * g + 3] * gsv};
                store_grad4<T>(p.dk, krow + d0, kv, p.grad_in_type != 0 && !p.dkdv_fp32);
                store_grad4<T>(p.dv, krow + d0, vv, p.grad_in_type != 0 && !p.dkdv_fp32);
            }
    }
#ifdef BWD16_LAB_STAMP
    if (tid == 0) {  // lab only: overwrites the head of the LSE input (dkdv reads p.rowc, not p.lse)
        uint32_t* dbg = (uint32_t*)p.lse + (size_t)blockIdx.x * 16;
        for (int k_ = 0; k_ < 6; ++k_) dbg[k_] = lacc[k_];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long rt_exit = __builtin_amdgcn_s_memrealtime();
        dbg[6] = (uint32_t)(rt_loop0 - rt_entry); dbg[7] = (uint32_t)(rt_loop1 - rt_loop0); dbg[8] = (uint32_t)(rt_exit - rt_loop1);
        dbg[9] = (uint32_t)rt_entry; dbg[10] = (uint32_t)rt_exit;
    }
#endif
    }  // hpass
    if (item + gridDim.x < n_items) __syncthreads();  // persistent grid: the epilogue staged through the tile buffers the next item's DMA writes
    }  // item
}

bool bwd_16_supported(const BwdParams& p) {
    if (p.in_prec != P_FP16 && p.in_prec != P_BF16) return false;
    if (p.dout_prec != p.in_prec || (p.D != 256 && p.D != 128 && p.D != 64) || p.mask) return false;
    if (p.Hkv && (p.Hkv > p.H || p.H % p.Hkv)) return false;
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    if (!al16(p.q) || !al16(p.k) || !al16(p.v) || !al16(p.dout) || !al16(p.dq) || !al16(p.dk) || !al16(p.dv)) return false;
    // 32-bit buffer offsets inside one (batch, head) slab
    return (uint64_t)p.Sq * 2 * p.D < (1ull << 31) && (uint64_t)p.Skv * 2 * p.D < (1ull << 31);
}

template <typename T, bool CAUSAL, int DP>
static hipError_t launch_bwd16_t(const BwdParams& p, hipStream_t stream) {
    constexpr int TILE_BYTES = 32 * 2 * DP;
    const int64_t rows = (int64_t)p.B * p.H * p.Sq;
    (void)rows;  // D = rowsum(dO o O) is computed inside bwd16_dq (bwd16_delta_kernel stays for reference / lab use)
    if (tuning().bwd_separate_delta.load(std::memory_order_relaxed)) hipLaunchKernelGGL(bwd16_delta_kernel<DP>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, p);
    const size_t lds_dq = 4 * TILE_BYTES, lds_kv = 8 * TILE_BYTES + 1024;
    if (hipError_t e = ensure_dynamic_lds((const void*)bwd16_dkdv_kernel<T, CAUSAL, DP>, lds_kv); e != hipSuccess) return e;
    if (hipError_t e = ensure_dynamic_lds((const void*)bwd16_dq_kernel<T, CAUSAL, DP>, lds_dq); e != hipSuccess) return e;
    const uint32_t nqb = (p.Sq + 127) / 128, nkb = (p.Skv + 127) / 128;
    // phases (the pre-quantised backward ABI computes dQ (+ D) and dK / dV in separate calls): bit 0 | bit 1 = the dQ kernel
    // (it also writes D and the row constants), bit 2 = dK / dV -- alone, the row constants are rebuilt from the caller's
    // (LSE, D) first
    const int ph = p.phases ? p.phases : 7;
    if constexpr (DP == 128 && !CAUSAL) {
        if (p.ds && ph == 7) {
            BwdParams pl = p;
            pl.ds_lab = tuning().bwd_ds_lab.load(std::memory_order_relaxed);  // (read once from UMFA_LAB_DS at start-up, or umfa_set_option: no getenv on a launch path)
            const BwdParams& p = pl;
            // dS-store form: D, row constants, dK / dV (+ dS to the scratch), dQ = scale dS K
            hipLaunchKernelGGL(bwd16_delta_kernel<DP>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, p);
            if (hipError_t e = launch_bwd16_rowc(p.lse, p.dvec, p.rowc, rows, p.units ? p.units + 6 : nullptr, stream); e != hipSuccess) return e;
            if (hipError_t e = ensure_dynamic_lds((const void*)bwd16_dkdv_kernel<T, false, DP, true>, lds_kv); e != hipSuccess) return e;
            hipLaunchKernelGGL((bwd16_dkdv_kernel<T, false, DP, true>), dim3(nkb * p.B * p.H), dim3(256), lds_kv, stream, p);
            const size_t lds_g = 4 * 2 * 64 * 2 * DP;  // four slots of (K tile, dS^T tile)
            if (hipError_t e = ensure_dynamic_lds((const void*)bwd16_dq_gemm_kernel<T>, lds_g); e != hipSuccess) return e;
            hipLaunchKernelGGL((bwd16_dq_gemm_kernel<T>), dim3(nqb * p.B * p.H), dim3(256), lds_g, stream, p);
            return hipGetLastError();
        }
    }
    if (!(ph & 3)) {
        if (hipError_t e = launch_bwd16_rowc(p.lse, p.dvec, p.rowc, rows, p.units ? p.units + 6 : nullptr, stream); e != hipSuccess) return e;
    } else
    if constexpr (DP == 128) {
        // one-workgroup-per-CU kernel for non-causal launches (same box, ms per backward, v2 / two-per-CU: FLUX 0.678 / 0.693,
        // B2 H8 S2048 0.145 / 0.159, B8 H16 S1024 0.304 / 0.299); causal launches keep the two-per-CU kernel (FLUX causal
        // 0.477 / 0.464, B4 H16 S8192 causal 3.46 / 3.45: unequal items balance better over two slots per CU).
        // UMFA_BWD_DQ=1 / 2 forces one of them (A/B).
        const int force = tuning().bwd_dq.load(std::memory_order_relaxed);  // umfa_set_option("bwd_dq", ...): tests drive both kernels in one process
        if (force == 2 || (force == 0 && !CAUSAL)) {
            const size_t lds_dq2 = 4 * 64 * 2 * DP;
            if (hipError_t e2 = ensure_dynamic_lds((const void*)bwd16_dq2_kernel<T, CAUSAL>, lds_dq2); e2 != hipSuccess) return e2;
            hipLaunchKernelGGL((bwd16_dq2_kernel<T, CAUSAL>), dim3(nqb * p.B * p.H), dim3(256), lds_dq2, stream, p);
        } else {
            hipLaunchKernelGGL((bwd16_dq_kernel<T, CAUSAL, DP>), dim3(nqb * p.B * p.H), dim3(256), lds_dq, stream, p);
        }
    } else {
        hipLaunchKernelGGL((bwd16_dq_kernel<T, CAUSAL, DP>), dim3(nqb * p.B * p.H), dim3(256), lds_dq, stream, p);
    }
    if (!(ph & 4)) return hipGetLastError();
    uint32_t kv_grid = nkb * p.B * p.H;
    // measured equal to one workgroup per item (FLUX 0.654-0.657 vs 0.656-0.680 ms, B1 H16 S8192 1.625 vs 1.631): off unless asked for
    if (!CAUSAL && DP == 128 && tuning().bwd_persist.load(std::memory_order_relaxed)) {  // persistent: one workgroup per CU (see the kernel)
        const uint32_t n_cu = (uint32_t)device_cu_count();
        if (kv_grid > n_cu) kv_grid = n_cu;
    }
    hipLaunchKernelGGL((bwd16_dkdv_kernel<T, CAUSAL, DP>), dim3(kv_grid), dim3(256), lds_kv, stream, p);
    return hipGetLastError();
}

hipError_t launch_bwd_16(const BwdParams& p, hipStream_t stream, const char** name) {
    if (!bwd_16_supported(p)) return hipErrorNotSupported;
    const bool bf = p.in_prec == P_BF16;
    if (p.D == 128) {
        *name = bf ? "fa_bwd16<bf16,128>" : "fa_bwd16<fp16,128>";
        if (bf) return p.causal ? launch_bwd16_t<__bf16, true, 128>(p, stream) : launch_bwd16_t<__bf16, false, 128>(p, stream);
#ifdef BWD16_LAB_ONLY128  // lab builds (fast asm inspection): bf16 head_dim 128 only
        return hipErrorNotSupported;
    }
    return hipErrorNotSupported;
}
#else
        return p.causal ? launch_bwd16_t<_Float16, true, 128>(p, stream) : launch_bwd16_t<_Float16, false, 128>(p, stream);
    }
    if (p.D == 256) {
        *name = bf ? "fa_bwd16<bf16,256>" : "fa_bwd16<fp16,256>";
        if (bf) return p.causal ? launch_bwd16_t<__bf16, true, 256>(p, stream) : launch_bwd16_t<__bf16, false, 256>(p, stream);
        return p.causal ? launch_bwd16_t<_Float16, true, 256>(p, stream) : launch_bwd16_t<_Float16, false, 256>(p, stream);
    }
    *name = bf ? "fa_bwd16<bf16,64>" : "fa_bwd16<fp16,64>";
    if (bf) return p.causal ? launch_bwd16_t<__bf16, true, 64>(p, stream) : launch_bwd16_t<__bf16, false, 64>(p, stream);
    return p.causal ? launch_bwd16_t<_Float16, true, 64>(p, stream) : launch_bwd16_t<_Float16, false, 64>(p, stream);
}
#endif

}  // namespace umfa
