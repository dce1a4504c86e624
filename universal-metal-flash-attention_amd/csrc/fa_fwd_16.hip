// fa_fwd_16.hip -- bf16 / fp16 flash-attention forward for gfx950 (the headline path).
//
// Replaces the Metal `attention` forward kernel the reference generates in its absent submodule
// and launches from MultiHeadAttention.forward / encodeForward (MFABridge.swift:2240-2248,
// 2525-2541): O = softmax(scale QK^T [+causal] [+mask]) V with online softmax, never
// materialising S, and WITHOUT the reference's dense fp32 [B,H,Sq,Skv] mask pre-pass
// (MFABridge.swift:368-590) -- the strided mask is read in-tile.
//
// Structure (cdna_hip_programming.md "Fused attention prefill"):
//   workgroup = 4 waves = 128 query rows of one (batch, head); 2 workgroups per CU (<= 256 VGPR);
//   key/value tiles of 64 rows, double-buffered in LDS, staged through registers with the
//   issue-early / write-late split (T14); one barrier per tile;
//   S^T = K Q^T on v_mfma_f32_32x32x16 (swapped operands, T12): a lane owns one query row, so
//   row max / row sum are in-lane plus one exchange with lane^32;
//   P^T (the S^T accumulator, rounded to the input type) is the B operand of O^T += V^T P^T with
//   no cross-lane movement; V^T fragments come from ds_read_b64_tr_b16 (T10);
//   K rows are XOR-swizzled for conflict-free ds_read_b128 (T2), V rows for the transposed reads.
//   Softmax is computed in the log2 domain: p = exp2(s * scale*log2e - m).
#include <cstdlib>

#include "fa_common.h"
#include "fa_fwd_16_launch.h"

namespace umfa {

bool fwd_16_supported(const FwdParams& p) {
    if (p.in_prec != P_FP16 && p.in_prec != P_BF16) return false;
    if (p.D == 0 || p.D % 8 != 0 || p.D > 256) return false;
    if (!(p.scale > 0.0f)) return false;
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    if (!al16(p.q) || !al16(p.k) || !al16(p.v) || !al16(p.o)) return false;
    for (int i = 0; i < 3; ++i)
        if (p.qs[i] % 8 || p.ks[i] % 8 || p.vs[i] % 8) return false;
    if (p.qs[3] != 1 || p.ks[3] != 1 || p.vs[3] != 1) return false;
    // 32-bit buffer offsets inside one (batch, head) slab
    const int64_t lim = (int64_t)1 << 30;  // elements (2 bytes each)
    if ((int64_t)p.Sq * p.qs[2] >= lim || (int64_t)p.Skv * p.ks[2] >= lim || (int64_t)p.Skv * p.vs[2] >= lim) return false;
    if (p.os[0] != (int64_t)p.D || p.os[1] != 1) return false;
    return true;
}

static int cu_count() { return device_cu_count(); }

static inline uint32_t dp16_of(uint32_t D) { return D <= 32 ? 32 : D <= 64 ? 64 : D <= 128 ? 128 : 256; }

// Split-KV plan.  Rounds 1-3 (tools/bench_one.py, B1 S4096 D128; the fold then cost an agent-scope release fence per part): splitting paid exactly
// when there were fewer work items than CUs (64 items: 90 -> 48 us with 4 parts; 128 items: 95 -> 73 us with 2) and lost once every CU had an item
// (256 items: 111 -> 120 us).  Round 4 (fence-free fold, fp16 P V conversion inside the kernel): the number of parts is the minimum of a small cost
// model -- see below -- for launches of at most one item per CU.
FwdSplitPlan fwd_16_split_plan(const FwdParams& p) {
    FwdSplitPlan plan;
    const uint32_t nqb = (p.Sq + 127) / 128, items = nqb * p.B * p.H;
    const uint32_t dp = dp16_of(p.D);
    const uint32_t cus = (uint32_t)cu_count();
    const uint32_t ntiles = (p.Skv + 63) / 64;
    plan.n_full = items;
    plan.nsplit = 1;
    plan.buf_bytes = plan.cnt_bytes = 0;
    if (items == 0) return plan;
    if (tuning().no_split.load(std::memory_order_relaxed)) {  // (lab: no parts; the decode form's gate as below)
        plan.decode = (!p.causal && fwd16_decode_shape(p) && (uint64_t)items * (dp == 128 ? 2u : 1u) <= cus) ? 1u : 0u;
        return plan;
    }
    if (p.causal) {
        // Causal items are uneven already; what a SHORT causal launch (every item resident at once) waits for is its
        // longest item: q-block nqb - 1 sweeps every key tile alone.  Lab option force_split = 2: cut the heavy half of each
        // head's q-blocks (qb >= nqb / 2) into two key ranges.  MEASURED SLOWER with this kernel's release / acquire fold
        // (round 3, graph-replayed us, split / whole: B4 H16 S1024 D64 [BASELINE config 2] 40.4 / 20.7, B8 H16 S512 D64
        // 37.8 / 14.0, B2 H16 S2048 D64 48.5 / 31.6, B1 H32 S2048 D128 81.7 / 55.1): 512 agent-scope release fences in a 20-us
        // launch cost more than the halved critical path returns.  Off by default; results are identical to 1e-2.
        const int force = tuning().force_split.load(std::memory_order_relaxed);
        const bool want = force == 2;
        // Round 6: balanced causal pairs (fa_fwd_16_kernel.h CBAL) -- the pair (i, nqb - 1 - i) of a head's q-blocks on two workgroups of
        // EQUAL length, the long q-block's tail published mid-sweep by the part that goes on with the short one: as many workgroups as
        // before, one fold per pair, nobody runs alone.  Where (profiles/r6/cbal_matrix.jsonl, graph-replayed, both schedules forced in one
        // process; unpaired -> paired us): head_dim 128 from eight q-blocks per head up to four workgroups per CU -- B1 H8 S4096 102 -> 67,
        // B2 H8 S2048 56 -> 40, B4 H8 S1024 32.6 -> 27.7, B8 H8 S2048 122 -> 109; at 2048 items level, S = 512 slower (two tiles against a
        // fold); head_dim 64 from sixteen q-blocks up to two per CU -- B1 H8 S4096 58 -> 49, B4 H8 S2048 36.3 -> 34.6.  NOT BASELINE config 2
        // (B4 H16 S1024 D64: 23.6 -> 24.3): in-kernel stamps show why -- with every CU holding two 9-tile workgroups for the whole launch
        // a tile takes 1.83 us against 1.2 us for a workgroup alone (the SIMD's two waves do not overlap at head_dim 64), so 18 tile steps
        // per CU cost what 16 alone + 2 shared did (profiles/r6/cbal_stamps_config2.txt).
        const int cb = tuning().cbal.load(std::memory_order_relaxed);
        const bool cb_shape = nqb >= 2 && (dp == 64 || dp == 128) && p.D == dp && p.mask_kind == MK_NONE && dma_enabled();  // (odd nqb: the middle q-block whole)
        const bool cb_auto = p.Skv >= p.Sq && (dp == 128 ? (nqb >= 8 && items <= 4 * cus) : (nqb >= 16 && items <= 2 * cus));
        if (!want && cb != 2 && cb_shape && (cb == 1 || cb_auto)) {
            const int dl = tuning().cbal_delta.load(std::memory_order_relaxed);  // (< 0: the plan's own choice)
            const size_t npairs = (size_t)p.B * p.H * (nqb / 2);
            plan.cbal = 1;
            // the folding part is shorter by the fold's price in tiles: nothing while a CU holds one workgroup, two tiles from two per CU on (head_dim 128)
            plan.cbal_delta = dl >= 0 ? (uint32_t)(dl > 16 ? 16 : dl) : (dp == 128 && items > cus ? 2u : 0u);
            plan.buf_bytes = npairs * 4 * (4 * (dp / 32) + 1) * 1024;
            plan.cnt_bytes = (npairs * sizeof(uint32_t) + 15) & ~(size_t)15;
            return plan;
        }
        if (!want || (nqb & 1) || nqb < 4 || p.Sq != p.Skv || dp > 128) return plan;
        plan.n_full = items / 2;
        plan.nsplit = 2;
        plan.buf_bytes = (size_t)(items / 2) * 2 * 4 * (16 * (dp / 32) + 4) * 64 * sizeof(float);
        plan.cnt_bytes = ((size_t)(items / 2) * sizeof(uint32_t) + 15) & ~(size_t)15;
        return plan;
    }
    // How many parts (round 4, from measurements: profiles/r4/decode_k_probe.jsonl, split_gap_probe.jsonl, split_plan_random*.jsonl).  A launch of
    // items x k workgroups puts w = ceil(items k / CUs) of them on the fullest CU; two co-resident workgroups run ~1.6 x as fast as one after the
    // other, a third waits for a slot (R = 2 workgroups fit; one at head_dim 256); a part costs the fold ~3 key tiles (the folding workgroup reads the
    // parts one after the other, a dependent round trip of write-through memory each).  In key tiles:
    //     cost(k) = ntiles / k x f(w) + c k,   f(1) = 1, f(2) = 1.25, f(3) = 2.25, f(4) = 2.5, ...
    // c = 3 in rounds 4-5; round 6: 2 -- the fold takes 16-byte loads, reads part o + 1 while it folds part o, and waves without rows (decode: three of
    // four) stay out of it: B1 H8 Sq1 Skv131072 202 -> 139 us, B1 H32 Sq1 Skv8192 55 -> 34, B16 H8 Sq1 Skv4096 65 -> 48.  Over 16 launch sizes with every part
    // count forced (profiles/r6/split_plan_probe.jsonl) c = 2 picks within 4 % of the best count everywhere (geometric mean 0.7 %); c = 3 was 21 % off at worst.
    // and the plan is its minimum over k = 1 ... min(kmax, ntiles / 4).  This replaces three rules that each fitted one regime: "one workgroup per CU"
    // (left 192 workgroups on 256 CUs at 96 items), "two per CU for decode-like calls" (B1 H32 Sq1 Skv8192: 16 parts 75 us, 8 parts 51), "no split
    // above half an item per CU" (160 items, three parts: 92.7 -> 69.5 us).  Checked against every forced part count: split_plan_random*.jsonl.
    // Decode form of the kernel (fa_fwd_16_kernel.h KS = 4, round 6: <= 32 query rows, the four waves split every 128-key tile's keys): one workgroup per CU at
    // head_dim 128 (128 KiB of LDS), two at 64 -- so it is taken while the items fit one round of that residency, and the part count below is chosen for it.
    // Graph-replayed, both forms alternating in one process, each with its best part count (profiles/r6/decode_form_ab.txt): B1 H8 Sq4 Skv8192 35.4 -> 26.3-29.5 us,
    // B1 H64 Sq1 Skv4096 D64 21.1 -> 17.8, B2 H16 Sq16 Skv16384 56.1 -> 50.5, B16 H8 Sq1 Skv4096 48.9 -> 45.9; from one item per CU on the plain form's two
    // workgroups per CU win (B8 H32 Sq1 Skv8192 185 against 204) and keep the launch.
    const int dks = tuning().decode_ks.load(std::memory_order_relaxed);
    plan.decode = (fwd16_decode_shape(p) && (dks == 1 || (uint64_t)items * (dp == 128 ? 2u : 1u) <= cus)) ? 1u : 0u;
    const uint32_t kmax = nqb == 1 ? 32u : 8u;
    const uint32_t R = (dp > 128 || (plan.decode && dp == 128)) ? 1u : 2u;
    const int force = tuning().force_split.load(std::memory_order_relaxed);  // experiments: split every item k ways
    uint32_t k = 1;
    if (force >= 2 && force <= 32) {
        k = (uint32_t)force;
        if (k > ntiles / 4) k = ntiles / 4;
    } else if (items <= cus) {  // (more than one item per CU: the dispatcher refills slots as they free up, splitting only adds folds)
        double best = 1e30;
        for (uint32_t kk = 1; kk <= kmax && (kk == 1 || kk <= ntiles / 4); ++kk) {
            const uint64_t w = ((uint64_t)items * kk + cus - 1) / cus;
            const uint64_t rounds = (w + R - 1) / R, last = w - (rounds - 1) * R;  // full rounds of R co-resident workgroups, then `last`
            const double f = (double)(rounds - 1) * (R == 2 ? 1.25 : 1.0) + (last == 2 ? 1.25 : 1.0);
            const double cost = (double)ntiles / kk * f + (kk > 1 ? 2.0 * kk : 0.0);  // (round 6: 2 tiles per part -- the fold reads a part ahead, below)
            if (cost < best * 0.97) { best = cost; k = kk; }  // (a tie goes to fewer parts)
        }
    }
    if (k < 2) return plan;
    plan.n_full = 0;
    plan.nsplit = k;
    plan.buf_bytes = (size_t)items * k * 4 * (16 * (dp / 32) + 4) * 64 * sizeof(float);
    plan.cnt_bytes = ((size_t)items * sizeof(uint32_t) + 15) & ~(size_t)15;
    return plan;
}

// LDS-DMA staging when head_dim fills the padded row exactly; register staging otherwise.
template <typename T, int DP, bool CAUSAL, bool HAS_MASK, typename OUT>
static hipError_t launch_one(const FwdParams& p, hipStream_t stream) {
    if constexpr (__is_same(T, __bf16)) {
        if (p.pv16) return launch_fwd16_pv<DP, CAUSAL, HAS_MASK, OUT>(p, stream);  // bf16 operands, fp16 P V (the default): fa_fwd_16_pv.hip
    }
    if ((int)p.D == DP && dma_enabled()) {
        if constexpr (!CAUSAL && !HAS_MASK && (DP == 64 || DP == 128)) {
            if (fwd16_decode_form(p)) return launch_dma<T, DP, CAUSAL, HAS_MASK, OUT, true, 128, 0, 4>(p, stream);  // decode form: four key quarters per tile
        }
        // 32-key tiles + LDS-DMA at head_dim 128: 166 VGPR / 32 KiB LDS -> three resident workgroups per CU
        // (lab, same box: FLUX 768 items 264 -> 242 us; 3072 items 895 -> 870 us; never slower)
        // (causal launches lose with it: 184 vs 146 us at the FLUX shape, so they keep 64-key tiles)
        if constexpr (DP == 128 && !HAS_MASK && !CAUSAL) {
            if (!tuning().bn64.load(std::memory_order_relaxed)) return launch_dma<T, DP, CAUSAL, HAS_MASK, OUT, true, 32>(p, stream);
        }
        if constexpr (DP == 64 && !HAS_MASK) {
#ifdef UMFA_D64_FORMS
            const int form = fwd16_d64_form(p);
            if (form == 1) return launch_dma<T, DP, CAUSAL, HAS_MASK, OUT, true, 64, 0, 1, 1>(p, stream);
            if (form == 2) return launch_dma<T, DP, CAUSAL, HAS_MASK, OUT, true, 64, 0, 2>(p, stream);
#endif
        }
        if constexpr (CAUSAL && !HAS_MASK && (DP == 64 || DP == 128)) {
            if (p.cbal && p.part_buf && p.part_cnt) return launch_dma<T, DP, CAUSAL, HAS_MASK, OUT, true, 64, 0, 1, 0, true>(p, stream);
        }
        return launch_dma<T, DP, CAUSAL, HAS_MASK, OUT, true, 64>(p, stream);
    }
    return launch_dma<T, DP, CAUSAL, HAS_MASK, OUT, false, 64>(p, stream);
}

template <typename T, int DP, typename OUT>
static hipError_t launch_flags(const FwdParams& p, hipStream_t stream) {
    const bool mk = p.mask_kind != MK_NONE;
    if (p.causal) return mk ? launch_one<T, DP, true, true, OUT>(p, stream) : launch_one<T, DP, true, false, OUT>(p, stream);
    return mk ? launch_one<T, DP, false, true, OUT>(p, stream) : launch_one<T, DP, false, false, OUT>(p, stream);
}

template <typename T, int DP>
static hipError_t launch_out(const FwdParams& p, hipStream_t stream) {
    if (p.out_prec == P_FP32) return launch_flags<T, DP, float>(p, stream);
    // 16-bit epilogue only in the input type (the torch binding's cast-back, metal_sdpa_backend.cpp:1442-1444)
    if (p.out_prec == p.in_prec) return launch_flags<T, DP, T>(p, stream);
    return hipErrorNotSupported;
}

template <typename T>
static hipError_t launch_dp(const FwdParams& p, hipStream_t stream, const char** name) {
    constexpr bool bf = sizeof(T) == 2 && __is_same(T, __bf16);
    const bool pv = bf && p.pv16;
    if (p.D <= 32) { *name = pv ? "fa_fwd16<bf16,32,pv16>" : bf ? "fa_fwd16<bf16,32>" : "fa_fwd16<fp16,32>"; return launch_out<T, 32>(p, stream); }
    if (p.D <= 64) {
        const int form = fwd16_d64_form(p);
        if (form == 1) *name = pv ? "fa_fwd16<bf16,64,pv16,pipe>" : bf ? "fa_fwd16<bf16,64,pipe>" : "fa_fwd16<fp16,64,pipe>";
        else if (form == 2) *name = pv ? "fa_fwd16<bf16,64,pv16,ks2>" : bf ? "fa_fwd16<bf16,64,ks2>" : "fa_fwd16<fp16,64,ks2>";
        else *name = pv ? "fa_fwd16<bf16,64,pv16>" : bf ? "fa_fwd16<bf16,64>" : "fa_fwd16<fp16,64>";
        if (fwd16_decode_form(p)) *name = pv ? "fa_fwd16<bf16,64,pv16,dec>" : bf ? "fa_fwd16<bf16,64,dec>" : "fa_fwd16<fp16,64,dec>";
        return launch_out<T, 64>(p, stream);
    }
    const bool dec = fwd16_decode_form(p);  // (launch_one / launch_fwd16_pv take the decode form then)
    if (p.D <= 128) {
        *name = dec ? (pv ? "fa_fwd16<bf16,128,pv16,dec>" : bf ? "fa_fwd16<bf16,128,dec>" : "fa_fwd16<fp16,128,dec>")
                    : (pv ? "fa_fwd16<bf16,128,pv16>" : bf ? "fa_fwd16<bf16,128>" : "fa_fwd16<fp16,128>");
        return launch_out<T, 128>(p, stream);
    }
    *name = pv ? "fa_fwd16<bf16,256,pv16>" : bf ? "fa_fwd16<bf16,256>" : "fa_fwd16<fp16,256>";
    return launch_out<T, 256>(p, stream);
}

hipError_t launch_fwd_16(const FwdParams& p, hipStream_t stream, const char** name) {
    if (!fwd_16_supported(p)) return hipErrorNotSupported;
    return p.in_prec == P_BF16 ? launch_dp<__bf16>(p, stream, name) : launch_dp<_Float16>(p, stream, name);
}

}  // namespace umfa
