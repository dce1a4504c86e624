// How many VALU issue cycles really hide beside one MFMA on gfx950 (one wave per SIMD, every CU busy)?
// Each variant: a loop body of REP x { one MFMA ; a fixed filler mix on independent registers }, clock-stamped.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/lab_bin/coissue_probe tools/lab/coissue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#include <vector>

#define STR2(x) #x
#define STR(x) STR2(x)

// fillers operate on v[200:231] (never read by the MFMAs), MFMAs on a[0:63] / v[232:247]
#define F_EXP(r) "v_exp_f32 v" STR(r) ", v" STR(r) "\n\t"
#define F_FMA(r) "v_fma_f32 v" STR(r) ", v" STR(r) ", v248, v249\n\t"
#define F_ADD(r) "v_add_f32 v250, v250, v" STR(r) "\n\t"
#define F_ADD2(r) "v_add_f32 v251, v251, v" STR(r) "\n\t"
#define F_CVT(r) "v_cvt_pk_bf16_f32 v" STR(r) ", v" STR(r) ", v" STR(r) "\n\t"
#define F_MAX(r) "v_max3_f32 v252, v252, v" STR(r) ", v" STR(r) "\n\t"
#define F_CVT8(r) "v_cvt_pk_fp8_f32 v" STR(r) ", v" STR(r) ", v" STR(r) "\n\t"
#define MFMA_BF(acc) "v_mfma_f32_32x32x16_bf16 a[" STR(acc) ":" STR(acc) "+15], v[232:235], v[236:239], a[" STR(acc) ":" STR(acc) "+15]\n\t"
#define MFMA_F8(acc) "v_mfma_scale_f32_32x32x64_f8f6f4 a[" STR(acc) ":" STR(acc) "+15], v[232:239], v[240:247], a[" STR(acc) ":" STR(acc) "+15], v253, v253 op_sel_hi:[0,0,0]\n\t"
#define MFMA_I8(acc) "v_mfma_i32_32x32x32_i8 a[" STR(acc) ":" STR(acc) "+15], v[232:235], v[236:239], a[" STR(acc) ":" STR(acc) "+15]\n\t"

#define CLOB "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", \
             "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247",   \
             "v248", "v249", "v250", "v251", "v252", "v253", "a0", "a15", "a16", "a31", "a32", "a47", "a48", "a63", "v255", "a255"

template <int V>
__global__ __launch_bounds__(256, 1) void probe(unsigned long long* cyc, int n) {
    asm volatile("v_mov_b32 v248, 1.0\n\tv_mov_b32 v249, 0\n\tv_mov_b32 v250, 0\n\tv_mov_b32 v251, 0\n\tv_mov_b32 v252, 0\n\tv_mov_b32 v253, 0x7f7f7f7f" ::: CLOB);
    asm volatile("v_mov_b32 v232, 0x3f803f80\n\tv_mov_b32 v233, 0x3f803f80\n\tv_mov_b32 v234, 0x3f80bf80\n\tv_mov_b32 v235, 0x3f803f80\n\t"
                 "v_mov_b32 v236, 0x3f803f80\n\tv_mov_b32 v237, 0xbf803f80\n\tv_mov_b32 v238, 0x3f803f80\n\tv_mov_b32 v239, 0x3f803f80" ::: CLOB);
    asm volatile("v_mov_b32 v200, 0.5\n\tv_mov_b32 v201, 0.5\n\tv_mov_b32 v202, 0.5\n\tv_mov_b32 v203, 0.5\n\tv_mov_b32 v204, 0.5\n\tv_mov_b32 v205, 0.5\n\t"
                 "v_mov_b32 v206, 0.5\n\tv_mov_b32 v207, 0.5\n\tv_mov_b32 v208, 0.5\n\tv_mov_b32 v209, 0.5\n\tv_mov_b32 v210, 0.5\n\tv_mov_b32 v211, 0.5" ::: CLOB);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        // 4 "gaps" per iteration, accumulators a0 / a16 / a32 / a48
        if constexpr (V == 0) asm volatile(MFMA_BF(0) MFMA_BF(16) MFMA_BF(32) MFMA_BF(48) ::: CLOB);
        if constexpr (V == 1) asm volatile(MFMA_BF(0) F_EXP(200) MFMA_BF(16) F_EXP(201) MFMA_BF(32) F_EXP(202) MFMA_BF(48) F_EXP(203) ::: CLOB);
        if constexpr (V == 2) asm volatile(MFMA_BF(0) F_EXP(200) F_EXP(204) MFMA_BF(16) F_EXP(201) F_EXP(205) MFMA_BF(32) F_EXP(202) F_EXP(206) MFMA_BF(48) F_EXP(203) F_EXP(207) ::: CLOB);
        if constexpr (V == 3) asm volatile(MFMA_BF(0) F_FMA(200) F_FMA(201) F_FMA(202) F_FMA(203) MFMA_BF(16) F_FMA(204) F_FMA(205) F_FMA(206) F_FMA(207)
                                            MFMA_BF(32) F_FMA(200) F_FMA(201) F_FMA(202) F_FMA(203) MFMA_BF(48) F_FMA(204) F_FMA(205) F_FMA(206) F_FMA(207) ::: CLOB);
        if constexpr (V == 4) asm volatile(MFMA_BF(0) F_FMA(200) F_FMA(201) F_FMA(202) F_FMA(203) F_FMA(208) F_FMA(209) MFMA_BF(16) F_FMA(204) F_FMA(205) F_FMA(206) F_FMA(207) F_FMA(210) F_FMA(211)
                                            MFMA_BF(32) F_FMA(200) F_FMA(201) F_FMA(202) F_FMA(203) F_FMA(208) F_FMA(209) MFMA_BF(48) F_FMA(204) F_FMA(205) F_FMA(206) F_FMA(207) F_FMA(210) F_FMA(211) ::: CLOB);
        // the real per-gap mix of the bf16 attention tile: fma, exp, add, (cvt | max3) = 4.0 VALU per gap
        if constexpr (V == 5) asm volatile(MFMA_BF(0) F_FMA(200) F_EXP(204) F_ADD(208) F_CVT(209) MFMA_BF(16) F_FMA(201) F_EXP(205) F_ADD2(210) F_MAX(211)
                                            MFMA_BF(32) F_FMA(202) F_EXP(206) F_ADD(208) F_CVT(209) MFMA_BF(48) F_FMA(203) F_EXP(207) F_ADD2(210) F_MAX(211) ::: CLOB);
        // the same mix without MFMAs: pure VALU issue time
        if constexpr (V == 6) asm volatile(F_FMA(200) F_EXP(204) F_ADD(208) F_CVT(209) F_FMA(201) F_EXP(205) F_ADD2(210) F_MAX(211)
                                            F_FMA(202) F_EXP(206) F_ADD(208) F_CVT(209) F_FMA(203) F_EXP(207) F_ADD2(210) F_MAX(211) ::: CLOB);
        // fp8 PV: one 64-cycle MFMA per 8 scores' worth of VALU (2 x the mix) -- i.e. int8+fp8 tile = 16 i8 + 8 f8 MFMAs, 64 x mix
        if constexpr (V == 7) asm volatile(MFMA_F8(0) F_FMA(200) F_EXP(204) F_ADD(208) F_CVT8(209) F_FMA(201) F_EXP(205) F_ADD2(210) F_MAX(211)
                                            MFMA_F8(16) F_FMA(202) F_EXP(206) F_ADD(208) F_CVT8(209) F_FMA(203) F_EXP(207) F_ADD2(210) F_MAX(211) ::: CLOB);
        // int8+fp8 tile ratio: per 8 scores (8 x mix): 2 i8 MFMAs + 1 f8 MFMA
        if constexpr (V == 8) asm volatile(MFMA_I8(0) F_FMA(200) F_EXP(204) F_ADD(208) F_CVT8(209) F_FMA(201) F_EXP(205) F_ADD2(210) F_MAX(211)
                                            MFMA_I8(16) F_FMA(202) F_EXP(206) F_ADD(208) F_CVT8(209) F_FMA(203) F_EXP(207) F_ADD2(210) F_MAX(211)
                                            MFMA_F8(32) F_FMA(200) F_EXP(204) F_ADD(208) F_CVT8(209) F_FMA(201) F_EXP(205) F_ADD2(210) F_MAX(211)
                                            F_FMA(202) F_EXP(206) F_ADD(208) F_CVT8(209) F_FMA(203) F_EXP(207) F_ADD2(210) F_MAX(211) ::: CLOB);
        if constexpr (V == 9) asm volatile(F_EXP(200) F_EXP(201) F_EXP(202) F_EXP(203) F_EXP(204) F_EXP(205) F_EXP(206) F_EXP(207) F_EXP(208) F_EXP(209) F_EXP(210) F_EXP(211) F_EXP(200) F_EXP(201) F_EXP(202) F_EXP(203) ::: CLOB);
        if constexpr (V == 10) asm volatile(F_FMA(200) F_FMA(201) F_FMA(202) F_FMA(203) F_FMA(204) F_FMA(205) F_FMA(206) F_FMA(207) F_FMA(208) F_FMA(209) F_FMA(210) F_FMA(211) F_FMA(200) F_FMA(201) F_FMA(202) F_FMA(203) ::: CLOB);
        // bf16 mix with only 3 VALU per gap (what dropping one instruction per score would buy)
        if constexpr (V == 11) asm volatile(MFMA_BF(0) F_EXP(204) F_ADD(208) F_CVT(209) MFMA_BF(16) F_EXP(205) F_ADD2(210) F_MAX(211)
                                             MFMA_BF(32) F_EXP(206) F_ADD(208) F_CVT(209) MFMA_BF(48) F_EXP(207) F_ADD2(210) F_MAX(211) ::: CLOB);
        // i8 QK half of the current int8 kernel: 16 i8 + 32 f16 MFMAs per 64 x mix = per 4 mix: 1 i8 + 2 bf16
        if constexpr (V == 12) asm volatile(MFMA_I8(0) F_FMA(200) F_EXP(204) F_ADD(208) F_CVT(209) MFMA_BF(16) F_FMA(201) F_EXP(205) F_ADD2(210) F_MAX(211)
                                             MFMA_BF(32) F_FMA(202) F_EXP(206) F_ADD(208) F_CVT(209) F_FMA(203) F_EXP(207) F_ADD2(210) F_MAX(211) ::: CLOB);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V>
void run(const char* name, double valu_model, unsigned long long* cyc) {
    const int n = 4000;
    for (int rep = 0; rep < 2; ++rep) { probe<V><<<256, 256>>>(cyc, n); hipDeviceSynchronize(); }
    double s = 0; for (int b = 0; b < 256; ++b) s += (double)cyc[b];
    printf("%-86s %8.1f cycles / iteration   (issue model: %.0f)\n", name, s / 256 / n, valu_model);
}

int main() {
    unsigned long long* cyc;
    hipMallocManaged(&cyc, 256 * 8);
    run<0>("4 x mfma bf16 32x32x16", 128, cyc);
    run<1>("4 x (mfma bf16 + 1 exp)", 128, cyc);
    run<2>("4 x (mfma bf16 + 2 exp)", 128, cyc);
    run<3>("4 x (mfma bf16 + 4 fma)", 128, cyc);
    run<4>("4 x (mfma bf16 + 6 fma)", 128, cyc);
    run<5>("4 x (mfma bf16 + fma exp add cvt|max3)   = the bf16 tile's gap", 4 * (8 + 4 + 8 + 4 + 4), cyc);
    run<6>("4 x (fma exp add cvt|max3), no MFMA", 4 * 20, cyc);
    run<7>("2 x (mfma fp8 32x32x64 + 2 x mix)", 160, cyc);
    run<8>("2 mfma i8 + 1 mfma fp8 + 8 x mix  = 1/8 of the int8+fp8 tile", 160 + 24, cyc);
    run<9>("16 exp", 128, cyc);
    run<10>("16 fma", 64, cyc);
    run<11>("4 x (mfma bf16 + exp add cvt|max3): 3 VALU per gap", 4 * (8 + 16), cyc);
    run<12>("1 mfma i8 + 2 mfma bf16 + 4 x mix = 1/16 of the current int8 tile", 3 * 8 + 80, cyc);
    return 0;
}
