// VALU issue rate of ONE wave per SIMD vs TWO, by instruction form (long unrolled bodies: branch cost amortised).
#include <hip/hip_runtime.h>
#include <cstdio>
#define R8(x) x x x x x x x x
#define CLOB "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","a0","a15","a16","a31"
// 16 independent destination registers v100..v115; sources v116..v123
#define FMA16 "v_fma_f32 v100, v100, v116, v117\n v_fma_f32 v101, v101, v118, v119\n v_fma_f32 v102, v102, v120, v121\n v_fma_f32 v103, v103, v122, v123\n" \
              "v_fma_f32 v104, v104, v117, v118\n v_fma_f32 v105, v105, v119, v120\n v_fma_f32 v106, v106, v121, v122\n v_fma_f32 v107, v107, v123, v116\n" \
              "v_fma_f32 v108, v108, v118, v119\n v_fma_f32 v109, v109, v120, v121\n v_fma_f32 v110, v110, v122, v123\n v_fma_f32 v111, v111, v116, v117\n" \
              "v_fma_f32 v112, v112, v119, v120\n v_fma_f32 v113, v113, v121, v122\n v_fma_f32 v114, v114, v123, v116\n v_fma_f32 v115, v115, v117, v118\n"
#define ADD16 "v_add_f32 v100, v100, v116\n v_add_f32 v101, v101, v117\n v_add_f32 v102, v102, v118\n v_add_f32 v103, v103, v119\n" \
              "v_add_f32 v104, v104, v120\n v_add_f32 v105, v105, v121\n v_add_f32 v106, v106, v122\n v_add_f32 v107, v107, v123\n" \
              "v_add_f32 v108, v108, v116\n v_add_f32 v109, v109, v117\n v_add_f32 v110, v110, v118\n v_add_f32 v111, v111, v119\n" \
              "v_add_f32 v112, v112, v120\n v_add_f32 v113, v113, v121\n v_add_f32 v114, v114, v122\n v_add_f32 v115, v115, v123\n"
#define EXP16 "v_exp_f32 v100, v100\n v_exp_f32 v101, v101\n v_exp_f32 v102, v102\n v_exp_f32 v103, v103\n v_exp_f32 v104, v104\n v_exp_f32 v105, v105\n v_exp_f32 v106, v106\n v_exp_f32 v107, v107\n" \
              "v_exp_f32 v108, v108\n v_exp_f32 v109, v109\n v_exp_f32 v110, v110\n v_exp_f32 v111, v111\n v_exp_f32 v112, v112\n v_exp_f32 v113, v113\n v_exp_f32 v114, v114\n v_exp_f32 v115, v115\n"
#define FMAS16 "v_fma_f32 v100, v100, s4, v117\n v_fma_f32 v101, v101, s4, v119\n v_fma_f32 v102, v102, s4, v121\n v_fma_f32 v103, v103, s4, v123\n" \
              "v_fma_f32 v104, v104, s4, v118\n v_fma_f32 v105, v105, s4, v120\n v_fma_f32 v106, v106, s4, v122\n v_fma_f32 v107, v107, s4, v116\n" \
              "v_fma_f32 v108, v108, s4, v119\n v_fma_f32 v109, v109, s4, v121\n v_fma_f32 v110, v110, s4, v123\n v_fma_f32 v111, v111, s4, v117\n" \
              "v_fma_f32 v112, v112, s4, v120\n v_fma_f32 v113, v113, s4, v122\n v_fma_f32 v114, v114, s4, v116\n v_fma_f32 v115, v115, s4, v118\n"
#define CVT16 "v_cvt_pk_bf16_f32 v100, v116, v117\n v_cvt_pk_bf16_f32 v101, v118, v119\n v_cvt_pk_bf16_f32 v102, v120, v121\n v_cvt_pk_bf16_f32 v103, v122, v123\n" \
              "v_cvt_pk_bf16_f32 v104, v116, v117\n v_cvt_pk_bf16_f32 v105, v118, v119\n v_cvt_pk_bf16_f32 v106, v120, v121\n v_cvt_pk_bf16_f32 v107, v122, v123\n" \
              "v_cvt_pk_bf16_f32 v108, v116, v117\n v_cvt_pk_bf16_f32 v109, v118, v119\n v_cvt_pk_bf16_f32 v110, v120, v121\n v_cvt_pk_bf16_f32 v111, v122, v123\n" \
              "v_cvt_pk_bf16_f32 v112, v116, v117\n v_cvt_pk_bf16_f32 v113, v118, v119\n v_cvt_pk_bf16_f32 v114, v120, v121\n v_cvt_pk_bf16_f32 v115, v122, v123\n"
#define MAX16 "v_max3_f32 v100, v100, v116, v117\n v_max3_f32 v101, v101, v118, v119\n v_max3_f32 v102, v102, v120, v121\n v_max3_f32 v103, v103, v122, v123\n" \
              "v_max3_f32 v104, v104, v117, v118\n v_max3_f32 v105, v105, v119, v120\n v_max3_f32 v106, v106, v121, v122\n v_max3_f32 v107, v107, v123, v116\n" \
              "v_max3_f32 v108, v108, v118, v119\n v_max3_f32 v109, v109, v120, v121\n v_max3_f32 v110, v110, v122, v123\n v_max3_f32 v111, v111, v116, v117\n" \
              "v_max3_f32 v112, v112, v119, v120\n v_max3_f32 v113, v113, v121, v122\n v_max3_f32 v114, v114, v123, v116\n v_max3_f32 v115, v115, v117, v118\n"
#define MFMA "v_mfma_f32_32x32x16_bf16 a[0:15], v[116:119], v[120:123], a[0:15]\n"
// one MFMA + k VALU (fma, VOP3 three VGPR sources)
#define M_F4 MFMA "v_fma_f32 v100, v100, v116, v117\n v_fma_f32 v101, v101, v118, v119\n v_fma_f32 v102, v102, v120, v121\n v_fma_f32 v103, v103, v122, v123\n"
#define M_A4 MFMA "v_add_f32 v100, v100, v116\n v_add_f32 v101, v101, v117\n v_add_f32 v102, v102, v118\n v_add_f32 v103, v103, v119\n"
#define M_A5 M_A4 "v_add_f32 v104, v104, v120\n"
#define M_A6 M_A5 "v_add_f32 v105, v105, v121\n"
#define M_MIX MFMA "v_fma_f32 v100, v100, s4, v117\n v_exp_f32 v101, v101\n v_add_f32 v102, v102, v118\n v_cvt_pk_bf16_f32 v103, v120, v121\n"
#define M_MIX5 M_MIX "v_max3_f32 v104, v104, v122, v123\n"

#define PKFMA8 "v_pk_fma_f32 v[100:101], v[100:101], v[116:117], v[118:119]\n v_pk_fma_f32 v[102:103], v[102:103], v[118:119], v[120:121]\n v_pk_fma_f32 v[104:105], v[104:105], v[120:121], v[122:123]\n v_pk_fma_f32 v[106:107], v[106:107], v[122:123], v[116:117]\n v_pk_fma_f32 v[108:109], v[108:109], v[116:117], v[118:119]\n v_pk_fma_f32 v[110:111], v[110:111], v[118:119], v[120:121]\n v_pk_fma_f32 v[112:113], v[112:113], v[120:121], v[122:123]\n v_pk_fma_f32 v[114:115], v[114:115], v[122:123], v[116:117]\n "
#define PKFMAB8 "v_pk_fma_f32 v[100:101], v[100:101], v[116:117], v[118:119] op_sel_hi:[1,0,0]\n v_pk_fma_f32 v[102:103], v[102:103], v[118:119], v[120:121] op_sel_hi:[1,0,0]\n v_pk_fma_f32 v[104:105], v[104:105], v[120:121], v[122:123] op_sel_hi:[1,0,0]\n v_pk_fma_f32 v[106:107], v[106:107], v[122:123], v[116:117] op_sel_hi:[1,0,0]\n v_pk_fma_f32 v[108:109], v[108:109], v[116:117], v[118:119] op_sel_hi:[1,0,0]\n v_pk_fma_f32 v[110:111], v[110:111], v[118:119], v[120:121] op_sel_hi:[1,0,0]\n v_pk_fma_f32 v[112:113], v[112:113], v[120:121], v[122:123] op_sel_hi:[1,0,0]\n v_pk_fma_f32 v[114:115], v[114:115], v[122:123], v[116:117] op_sel_hi:[1,0,0]\n "
#define PKADD8 "v_pk_add_f32 v[100:101], v[100:101], v[116:117]\n v_pk_add_f32 v[102:103], v[102:103], v[118:119]\n v_pk_add_f32 v[104:105], v[104:105], v[120:121]\n v_pk_add_f32 v[106:107], v[106:107], v[122:123]\n v_pk_add_f32 v[108:109], v[108:109], v[116:117]\n v_pk_add_f32 v[110:111], v[110:111], v[118:119]\n v_pk_add_f32 v[112:113], v[112:113], v[120:121]\n v_pk_add_f32 v[114:115], v[114:115], v[122:123]\n "
#define PKMUL8 "v_pk_mul_f32 v[100:101], v[100:101], v[116:117]\n v_pk_mul_f32 v[102:103], v[102:103], v[118:119]\n v_pk_mul_f32 v[104:105], v[104:105], v[120:121]\n v_pk_mul_f32 v[106:107], v[106:107], v[122:123]\n v_pk_mul_f32 v[108:109], v[108:109], v[116:117]\n v_pk_mul_f32 v[110:111], v[110:111], v[118:119]\n v_pk_mul_f32 v[112:113], v[112:113], v[120:121]\n v_pk_mul_f32 v[114:115], v[114:115], v[122:123]\n "
#define M_PK MFMA "v_pk_fma_f32 v[100:101], v[100:101], v[116:117], v[118:119] op_sel_hi:[1,0,0]\n v_exp_f32 v102, v102\n v_exp_f32 v103, v103\n v_pk_add_f32 v[104:105], v[104:105], v[120:121]\n v_cvt_pk_bf16_f32 v106, v120, v121\n"
#define M_PK6 M_PK "v_max3_f32 v107, v107, v122, v123\n"
#define DOT16 "v_dot2_f32_bf16 v100, v116, v119, v100\n v_dot2_f32_bf16 v101, v117, v120, v101\n v_dot2_f32_bf16 v102, v118, v121, v102\n v_dot2_f32_bf16 v103, v119, v122, v103\n v_dot2_f32_bf16 v104, v120, v123, v104\n v_dot2_f32_bf16 v105, v121, v116, v105\n v_dot2_f32_bf16 v106, v122, v117, v106\n v_dot2_f32_bf16 v107, v123, v118, v107\n v_dot2_f32_bf16 v108, v116, v119, v108\n v_dot2_f32_bf16 v109, v117, v120, v109\n v_dot2_f32_bf16 v110, v118, v121, v110\n v_dot2_f32_bf16 v111, v119, v122, v111\n v_dot2_f32_bf16 v112, v120, v123, v112\n v_dot2_f32_bf16 v113, v121, v116, v113\n v_dot2_f32_bf16 v114, v122, v117, v114\n v_dot2_f32_bf16 v115, v123, v118, v115\n "
#define DOTC16 "v_dot2c_f32_bf16 v100, v116, v119\n v_dot2c_f32_bf16 v101, v117, v120\n v_dot2c_f32_bf16 v102, v118, v121\n v_dot2c_f32_bf16 v103, v119, v122\n v_dot2c_f32_bf16 v104, v120, v123\n v_dot2c_f32_bf16 v105, v121, v116\n v_dot2c_f32_bf16 v106, v122, v117\n v_dot2c_f32_bf16 v107, v123, v118\n v_dot2c_f32_bf16 v108, v116, v119\n v_dot2c_f32_bf16 v109, v117, v120\n v_dot2c_f32_bf16 v110, v118, v121\n v_dot2c_f32_bf16 v111, v119, v122\n v_dot2c_f32_bf16 v112, v120, v123\n v_dot2c_f32_bf16 v113, v121, v116\n v_dot2c_f32_bf16 v114, v122, v117\n v_dot2c_f32_bf16 v115, v123, v118\n "
#define M_DOT MFMA "v_fma_f32 v100, v100, s4, v117\n v_exp_f32 v101, v101\n v_dot2_f32_bf16 v102, v118, v119, v102\n v_cvt_pk_bf16_f32 v103, v120, v121\n v_dot2c_f32_bf16 v104, v120, v121\n"
template <int V>
__global__ void probe(unsigned long long* cyc, int n) {
    asm volatile("v_mov_b32 v116, 1.0\n v_mov_b32 v117, 0\n v_mov_b32 v118, 1.0\n v_mov_b32 v119, 0\n v_mov_b32 v120, 1.0\n v_mov_b32 v121, 0\n v_mov_b32 v122, 1.0\n v_mov_b32 v123, 0\n s_mov_b32 s4, 1.0" ::: CLOB, "s4");
    asm volatile("v_mov_b32 v100, 0.5\n v_mov_b32 v101, 0.5\n v_mov_b32 v102, 0.5\n v_mov_b32 v103, 0.5\n v_mov_b32 v104, 0.5\n v_mov_b32 v105, 0.5\n v_mov_b32 v106, 0.5\n v_mov_b32 v107, 0.5\n"
                 "v_mov_b32 v108, 0.5\n v_mov_b32 v109, 0.5\n v_mov_b32 v110, 0.5\n v_mov_b32 v111, 0.5\n v_mov_b32 v112, 0.5\n v_mov_b32 v113, 0.5\n v_mov_b32 v114, 0.5\n v_mov_b32 v115, 0.5" ::: CLOB);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        if constexpr (V == 0) asm volatile(R8(FMA16) ::: CLOB);
        if constexpr (V == 1) asm volatile(R8(ADD16) ::: CLOB);
        if constexpr (V == 2) asm volatile(R8(EXP16) ::: CLOB);
        if constexpr (V == 3) asm volatile(R8(FMAS16) ::: CLOB, "s4");
        if constexpr (V == 4) asm volatile(R8(CVT16) ::: CLOB);
        if constexpr (V == 5) asm volatile(R8(MAX16) ::: CLOB);
        if constexpr (V == 6) asm volatile(R8(M_F4 M_F4) ::: CLOB);
        if constexpr (V == 7) asm volatile(R8(M_A4 M_A4) ::: CLOB);
        if constexpr (V == 8) asm volatile(R8(M_A5 M_A5) ::: CLOB);
        if constexpr (V == 9) asm volatile(R8(M_A6 M_A6) ::: CLOB);
        if constexpr (V == 10) asm volatile(R8(M_MIX M_MIX) ::: CLOB, "s4");
        if constexpr (V == 11) asm volatile(R8(M_MIX5 M_MIX5) ::: CLOB, "s4");
        if constexpr (V == 12) asm volatile(R8(MFMA MFMA) ::: CLOB);
        if constexpr (V == 13) asm volatile(R8(PKFMA8 PKFMA8) ::: CLOB);
        if constexpr (V == 14) asm volatile(R8(PKFMAB8 PKFMAB8) ::: CLOB);
        if constexpr (V == 15) asm volatile(R8(PKADD8 PKADD8) ::: CLOB);
        if constexpr (V == 16) asm volatile(R8(PKMUL8 PKMUL8) ::: CLOB);
        if constexpr (V == 19) asm volatile(R8(DOT16) ::: CLOB);
        if constexpr (V == 20) asm volatile(R8(DOTC16) ::: CLOB);
        if constexpr (V == 21) asm volatile(R8(M_DOT M_DOT) ::: CLOB, "s4");
        if constexpr (V == 17) asm volatile(R8(M_PK M_PK) ::: CLOB);
        if constexpr (V == 18) asm volatile(R8(M_PK6 M_PK6) ::: CLOB);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int V>
void run(const char* name, int per_iter, unsigned long long* cyc) {
    for (int threads : {256, 512}) {
        const int n = 500;
        for (int rep = 0; rep < 2; ++rep) { probe<V><<<256, threads>>>(cyc, n); hipDeviceSynchronize(); }
        double s = 0; for (int b = 0; b < 256; ++b) s += (double)cyc[b];
        printf("%-58s %d wave(s)/SIMD: %7.2f cycles per %s\n", name, threads / 256, s / 256 / n / per_iter, V >= 6 ? "MFMA gap" : "instruction");
    }
}
int main() {
    unsigned long long* cyc;
    hipMallocManaged(&cyc, 256 * 8);
    run<0>("v_fma_f32 (3 VGPR sources)", 128, cyc);
    run<1>("v_add_f32", 128, cyc);
    run<2>("v_exp_f32", 128, cyc);
    run<3>("v_fma_f32 v, v, s, v", 128, cyc);
    run<4>("v_cvt_pk_bf16_f32", 128, cyc);
    run<5>("v_max3_f32", 128, cyc);
    run<13>("v_pk_fma_f32", 128, cyc);
    run<14>("v_pk_fma_f32 op_sel_hi:[1,0,0] (broadcast low of src1)", 128, cyc);
    run<15>("v_pk_add_f32", 128, cyc);
    run<16>("v_pk_mul_f32", 128, cyc);
    run<19>("v_dot2_f32_bf16", 128, cyc);
    run<20>("v_dot2c_f32_bf16", 128, cyc);
    run<12>("mfma only", 16, cyc);
    run<21>("mfma + fma(s) exp dot2 cvt dot2c", 16, cyc);
    run<17>("mfma + pk_fma exp exp pk_add cvt (2 scores)", 16, cyc);
    run<18>("mfma + pk_fma exp exp pk_add cvt max3", 16, cyc);
    run<6>("mfma + 4 v_fma (3 VGPR)", 16, cyc);
    run<7>("mfma + 4 v_add", 16, cyc);
    run<8>("mfma + 5 v_add", 16, cyc);
    run<9>("mfma + 6 v_add", 16, cyc);
    run<10>("mfma + fma(s) exp add cvt", 16, cyc);
    run<11>("mfma + fma(s) exp add cvt max3", 16, cyc);
    return 0;
}
