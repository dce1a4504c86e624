// v_mfma_f32_4x4x4_16b_bf16 as a lane-local row sum: operand map check (which operand order gives "sum of the lane's own four
// values") and its price beside v_mfma_f32_32x32x16_bf16 + softmax fillers, one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void sem(const unsigned short* p, float* outA, float* outB) {
    const int l = threadIdx.x;
    s4 pv, ones;
    for (int i = 0; i < 4; ++i) { pv[i] = (short)p[l * 4 + i]; ones[i] = 0x3f80; }
    f4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
    asm volatile("s_nop 4\n v_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, %0\n s_nop 7\n s_nop 7" : "+v"(a) : "v"(ones), "v"(pv));   // A = ones, B = P
    asm volatile("s_nop 4\n v_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, %0\n s_nop 7\n s_nop 7" : "+v"(b) : "v"(pv), "v"(ones));   // A = P, B = ones
    for (int i = 0; i < 4; ++i) { outA[l * 4 + i] = a[i]; outB[l * 4 + i] = b[i]; }
}
#define R8(x) x x x x x x x x
#define CLOB "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","a0","a15","a16","a31"
#define MFMA "v_mfma_f32_32x32x16_bf16 a[0:15], v[116:119], v[120:123], a[0:15]\n"
#define MINI "v_mfma_f32_4x4x4_16b_bf16 v[108:111], v[116:117], v[120:121], v[108:111]\n"
#define MINI2 "v_mfma_f32_4x4x4_16b_bf16 v[112:115], v[116:117], v[122:123], v[112:115]\n"
// old body: fma exp add cvt max3 ; msum body: fma exp cvt (+ mini every 4th gap) ; lazy body
#define G_OLD MFMA "v_fma_f32 v100, v100, s4, v117\n v_exp_f32 v101, v101\n v_add_f32 v102, v102, v118\n v_cvt_pk_bf16_f32 v103, v120, v121\n v_max3_f32 v104, v104, v122, v123\n"
#define G_LAZY MFMA "v_fma_f32 v100, v100, s4, v117\n v_exp_f32 v101, v101\n v_cvt_pk_bf16_f32 v103, v120, v121\n v_fma_f32 v105, v105, s4, v117\n"
#define G_LAZYM MFMA "v_fma_f32 v100, v100, s4, v117\n v_exp_f32 v101, v101\n v_cvt_pk_bf16_f32 v103, v120, v121\n" MINI
#define G_LAZY3 MFMA "v_fma_f32 v100, v100, s4, v117\n v_exp_f32 v101, v101\n v_cvt_pk_bf16_f32 v103, v120, v121\n"
template <int V>
__global__ void probe(unsigned long long* cyc, int n) {
    asm volatile("v_mov_b32 v116, 1.0\n v_mov_b32 v117, 0\n v_mov_b32 v118, 1.0\n v_mov_b32 v119, 0\n v_mov_b32 v120, 1.0\n v_mov_b32 v121, 0\n v_mov_b32 v122, 1.0\n v_mov_b32 v123, 0\n s_mov_b32 s4, 1.0" ::: CLOB, "s4");
    asm volatile("v_mov_b32 v100, 0.5\n v_mov_b32 v101, 0.5\n v_mov_b32 v102, 0.5\n v_mov_b32 v103, 0.5\n v_mov_b32 v104, 0.5\n v_mov_b32 v105, 0.5\n v_mov_b32 v106, 0.5\n v_mov_b32 v107, 0.5\n"
                 "v_mov_b32 v108, 0.5\n v_mov_b32 v109, 0.5\n v_mov_b32 v110, 0.5\n v_mov_b32 v111, 0.5\n v_mov_b32 v112, 0.5\n v_mov_b32 v113, 0.5\n v_mov_b32 v114, 0.5\n v_mov_b32 v115, 0.5" ::: CLOB);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        if constexpr (V == 0) asm volatile(R8(MFMA MFMA MFMA MFMA) ::: CLOB);                       // 32 gaps
        if constexpr (V == 1) asm volatile(R8(G_OLD G_OLD G_OLD G_OLD) ::: CLOB, "s4");
        if constexpr (V == 2) asm volatile(R8(G_LAZY G_LAZY G_LAZY G_LAZY) ::: CLOB, "s4");
        if constexpr (V == 3) asm volatile(R8(G_LAZYM G_LAZY3 G_LAZY3 G_LAZY3) ::: CLOB, "s4");     // one mini per 4 gaps (the kernel's rate)
        if constexpr (V == 4) asm volatile(R8(G_LAZYM G_LAZYM G_LAZYM G_LAZYM) ::: CLOB, "s4");     // one mini per gap
        if constexpr (V == 5) asm volatile(R8(MFMA MINI MFMA MINI2 MFMA MINI MFMA MINI2) ::: CLOB);  // bare: big + mini alternating
        if constexpr (V == 6) asm volatile(R8(MINI MINI2 MINI MINI2) ::: CLOB);                      // minis alone (two accumulators)
        if constexpr (V == 7) asm volatile(R8(MINI MINI MINI MINI) ::: CLOB);                        // minis alone, one accumulator chain
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int V> void run(const char* name, int per_iter) {
    unsigned long long* d; hipMalloc(&d, 8 * 1024);
    const int n = 2000;
    hipLaunchKernelGGL(probe<V>, dim3(1024), dim3(256), 0, 0, d, n); hipDeviceSynchronize();
    hipLaunchKernelGGL(probe<V>, dim3(1024), dim3(256), 0, 0, d, n); hipDeviceSynchronize();
    std::vector<unsigned long long> h(1024); hipMemcpy(h.data(), d, 8 * 1024, hipMemcpyDeviceToHost);
    double s = 0; for (auto x : h) s += x; s /= 1024;
    printf("%-46s %.1f cycles per unit (%d units per iteration)\n", name, s / n / per_iter, per_iter);
    hipFree(d);
}
int main() {
    std::vector<unsigned short> hp(256); std::vector<float> ref(64);
    for (int l = 0; l < 64; ++l) { float s = 0; for (int i = 0; i < 4; ++i) { int v = (l * 7 + i * 3) % 13 + 1; float f = (float)v; unsigned u; memcpy(&u, &f, 4); hp[l * 4 + i] = u >> 16; s += f; } ref[l] = s; }
    unsigned short* dp; float *da, *db; hipMalloc(&dp, 512); hipMalloc(&da, 1024); hipMalloc(&db, 1024);
    hipMemcpy(dp, hp.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(sem, dim3(1), dim3(64), 0, 0, dp, da, db); hipDeviceSynchronize();
    std::vector<float> ha(256), hb(256); hipMemcpy(ha.data(), da, 1024, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), db, 1024, hipMemcpyDeviceToHost);
    int okA = 0, okB = 0; for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) { okA += ha[l * 4 + i] == ref[l]; okB += hb[l * 4 + i] == ref[l]; }
    printf("A=ones,B=P: %d/256 registers equal the lane's own sum; A=P,B=ones: %d/256\n", okA, okB);
    printf("lane 5: ref %.0f  A-order %.0f %.0f %.0f %.0f  B-order %.0f %.0f %.0f %.0f\n", ref[5], ha[20], ha[21], ha[22], ha[23], hb[20], hb[21], hb[22], hb[23]);
    run<0>("bare 32x32x16", 32); run<1>("gap = MFMA + fma exp add cvt max3 (old)", 32); run<2>("gap = MFMA + fma exp cvt fma (lazy, no mini)", 32);
    run<3>("gap = MFMA + fma exp cvt, mini every 4th", 32); run<4>("gap = MFMA + fma exp cvt + mini each", 32); run<5>("bare big + mini alternating", 32);
    run<6>("mini alone, two accumulators", 32); run<7>("mini alone, one chain", 32);
    return 0;
}
