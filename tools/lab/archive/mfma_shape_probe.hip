// Bare bf16 MFMA loops on pseudo-random register operands, one wave per SIMD, every CU busy: 32x32x16 vs 16x16x32 at the same
// FLOP count.  Reports TFLOP/s by wall clock over many launches (the board's power management decides the clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline bf16x8 rnd8(unsigned s) {
    bf16x8 v;
    for (int j = 0; j < 8; ++j) { s = s * 1664525u + 1013904223u; v[j] = (__bf16)(((int)(s >> 9) % 2001 - 1000) * 1e-3f); }
    return v;
}
template <int SHAPE>
__global__ __launch_bounds__(256) void probe(float* out, int iters, int zero) {
    const unsigned lane = threadIdx.x + blockIdx.x * 256;
    bf16x8 a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = rnd8(lane * 16 + i); b[i] = rnd8(lane * 16 + 8 + i); }
    if (zero) for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) { a[i][j] = (__bf16)0.0f; b[i][j] = (__bf16)0.0f; }
    float acc_out = 0.0f;
    if constexpr (SHAPE == 32) {
        f32x16 c[4] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int k = 0; k < 4; ++k) c[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + k) & 7], b[(i + 2 * k) & 7], c[k], 0, 0, 0);
            }
        }
        for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc_out += c[k][r];
    } else {
        f32x4 c[16] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int k = 0; k < 8; ++k) c[(2 * i + k) & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + k) & 7], b[(i + 3 * k) & 7], c[(2 * i + k) & 15], 0, 0, 0);
            }
        }
        for (int k = 0; k < 16; ++k) for (int r = 0; r < 4; ++r) acc_out += c[k][r];
    }
    if (acc_out == 12345.678f) out[0] = acc_out;
}
template <int SHAPE>
void run(const char* name, int zero, float* out) {
    const int iters = 4000;
    // FLOP per wave per iteration: 32-shape: 32 MFMAs x 32768; 16-shape: 64 MFMAs x 16384 -- equal
    const double flop = 256.0 * 4 * iters * 32 * 32768.0;
    for (int i = 0; i < 20; ++i) probe<SHAPE><<<256, 256>>>(out, iters, zero);
    hipDeviceSynchronize();
    const int n = 300;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) probe<SHAPE><<<256, 256>>>(out, iters, zero);
    hipDeviceSynchronize();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%-28s %s operands: %.3f ms per launch, %.1f TFLOP/s\n", name, zero ? "zero  " : "random", s / n * 1e3, flop * n / s / 1e12);
}
int main() {
    float* out; hipMalloc(&out, 64);
    for (int rep = 0; rep < 2; ++rep) {
        run<32>("v_mfma_f32_32x32x16_bf16", 0, out);
        run<16>("v_mfma_f32_16x16x32_bf16", 0, out);
    }
    run<32>("v_mfma_f32_32x32x16_bf16", 1, out);
    run<16>("v_mfma_f32_16x16x32_bf16", 1, out);
    return 0;
}
