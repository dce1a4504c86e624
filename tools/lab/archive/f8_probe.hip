// Probe of v_mfma_f32_32x32x64_f8f6f4 (fp8 e4m3 operands) on gfx950: (1) which A byte positions multiply which B byte
// positions (k-slot correspondence), (2) row/column lane maps, (3) cycles per instruction vs v_mfma_f32_32x32x16_bf16 and
// v_mfma_i32_32x32x32_i8, (4) v_cvt_pk_fp8_f32 semantics (rounding, saturation, byte placement).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/lab_bin/f8_probe tools/lab/f8_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <cmath>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

__device__ __forceinline__ v16f mfma_f8(v8i a, v8i b, v16f c) {
    // cbsz = 0 (A fp8 e4m3), blgp = 0 (B fp8 e4m3); scales: E8M0 127 = 1.0 in byte 0
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

// (1) match[pa][pb] = 1 iff A byte position pa (= 32 * lane_half + byte) multiplies B byte position pb
__global__ void probe_match(unsigned char* match) {
    const int lane = threadIdx.x, h = lane >> 5;
    for (int pa = 0; pa < 64; ++pa)
        for (int pb = 0; pb < 64; ++pb) {
            v8i a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
            if (h == (pa >> 5)) a[(pa & 31) >> 2] = 0x38 << (8 * (pa & 3));  // 1.0 in e4m3
            if (h == (pb >> 5)) b[(pb & 31) >> 2] = 0x38 << (8 * (pb & 3));
            v16f c = {0};
            c = mfma_f8(a, b, c);
            if (lane == 0) match[pa * 64 + pb] = c[0] != 0.0f;
        }
}

// (2) full product with given per-lane bytes; host checks against its layout hypothesis
__global__ void probe_full(const unsigned char* abytes, const unsigned char* bbytes, float* cout) {
    const int lane = threadIdx.x;
    v8i a, b;
    memcpy(&a, abytes + lane * 32, 32);
    memcpy(&b, bbytes + lane * 32, 32);
    v16f c = {0};
    c = mfma_f8(a, b, c);
    for (int r = 0; r < 16; ++r) cout[lane * 16 + r] = c[r];
}

// (3) timing: N dependent-free MFMAs per wave, one wave per SIMD (256 threads), clock stamps
template <int KIND>
__global__ __launch_bounds__(256, 1) void probe_rate(unsigned long long* cyc, float* sink, int n) {
    v8i a8 = {(int)threadIdx.x, 1, 2, 3, 4, 5, 6, 7}, b8 = {7, 6, 5, 4, 3, 2, 1, (int)threadIdx.x};
    v16f c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    v16i i0 = {0}, i1 = {0};
    v8bf ab, bb;
    for (int j = 0; j < 8; ++j) { ab[j] = (__bf16)(float)(threadIdx.x + j); bb[j] = (__bf16)(float)(j * 3 - (int)threadIdx.x); }
    v4i ai = {1, 2, 3, (int)threadIdx.x}, bi = {4, 5, 6, (int)threadIdx.x};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        if constexpr (KIND == 0) {
            c0 = mfma_f8(a8, b8, c0); c1 = mfma_f8(a8, b8, c1); c2 = mfma_f8(a8, b8, c2); c3 = mfma_f8(a8, b8, c3);
        } else if constexpr (KIND == 1) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c3, 0, 0, 0);
        } else if constexpr (KIND == 2) {
            i0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ai, bi, i0, 0, 0, 0); i1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ai, bi, i1, 0, 0, 0);
            i0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ai, bi, i0, 0, 0, 0); i1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ai, bi, i1, 0, 0, 0);
        } else {  // KIND 3: the non-scaled f8f6f4 form
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c0, 0, 0, 0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c1, 0, 0, 0, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c2, 0, 0, 0, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c3, 0, 0, 0, 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r] + (float)(i0[r] + i1[r]);
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// (4) cvt semantics
__global__ void probe_cvt(const float* in, unsigned* out, int n) {
    const int i = threadIdx.x;
    if (i < n) {
        unsigned w = 0xAAAAAAAAu;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(in[2 * i], in[2 * i + 1], w, false);  // low word
        out[2 * i] = w;
        unsigned w2 = 0xAAAAAAAAu;
        w2 = __builtin_amdgcn_cvt_pk_fp8_f32(in[2 * i], in[2 * i + 1], w2, true);  // high word
        out[2 * i + 1] = w2;
    }
}

static float e4m3_to_float(unsigned char b) {
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v = e == 0 ? std::ldexp((float)m, -9) : (e == 15 && m == 7) ? NAN : std::ldexp(1.0f + m / 8.0f, e - 7);
    return s ? -v : v;
}

int main() {
    unsigned char* match;
    hipMallocManaged(&match, 4096);
    probe_match<<<1, 64>>>(match);
    hipDeviceSynchronize();
    int ident = 1, nmatch = 0;
    for (int a = 0; a < 64; ++a)
        for (int b = 0; b < 64; ++b) {
            nmatch += match[a * 64 + b];
            if (match[a * 64 + b] != (a == b)) ident = 0;
        }
    printf("match matrix: %d ones, identity=%d\n", nmatch, ident);
    if (!ident)
        for (int a = 0; a < 64; ++a) {
            printf("A pos %2d ->", a);
            for (int b = 0; b < 64; ++b) if (match[a * 64 + b]) printf(" %d", b);
            printf("\n");
        }
    // full product under the hypothesis: A[i = lane%32][k = 32*(lane/32) + byte], B[k][j = lane%32] same, C standard 32x32
    std::vector<unsigned char> A(64 * 32), Bm(64 * 32);
    float Af[32][64], Bf[64][32];
    srand(1);
    for (int l = 0; l < 64; ++l)
        for (int p = 0; p < 32; ++p) {
            // small integer-valued fp8: values in {-4..4} exactly representable
            int va = rand() % 9 - 4, vb = rand() % 9 - 4;
            auto enc = [](int v) -> unsigned char { unsigned char s = v < 0 ? 0x80 : 0; int a = abs(v); unsigned char code[5] = {0x00, 0x38, 0x40, 0x44, 0x48}; return s | code[a]; };
            A[l * 32 + p] = enc(va); Bm[l * 32 + p] = enc(vb);
            Af[l % 32][32 * (l / 32) + p] = (float)va; Bf[32 * (l / 32) + p][l % 32] = (float)vb;
        }
    unsigned char *dA, *dB; float* dC;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 64 * 16 * 4);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, Bm.data(), 2048, hipMemcpyHostToDevice);
    probe_full<<<1, 64>>>(dA, dB, dC);
    std::vector<float> C(1024);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) {
            const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
            float ref = 0; for (int k = 0; k < 64; ++k) ref += Af[row][k] * Bf[k][col];
            if (ref != C[l * 16 + r]) ++bad;
        }
    printf("full product vs hypothesis (k = 32*half + byte, rows/cols = lane%%32, standard C map): %d mismatches of 1024\n", bad);
    // rates
    unsigned long long* cyc; float* sink;
    hipMallocManaged(&cyc, 256 * 8); hipMalloc(&sink, 256 * 256 * 4);
    const int n = 20000;
    const char* names[4] = {"scale_f32_32x32x64_f8f6f4 (fp8, scale 1.0)", "f32_32x32x16_bf16", "i32_32x32x32_i8", "scale_f32_32x32x64_f8f6f4 (scale bytes 0)"};
    for (int kind = 0; kind < 4; ++kind) {
        for (int rep = 0; rep < 2; ++rep) {
            if (kind == 0) probe_rate<0><<<256, 256>>>(cyc, sink, n);
            if (kind == 1) probe_rate<1><<<256, 256>>>(cyc, sink, n);
            if (kind == 2) probe_rate<2><<<256, 256>>>(cyc, sink, n);
            if (kind == 3) probe_rate<3><<<256, 256>>>(cyc, sink, n);
            hipDeviceSynchronize();
        }
        double s = 0; for (int b = 0; b < 256; ++b) s += (double)cyc[b];
        printf("%-48s %.2f cycles per MFMA (all 256 CUs busy)\n", names[kind], s / 256 / (4.0 * n));
    }
    // cvt
    float hin[32] = {0.0f, 1.0f, 0.3f, 0.33f, 1.0625f, 1.1875f, 447.0f, 448.0f, 449.0f, 500.0f, 1e6f, -3.0f, 0.001f, 0.002f, 1.0f / 512, 1.5f / 512,
                     0.0078125f, 0.01171875f, 17.0f, 18.0f, 19.0f, 20.0f, 0.0625f, 0.07f, 63.9f, 64.0f, 3.25f, 3.75f, 2.125f, 2.375f, 465.0f, INFINITY};
    float* din; unsigned* dout;
    hipMalloc(&din, 128); hipMalloc(&dout, 128);
    hipMemcpy(din, hin, 128, hipMemcpyHostToDevice);
    probe_cvt<<<1, 64>>>(din, dout, 16);
    unsigned hout[32];
    hipMemcpy(hout, dout, 128, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i) {
        const unsigned lo = hout[2 * i], hi = hout[2 * i + 1];
        printf("cvt_pk_fp8(%g, %g): low-word form %08x -> (%g, %g); high-word form %08x\n", hin[2 * i], hin[2 * i + 1], lo,
               e4m3_to_float(lo & 0xff), e4m3_to_float((lo >> 8) & 0xff), hi);
    }
    return 0;
}
