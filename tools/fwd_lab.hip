// tools/fwd_lab.hip -- standalone tuning/ablation driver for the bf16 D=128 forward kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DUMFA_ABL_...] -I universal-metal-flash-attention_amd/csrc \
//         tools/fwd_lab.hip -o gpurun_out/fwd_lab && gpurun_out/fwd_lab H [S] [reps]
// Not part of the product; results of ablation builds are wrong by construction (only timing matters).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

#include "fa_fwd_16_kernel.h"

using namespace umfa;

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

int main(int argc, char** argv) {
    const uint32_t H = argc > 1 ? atoi(argv[1]) : 24, S = argc > 2 ? atoi(argv[2]) : 4096, D = 128, B = 1;
    const int reps = argc > 3 ? atoi(argv[3]) : 30;
    const size_t n = (size_t)B * H * S * D;
    std::vector<uint16_t> h(n);
    uint64_t rng = 12345;
    auto fill = [&]() {
        for (size_t i = 0; i < n; ++i) {
            rng = rng * 6364136223846793005ull + 1442695040888963407ull;
            // sum of 4 uniforms ~ roughly normal, scaled to unit variance
            float u = 0;
            for (int k = 0; k < 4; ++k) u += (float)((rng >> (16 * k)) & 0xffff) / 65536.0f - 0.5f;
            float x = u * 1.732f;
            uint32_t bits;
            memcpy(&bits, &x, 4);
            h[i] = (uint16_t)((bits + 0x7fff + ((bits >> 16) & 1)) >> 16);
        }
    };
    void *q, *k, *v, *o;
    CK(hipMalloc(&q, n * 2)); CK(hipMalloc(&k, n * 2)); CK(hipMalloc(&v, n * 2)); CK(hipMalloc(&o, n * 2));
    fill(); CK(hipMemcpy(q, h.data(), n * 2, hipMemcpyHostToDevice));
    fill(); CK(hipMemcpy(k, h.data(), n * 2, hipMemcpyHostToDevice));
    fill(); CK(hipMemcpy(v, h.data(), n * 2, hipMemcpyHostToDevice));
    FwdParams p;
    memset(&p, 0, sizeof(p));
    p.q = q; p.k = k; p.v = v; p.o = o;
    p.B = B; p.H = H; p.Sq = S; p.Skv = S; p.D = D;
    for (auto* s : {p.qs, p.ks, p.vs}) { s[0] = (int64_t)H * S * D; s[1] = (int64_t)S * D; s[2] = D; s[3] = 1; }
    p.os[0] = D; p.os[1] = 1;
    p.scale = 0.08838834764831845f;
    p.in_prec = P_BF16; p.out_prec = P_BF16;
    const uint32_t items = ((S + 127) / 128) * B * H;
    p.n_full = items; p.nsplit = 1;
#ifdef UMFA_LAB_STAMPS
    unsigned long long* dbg;
    CK(hipMalloc(&dbg, (size_t)items * 64));
    p.part_buf = (float*)dbg;
#endif
    #ifndef UMFA_LAB_BN
#define UMFA_LAB_BN 64
#endif
#ifdef UMFA_LAB_DMA
    auto kfn = fa_fwd16_kernel<__bf16, 128, false, false, __bf16, true, UMFA_LAB_BN>;
#else
    auto kfn = fa_fwd16_kernel<__bf16, 128, false, false, __bf16, false, UMFA_LAB_BN>;
#endif
    const size_t lds = 4 * UMFA_LAB_BN * 128 * 2;
    CK(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kfn, dim3(items), dim3(256), lds, 0, p);
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kfn, dim3(items), dim3(256), lds, 0, p);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        ts.push_back(ms);
    }
#ifdef UMFA_LAB_STAMPS
    {
        std::vector<unsigned long long> hd((size_t)items * 8);
        CK(hipMemcpy(hd.data(), dbg, hd.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull;
        for (uint32_t i = 0; i < items; ++i) t0 = std::min(t0, hd[i * 8]);
        // real-time clock = 100 MHz -> 10 ns ticks
        auto us = [&](unsigned long long t) { return (double)(t - t0) * 0.01; };
        double pro = 0, loop = 0, epi = 0, clk = 0, last_end = 0;
        printf("block  xcc   start    prologue   loop      epilogue  end   (us)\n");
        for (uint32_t i = 0; i < items; ++i) {
            const unsigned long long* d = &hd[i * 8];
            pro += us(d[1]) - us(d[0]); loop += us(d[2]) - us(d[1]); epi += us(d[3]) - us(d[2]);
            clk += (double)(d[5] - d[4]) / ((double)(d[2] - d[0]) * 10.0);  // cycles per ns = GHz
            last_end = std::max(last_end, us(d[3]));
            if (i < 4 || (i % 64) == 0 || i + 4 >= items)
                printf("%5u  %3llu  %8.2f  %8.2f  %8.2f  %8.2f  %8.2f\n", i, d[6], us(d[0]), us(d[1]) - us(d[0]),
                       us(d[2]) - us(d[1]), us(d[3]) - us(d[2]), us(d[3]));
        }
        printf("mean prologue %.2f us, loop %.2f us, epilogue %.2f us; last end %.2f us; mean in-kernel clock %.3f GHz\n",
               pro / items, loop / items, epi / items, last_end, clk / items);
    }
#endif
    std::sort(ts.begin(), ts.end());
    const double fl = 4.0 * B * H * (double)S * S * D;
    printf("H=%u S=%u wgs=%u  median %.1f us  min %.1f us  %.1f TFLOP/s (median)\n", H, S, items, ts[ts.size() / 2] * 1e3,
           ts[0] * 1e3, fl / (ts[ts.size() / 2] * 1e-3) / 1e12);
    return 0;
}
