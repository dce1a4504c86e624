// tools/fwd_lab.hip -- standalone tuning/ablation driver for the bf16 D=128 forward kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DUMFA_ABL_...] -I universal-metal-flash-attention_amd/csrc \
//         tools/fwd_lab.hip -o gpurun_out/fwd_lab && gpurun_out/fwd_lab H [S] [reps]
// Not part of the product; results of ablation builds are wrong by construction (only timing matters).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
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
#ifndef UMFA_LAB_D
#define UMFA_LAB_D 128   /* -DUMFA_LAB_D=64 -DUMFA_LAB_CAUSAL=1 -DUMFA_LAB_DMA [-DUMFA_LAB_PV16=1]: BASELINE config 2's kernel (H 16, S 1024, reps, B 4) */
#endif
#ifndef UMFA_LAB_CAUSAL
#define UMFA_LAB_CAUSAL 0
#endif
#ifndef UMFA_LAB_KS
#define UMFA_LAB_KS 1   /* 2: the key-split form (head_dim 64, LDS-DMA) */
#endif
#ifndef UMFA_LAB_PIPE
#define UMFA_LAB_PIPE 0   /* 1: the software-pipelined loop (head_dim 64, LDS-DMA) */
#endif
#ifndef UMFA_LAB_PV16
#define UMFA_LAB_PV16 0
#endif
    const uint32_t H = argc > 1 ? atoi(argv[1]) : 24, S = argc > 2 ? atoi(argv[2]) : 4096, D = UMFA_LAB_D, B = argc > 4 ? atoi(argv[4]) : 1;
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
    p.scale = 1.0f / sqrtf((float)D);
    p.causal = UMFA_LAB_CAUSAL;
    p.pv16 = UMFA_LAB_PV16;
    p.in_prec = P_BF16; p.out_prec = P_BF16;
    const uint32_t items = ((S + 127) / 128) * B * H;
    p.n_full = items; p.nsplit = 1;
#ifndef UMFA_LAB_CBAL
#define UMFA_LAB_CBAL 0   /* 1: balanced causal pairs (argv[5] = cbal_delta) */
#endif
    const size_t cb_bytes = UMFA_LAB_CBAL ? (size_t)(items / 2) * 4 * (4 * (UMFA_LAB_D / 32) + 1) * 1024 : 0;
    unsigned long long* dbg = nullptr;
    if (UMFA_LAB_CBAL) {
        char* blk;
        CK(hipMalloc(&blk, cb_bytes + (size_t)items * 160));
        CK(hipMemset(blk, 0, cb_bytes + (size_t)items * 160));
        p.part_buf = (float*)blk;
        uint32_t* cnt;
        CK(hipMalloc(&cnt, (size_t)items * 4));
        CK(hipMemset(cnt, 0, (size_t)items * 4));
        p.part_cnt = cnt;
        p.cbal = 1;
        p.cbal_delta = argc > 5 ? atoi(argv[5]) : 1;
        dbg = (unsigned long long*)(blk + cb_bytes);
    }
#ifdef UMFA_LAB_STAMPS
    if (!UMFA_LAB_CBAL) {
        CK(hipMalloc(&dbg, (size_t)items * 160));  // [items][8] stamps, then [items][6] loop buckets (UMFA_LAB_LOOP_STAMPS) / from [items][16] on: [items][4] CBAL stamps
        p.part_buf = (float*)dbg;
    }
#endif
    #ifndef UMFA_LAB_BN
#define UMFA_LAB_BN 64
#endif
#ifdef UMFA_LAB_DMA
    auto kfn = fa_fwd16_kernel<__bf16, UMFA_LAB_D, UMFA_LAB_CAUSAL != 0, false, __bf16, true, UMFA_LAB_BN, UMFA_LAB_PV16, UMFA_LAB_KS, UMFA_LAB_PIPE, UMFA_LAB_CBAL != 0>;
#else
    auto kfn = fa_fwd16_kernel<__bf16, UMFA_LAB_D, UMFA_LAB_CAUSAL != 0, false, __bf16, false, UMFA_LAB_BN, UMFA_LAB_PV16>;
#endif
#ifndef UMFA_LAB_NS
#define UMFA_LAB_NS (UMFA_LAB_KS == 2 ? 4 : 2)   /* ring depth of the LDS-DMA staging */
#endif
    size_t lds = 2 * UMFA_LAB_NS * UMFA_LAB_BN * UMFA_LAB_D * 2;
    if (UMFA_LAB_KS == 2 && lds < 4 * (16 * (UMFA_LAB_D / 32) + 2) * 256) lds = 4 * (16 * (UMFA_LAB_D / 32) + 2) * 256;
    CK(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kfn, dim3(items), dim3(256 * UMFA_LAB_KS), lds, 0, p);
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kfn, dim3(items), dim3(256 * UMFA_LAB_KS), lds, 0, p);
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
#ifdef UMFA_LAB_LOOP_STAMPS
        {
            std::vector<unsigned long long> lp((size_t)items * 6);
            CK(hipMemcpy(lp.data(), dbg + (size_t)items * 8, lp.size() * 8, hipMemcpyDeviceToHost));
            // the longest workgroups (q-block nqb - 1) and the mean: cycles of wave 0 per bucket
            double mean[6] = {0, 0, 0, 0, 0, 0};
            uint32_t longest = 0;
            for (uint32_t i = 0; i < items; ++i) {
                for (int j = 0; j < 6; ++j) mean[j] += (double)lp[i * 6 + j] / items;
                if (hd[i * 8 + 2] - hd[i * 8 + 1] > hd[longest * 8 + 2] - hd[longest * 8 + 1]) longest = i;
            }
            printf("loop cycles of wave 0 (request issue | compute issue | next-tile wait | barrier | compute: to last QK MFMA | softmax): mean %.0f %.0f %.0f %.0f %.0f %.0f; "
                   "longest workgroup (%u, loop %.2f us) %llu %llu %llu %llu %llu %llu\n",
                   mean[0], mean[1], mean[2], mean[3], mean[4], mean[5], longest, us(hd[longest * 8 + 2]) - us(hd[longest * 8 + 1]), lp[longest * 6], lp[longest * 6 + 1],
                   lp[longest * 6 + 2], lp[longest * 6 + 3], lp[longest * 6 + 4], lp[longest * 6 + 5]);
        }
#endif
#if UMFA_LAB_CBAL
        {
            std::vector<unsigned long long> cb((size_t)items * 4);
            CK(hipMemcpy(cb.data(), dbg + (size_t)items * 16, cb.size() * 8, hipMemcpyDeviceToHost));
            // parts B are blocks [0, items / 2), parts A the rest
            double sw = 0, sw_at = 0, wait = 0, fold = 0, a_loop_end = 0, b_end = 0, a_end = 0; uint32_t nb = 0, na = 0;
            for (uint32_t i = 0; i < items; ++i) {
                const unsigned long long* c = &cb[i * 4]; const unsigned long long* d = &hd[i * 8];
                if (i < items / 2) { if (c[0]) { sw += (double)(c[1] - c[0]) * 0.01; sw_at += us(c[0]); ++nb; } b_end += us(d[3]); }
                else { if (c[2]) { wait += (double)(c[2] - d[2]) * 0.01; fold += (double)(c[3] - c[2]) * 0.01; ++na; } a_loop_end += us(d[2]); a_end += us(d[3]); }
            }
            printf("CBAL: parts B: switch at %.2f us, takes %.2f us (%u), end %.2f | parts A: loop ends %.2f, flag wait %.2f us, payload + fold %.2f us (%u), end %.2f\n",
                   nb ? sw_at / nb : 0.0, nb ? sw / nb : 0.0, nb, b_end / (items / 2), a_loop_end / (items / 2), na ? wait / na : 0.0, na ? fold / na : 0.0, na, a_end / (items / 2));
        }
#endif
        printf("mean prologue %.2f us, loop %.2f us, epilogue %.2f us; last end %.2f us; mean in-kernel clock %.3f GHz\n",
               pro / items, loop / items, epi / items, last_end, clk / items);
        // when do workgroups start / end: deciles of the start and end stamps, and the longest workgroup
        std::vector<double> st, en, du;
        for (uint32_t i = 0; i < items; ++i) { st.push_back(us(hd[i * 8])); en.push_back(us(hd[i * 8 + 3])); du.push_back(us(hd[i * 8 + 3]) - us(hd[i * 8])); }
        std::sort(st.begin(), st.end()); std::sort(en.begin(), en.end()); std::sort(du.begin(), du.end());
        printf("start us: p0 %.2f p50 %.2f p90 %.2f p100 %.2f | end us: p10 %.2f p50 %.2f p90 %.2f p100 %.2f | duration us: p10 %.2f p50 %.2f p90 %.2f max %.2f\n",
               st[0], st[items / 2], st[items * 9 / 10], st[items - 1], en[items / 10], en[items / 2], en[items * 9 / 10], en[items - 1],
               du[items / 10], du[items / 2], du[items * 9 / 10], du[items - 1]);
    }
#endif
    std::sort(ts.begin(), ts.end());
    const double fl = 4.0 * B * H * (double)S * S * D * (UMFA_LAB_CAUSAL ? 0.5 : 1.0);
    printf("H=%u S=%u wgs=%u  median %.1f us  min %.1f us  %.1f TFLOP/s (median)\n", H, S, items, ts[ts.size() / 2] * 1e3,
           ts[0] * 1e3, fl / (ts[ts.size() / 2] * 1e-3) / 1e12);
    return 0;
}
