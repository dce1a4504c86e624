// fa_aux.hip -- the two bandwidth-bound neighbours of the attention path.
//
// rope_rotate    replaces rope_rotate_{float,half,bfloat} (MFABridge.swift:269-319): interleaved-pair rotary,
//                fp32 math, fp32 cos/sin tables [S,D] or [B,S,D] with pair-duplicated entries (only the even one
//                is read), strided BHSD source -> dense BHSD destination, negate_sin = inverse rotation.
// hadamard       replaces HadamardRotation.rotate (absent submodule; contract from MFABridge.swift:3433-3459 and
//                AGENTS.md:161-170): in-place Fast Walsh-Hadamard Transform of `num_blocks` consecutive groups
//                of `block_size` (a power of two) elements, normalised by 1/sqrt(N) so that applying it twice is
//                the identity.  Parity unpinned beyond that contract (no source, no vectors in the reference).
// Both are HBM-bound: rope reads 1 + writes 1 element (+ tables, L2-resident across heads); FWHT reads and
// writes each element once, all log2(N) butterfly stages run in LDS.
#include "fa_common.h"
#include "kernels.h"

namespace umfa {

// ---------------------------------------------------------------------------------------------- RoPE
// one thread = 8 consecutive elements (4 pairs) of one row: 16-byte load/store for 16-bit types
template <typename T>
__global__ __launch_bounds__(256) void rope_rotate_kernel(RopeParams p) {
    const uint32_t chunks_per_row = p.D / 8;
    const uint64_t total = (uint64_t)p.B * p.H * p.S * chunks_per_row;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256) {
        const uint32_t ch = (uint32_t)(i % chunks_per_row);
        const uint64_t row = i / chunks_per_row;
        const uint32_t s = (uint32_t)(row % p.S);
        const uint64_t bh = row / p.S;
        const uint32_t h = (uint32_t)(bh % p.H), b = (uint32_t)(bh / p.H);
        const T* src = (const T*)p.src + (int64_t)b * p.src_batch_stride + (int64_t)h * p.src_head_stride +
                       (int64_t)s * p.src_seq_stride + ch * 8;
        T* dst = (T*)p.dst + row * p.D + ch * 8;
        const int64_t t = (int64_t)b * p.table_batch_stride + (int64_t)s * p.D + ch * 8;
        float x[8];
        if constexpr (sizeof(T) == 2) {
            typedef T T8 __attribute__((ext_vector_type(8)));
            const T8 v = *(const T8*)src;
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = (float)v[j];
        } else {
            const f32x4 a = *(const f32x4*)src, c = *(const f32x4*)(src + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { x[j] = a[j]; x[4 + j] = c[j]; }
        }
        const f32x4 c0 = *(const f32x4*)(p.cos_table + t), c1 = *(const f32x4*)(p.cos_table + t + 4);
        const f32x4 s0 = *(const f32x4*)(p.sin_table + t), s1 = *(const f32x4*)(p.sin_table + t + 4);
        float y[8];
        rope_rotate8(x, c0, c1, s0, s1, p.negate_sin != 0, y);  // shared with the fused Q load (fa_common.h)
        if constexpr (sizeof(T) == 2) {
            typedef T T8 __attribute__((ext_vector_type(8)));
            T8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (T)y[j];
            *(T8*)dst = o;
        } else {
            *(f32x4*)dst = f32x4{y[0], y[1], y[2], y[3]};
            *(f32x4*)(dst + 4) = f32x4{y[4], y[5], y[6], y[7]};
        }
    }
}

// generic fallback: one thread per pair (head_dim or strides not multiples of 8)
template <typename T>
__global__ __launch_bounds__(256) void rope_rotate_pair_kernel(RopeParams p) {
    const uint32_t pairs = p.D / 2;
    const uint64_t total = (uint64_t)p.B * p.H * p.S * pairs;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256) {
        const uint32_t pr = (uint32_t)(i % pairs);
        const uint64_t row = i / pairs;
        const uint32_t s = (uint32_t)(row % p.S);
        const uint64_t bh = row / p.S;
        const uint32_t h = (uint32_t)(bh % p.H), b = (uint32_t)(bh / p.H);
        const T* src = (const T*)p.src + (int64_t)b * p.src_batch_stride + (int64_t)h * p.src_head_stride +
                       (int64_t)s * p.src_seq_stride + pr * 2;
        T* dst = (T*)p.dst + row * p.D + pr * 2;
        const int64_t t = (int64_t)b * p.table_batch_stride + (int64_t)s * p.D + pr * 2;
        const float c = p.cos_table[t];
        float sn = p.sin_table[t];
        if (p.negate_sin) sn = -sn;
        const float x0 = (float)src[0], x1 = (float)src[1];
        dst[0] = (T)(x0 * c - x1 * sn);
        dst[1] = (T)(x1 * c + x0 * sn);
    }
}

template <typename T>
static hipError_t launch_rope_t(const RopeParams& p, hipStream_t stream) {
    const bool vec = p.D % 8 == 0 && p.src_batch_stride % 8 == 0 && p.src_head_stride % 8 == 0 &&
                     p.src_seq_stride % 8 == 0 && p.table_batch_stride % 4 == 0 && ((uintptr_t)p.src & 15) == 0 &&
                     ((uintptr_t)p.dst & 15) == 0 && ((uintptr_t)p.cos_table & 15) == 0 && ((uintptr_t)p.sin_table & 15) == 0;
    const uint64_t work = (uint64_t)p.B * p.H * p.S * (vec ? p.D / 8 : p.D / 2);
    if (work == 0) return hipSuccess;
    const unsigned grid = (unsigned)((work + 255) / 256 < 2048 * 8 ? (work + 255) / 256 : 2048 * 8);
    if (vec) hipLaunchKernelGGL(rope_rotate_kernel<T>, dim3(grid), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(rope_rotate_pair_kernel<T>, dim3(grid), dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_rope(const RopeParams& p, int prec, hipStream_t stream) {
    if (p.D % 2) return hipErrorInvalidValue;
    if (prec == P_FP16) return launch_rope_t<_Float16>(p, stream);
    if (prec == P_BF16) return launch_rope_t<__bf16>(p, stream);
    return launch_rope_t<float>(p, stream);
}

// ---------------------------------------------------------------------------------------------- FWHT
template <typename T>
__global__ __launch_bounds__(256) void hadamard_kernel(T* data, uint32_t n, uint32_t log2n, float norm) {
    extern __shared__ __attribute__((aligned(16))) float buf[];
    T* blk = data + (size_t)blockIdx.x * n;
    for (uint32_t i = threadIdx.x; i < n; i += 256) buf[i] = (float)blk[i];
    __syncthreads();
    for (uint32_t st = 0; st < log2n; ++st) {
        const uint32_t half = 1u << st;
        for (uint32_t j = threadIdx.x; j < n / 2; j += 256) {
            const uint32_t lo = ((j >> st) << (st + 1)) | (j & (half - 1)), hi2 = lo + half;
            const float a = buf[lo], b = buf[hi2];
            buf[lo] = a + b;
            buf[hi2] = a - b;
        }
        __syncthreads();
    }
    for (uint32_t i = threadIdx.x; i < n; i += 256) blk[i] = (T)(buf[i] * norm);
}

hipError_t launch_hadamard(void* data, uint32_t block_size, uint32_t num_blocks, int prec, hipStream_t stream) {
    if (block_size == 0 || (block_size & (block_size - 1)) || block_size > 32768) return hipErrorInvalidValue;
    uint32_t lg = 0;
    while ((1u << lg) < block_size) ++lg;
    const float norm = 1.0f / sqrtf((float)block_size);
    const size_t lds = (size_t)block_size * sizeof(float);
    if (prec == P_FP32) {
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)hadamard_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(hadamard_kernel<float>, dim3(num_blocks), dim3(256), lds, stream, (float*)data, block_size, lg, norm);
    } else if (prec == P_FP16) {
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)hadamard_kernel<_Float16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(hadamard_kernel<_Float16>, dim3(num_blocks), dim3(256), lds, stream, (_Float16*)data, block_size, lg, norm);
    } else {
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)hadamard_kernel<__bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(hadamard_kernel<__bf16>, dim3(num_blocks), dim3(256), lds, stream, (__bf16*)data, block_size, lg, norm);
    }
    return hipGetLastError();
}

// ---- de-quantisation of caller-quantised operands (pre-quantised backward ABI, MFABridge.swift:1699-2163) -----------
// x = (q - zero_point) * scale with one scale per tensor, or one per block of `block_size` consecutive rows of a
// (batch, head) slab when block scales are given; INT4 = two values per byte, even index in the low nibble, stored
// value + 8 (QuantizationTests.swift:72-128).  The source may hold fewer heads than the destination (grouped K/V).
// dst16 != NULL: the result goes out as fp16 (operands of the 16-bit MFMA backward) -- since round 5 as x * 2^-e with the tensor's largest |x| in [1, 2)
// (DequantParams::amax_word / unit_amax: two launches), so nothing leaves fp16's range; without unit_amax a value outside it raises *overflow
__device__ __forceinline__ void dequant_store(const DequantParams& p, int64_t i, float x, bool& ovf, float mul, unsigned& amax) {
    if (p.amax_word) {  // first launch: the tensor's largest magnitude only
        const unsigned a = __float_as_uint(x) & 0x7fffffffu;
        amax = a > amax ? a : amax;
    } else if (p.dst16) {
        x *= mul;
        ovf |= !(fabsf(x) <= 65504.0f);
        ((_Float16*)p.dst16)[i] = (_Float16)x;
    } else {
        p.dst[i] = x;
    }
}

__global__ __launch_bounds__(256) void dequant_kernel(DequantParams p) {
    const int64_t n = (int64_t)p.B * p.H_dst * p.S * p.D;
    bool ovf = false;
    unsigned amax = 0;
    const float mul = p.unit_amax ? __uint_as_float((unsigned)(127 - unit_exponent(p.unit_amax[0])) << 23) : 1.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint32_t d = (uint32_t)(i % p.D);
        const uint32_t s = (uint32_t)((i / p.D) % p.S);
        const uint32_t h = (uint32_t)((i / ((int64_t)p.D * p.S)) % p.H_dst);
        const uint32_t b = (uint32_t)(i / ((int64_t)p.D * p.S * p.H_dst));
        const uint32_t hs = h / (p.H_dst / p.H_src);
        const int64_t slab = (int64_t)b * p.H_src + hs;
        const int64_t e = slab * p.S * p.D + (p.transposed ? (int64_t)d * p.S + s : (int64_t)s * p.D + d);
        float x;
        if (p.prec == P_INT8) {
            x = (float)((const int8_t*)p.src)[e];
        } else if (p.prec == P_INT4) {
            const uint8_t byte = ((const uint8_t*)p.src)[e >> 1];
            x = (float)(int)((e & 1) ? (byte >> 4) : (byte & 15)) - 8.0f;
        } else {
            dequant_store(p, i, load_as_float(p.src, e, p.prec), ovf, mul, amax);  // fp16 / bf16 / fp32 operands pass through
            continue;
        }
        float sc = p.scale;
        int zp = p.zero_point;
        if (p.block_size && p.block_scales) {
            const int64_t blk = slab * ((p.S + p.block_size - 1) / p.block_size) + s / p.block_size;
            sc = p.block_scales[blk];
            zp = p.block_zero_points ? p.block_zero_points[blk] : 0;
        }
        dequant_store(p, i, (x - (float)zp) * sc, ovf, mul, amax);
    }
    if (ovf && p.overflow) atomicOr(p.overflow, 1u);
    if (p.amax_word) {
        __shared__ unsigned wmax[4];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned o = (unsigned)__shfl_xor((int)amax, off, 64);
            amax = o > amax ? o : amax;
        }
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = amax;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; ++w) amax = wmax[w] > amax ? wmax[w] : amax;
            if (amax > __hip_atomic_load(p.amax_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                (void)__hip_atomic_fetch_max(p.amax_word, amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// row constants of bwd16_dkdv from (LSE, D) when the dQ kernel that normally writes them ran in another call
__global__ __launch_bounds__(256) void bwd16_rowc_kernel(const float* lse, const float* dvec, float* rowc, int64_t n, const float* d_mul) {
    const float to_dout_units = d_mul ? d_mul[0] : 1.0f;  // the caller's D is in true units, the kernels' dO (and V) are power-of-two multiples
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        rowc[i] = -lse[i] * UMFA_LOG2E;
        rowc[n + i] = -dvec[i] * to_dout_units;
    }
}

// dK / dV of grouped key/value heads: sum the per-query-head gradients of a group (fp32, fixed order), four elements per
// lane (16-byte loads; slab = Skv * D is a multiple of 8)
__global__ __launch_bounds__(256) void group_sum_kernel(const float* src, void* dst, uint32_t B, uint32_t H, uint32_t Hkv,
                                                        int64_t slab, int out_prec) {
    const uint32_t g = H / Hkv;
    const int64_t slab4 = slab / 4, n4 = (int64_t)B * Hkv * slab4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int64_t e = i % slab4, bh = i / slab4;
        const int64_t b = bh / Hkv, hk = bh % Hkv;
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        for (uint32_t j = 0; j < g; ++j) acc += ((const f32x4*)(src + (b * H + hk * g + j) * slab))[e];
        if (out_prec == P_FP16) {
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            ((h4*)dst)[i] = h4{(_Float16)acc[0], (_Float16)acc[1], (_Float16)acc[2], (_Float16)acc[3]};
        } else if (out_prec == P_BF16) {
            typedef __bf16 b4 __attribute__((ext_vector_type(4)));
            ((b4*)dst)[i] = b4{(__bf16)acc[0], (__bf16)acc[1], (__bf16)acc[2], (__bf16)acc[3]};
        } else {
            ((f32x4*)dst)[i] = acc;
        }
    }
}

__global__ __launch_bounds__(256) void group_sum_scalar_kernel(const float* __restrict__ src, void* __restrict__ dst, uint32_t B, uint32_t H, uint32_t Hkv,
                                                               int64_t slab, int out_prec) {
    const uint32_t g = H / Hkv;
    const int64_t n = (int64_t)B * Hkv * slab;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t e = i % slab, bh = i / slab;
        const int64_t b = bh / Hkv, hk = bh % Hkv;
        float acc = 0.0f;
        for (uint32_t j = 0; j < g; ++j) acc += src[(b * H + hk * g + j) * slab + e];
        if (out_prec == P_FP16) ((_Float16*)dst)[i] = (_Float16)acc;
        else if (out_prec == P_BF16) ((__bf16*)dst)[i] = (__bf16)acc;
        else ((float*)dst)[i] = acc;
    }
}

// dO of the quantised backward entries as the fp16 MFMA backward takes it: dO * 2^-e with ONE power of two e per call, chosen on the
// device so that the largest |dO| lands in [1, 2).  fp16 has five exponent bits; the gradients these entries see in bf16 / fp32 training
// are routinely 1e-5 ... 1e-9 -- as a plain cast they were fp16 subnormals or zero (dQ off by 3 % at |dO| ~ 1e-5, 35 % at 1e-7, all zeros
// at 1e-9, status 0: tools/lab/qbwd_range_probe.py), and dS = P (dP - D), rounded to fp16 inside the kernels, sank with them.  Every
// gradient is linear in dO, so the kernels give 2^e back in their epilogues (BwdParams::units): exact both ways.
// hdr: [0] largest |dO| as fp32 bits (zero on entry: the launcher's memset node), [1] 2^e, [2] 2^-e (floats, written here).
template <int PREC>
__device__ __forceinline__ void load8_as_float(const void* src, int64_t i, float (&x)[8]) {
    if constexpr (PREC == P_FP32) {
        const f32x4 lo = ((const f32x4*)src)[2 * i], hi = ((const f32x4*)src)[2 * i + 1];
#pragma unroll
        for (int j = 0; j < 4; ++j) { x[j] = lo[j]; x[4 + j] = hi[j]; }
    } else if constexpr (PREC == P_BF16) {
        const s16x8 raw = ((const s16x8*)src)[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = bf16_bits_to_float((uint16_t)raw[j]);
    } else {
        const f16x8 raw = ((const f16x8*)src)[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = (float)raw[j];
    }
}
// up to four dense tensors per launch (blockIdx.y picks one): 8-element groups, eight groups per thread in flight per sweep step
struct AmaxJob {
    const void* src[4];
    int64_t n8[4];
    uint32_t* word[4];
};
template <int PREC>
__global__ __launch_bounds__(256) void amax_dense_kernel(AmaxJob job) {
    __shared__ unsigned wmax[4];
    const void* const src = job.src[blockIdx.y];
    const int64_t n8 = job.n8[blockIdx.y];
    unsigned amax = 0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    constexpr int U = 8;  // 16-byte loads in flight per thread
    for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < n8; i0 += U * stride) {
        float x[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i0 + u * stride < n8) {
                load8_as_float<PREC>(src, i0 + u * stride, x[u]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) x[u][j] = 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned a = __float_as_uint(x[u][j]) & 0x7fffffffu;
                amax = a > amax ? a : amax;
            }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)amax, off, 64);
        amax = o > amax ? o : amax;
    }
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) amax = wmax[w] > amax ? wmax[w] : amax;
        // (same-address atomics serialise at ~50 ns each -- 1024 workgroups per tensor spent 50 us of a 61-us launch in them: few workgroups, and none
        // from a workgroup whose maximum is already covered)
        if (amax > __hip_atomic_load(job.word[blockIdx.y], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            (void)__hip_atomic_fetch_max(job.word[blockIdx.y], amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
template <int PREC>
__global__ __launch_bounds__(256) void cast_f16_unit_kernel(const void* src, _Float16* dst, int64_t n8, uint32_t* hdr) {
    const int e = unit_exponent(hdr[0]);
    const float mul = __uint_as_float((unsigned)(127 - e) << 23);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        hdr[1] = (unsigned)(127 + e) << 23;
        hdr[2] = (unsigned)(127 - e) << 23;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float x[8];
        load8_as_float<PREC>(src, i, x);
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (_Float16)(x[j] * mul);
        ((f16x8*)dst)[i] = o;
    }
}

// the largest |x| of up to four dense tensors of one precision, as fp32 bits, max-ed into *word[i] (the caller zeroes them); one launch
hipError_t launch_amax_dense_n(int count, const void* const* src, int prec, const int64_t* n, uint32_t* const* word, hipStream_t stream) {
    if (count < 1 || count > 4 || (prec != P_FP32 && prec != P_BF16 && prec != P_FP16)) return hipErrorInvalidValue;
    AmaxJob job = {};
    int64_t n8max = 0;
    for (int i = 0; i < count; ++i) {
        if (!src[i] || !word[i] || (n[i] & 7)) return hipErrorInvalidValue;
        job.src[i] = src[i]; job.n8[i] = n[i] / 8; job.word[i] = word[i];
        n8max = job.n8[i] > n8max ? job.n8[i] : n8max;
    }
    // (at most one atomic per workgroup, all on one word per tensor: few, fat workgroups -- 256 per tensor (config 4, four tensors of 33.5 MB: 44 us at 128, 41 at 256; 61 at 1024 with four loads each; four separate launches of 1024: 75), each thread eight 16-byte loads in flight)
    const int64_t want = (n8max + 8 * 256 - 1) / (8 * 256);
    const dim3 grid((unsigned)(want < 1 ? 1 : want < 256 ? want : 256), (unsigned)count);
    if (prec == P_FP32) hipLaunchKernelGGL(amax_dense_kernel<P_FP32>, grid, dim3(256), 0, stream, job);
    else if (prec == P_BF16) hipLaunchKernelGGL(amax_dense_kernel<P_BF16>, grid, dim3(256), 0, stream, job);
    else hipLaunchKernelGGL(amax_dense_kernel<P_FP16>, grid, dim3(256), 0, stream, job);
    return hipGetLastError();
}
hipError_t launch_amax_dense(const void* src, int prec, int64_t n, uint32_t* word, hipStream_t stream) {
    return launch_amax_dense_n(1, &src, prec, &n, &word, stream);
}

// BwdParams::units from the four tensors' largest magnitudes (fp32 bits): hdr[0] dO, hdr[4] Q, hdr[5] K, hdr[6] V -> floats hdr[8 ... 14]
__global__ void bwd_units_kernel(uint32_t* hdr, uint32_t* flag) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int edo = unit_exponent(hdr[0]), eq = unit_exponent(hdr[4]), ek = unit_exponent(hdr[5]), ev = unit_exponent(hdr[6]);
    float* t = (float*)(hdr + 8);
    t[0] = __builtin_amdgcn_ldexpf(1.0f, eq + ek);
    t[1] = __builtin_amdgcn_ldexpf(1.0f, edo + ev + ek);
    t[2] = __builtin_amdgcn_ldexpf(1.0f, edo + ev + eq);
    t[3] = __builtin_amdgcn_ldexpf(1.0f, edo);
    t[4] = __builtin_amdgcn_ldexpf(1.0f, -ev);
    t[5] = __builtin_amdgcn_ldexpf(1.0f, edo);
    t[6] = __builtin_amdgcn_ldexpf(1.0f, -(edo + ev));
    // The last reader of the four amax words leaves them zero for the next call (and the overflow word in front of the header, which nothing can
    // raise any more): the in-stream entry then needs no memset node in front of amax_dense_kernel's agent-scope fetch_max (StreamScratch::ensure_qhdr)
    if (flag) __hip_atomic_store(flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(hdr + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(hdr + 4, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(hdr + 5, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(hdr + 6, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
hipError_t launch_bwd_units(uint32_t* hdr, hipStream_t stream, uint32_t* flag) {
    hipLaunchKernelGGL(bwd_units_kernel, dim3(1), dim3(64), 0, stream, hdr, flag);
    return hipGetLastError();
}

hipError_t launch_cast_f16_unit(const void* src, int prec, void* dst, int64_t n, uint32_t* hdr, hipStream_t stream, bool amax_done) {
    if (!src || !dst || !hdr || (n & 7) || (prec != P_FP32 && prec != P_BF16 && prec != P_FP16)) return hipErrorInvalidValue;
    if (!amax_done) {  // (the quantised backward takes the amax of all four operands in ONE launch first: launch_amax_dense_n)
        if (hipError_t e = hipMemsetAsync(hdr, 0, 4, stream); e != hipSuccess) return e;
        if (hipError_t e = launch_amax_dense(src, prec, n, hdr, stream); e != hipSuccess) return e;
    }
    const int64_t n8 = n / 8;
    const unsigned grid = (unsigned)((n8 + 255) / 256 < 8192 ? (n8 + 255) / 256 : n8 ? 8192 : 1);
    if (prec == P_FP32) hipLaunchKernelGGL(cast_f16_unit_kernel<P_FP32>, dim3(grid), dim3(256), 0, stream, src, (_Float16*)dst, n8, hdr);
    else if (prec == P_BF16) hipLaunchKernelGGL(cast_f16_unit_kernel<P_BF16>, dim3(grid), dim3(256), 0, stream, src, (_Float16*)dst, n8, hdr);
    else hipLaunchKernelGGL(cast_f16_unit_kernel<P_FP16>, dim3(grid), dim3(256), 0, stream, src, (_Float16*)dst, n8, hdr);
    return hipGetLastError();
}

// ------------------------------------------------------------------ bool masks for the one-wave-per-SIMD kernels (fa_fwd16_w64, MASKT)
// A bool mask tensor is re-packed once per call into what a wave of that kernel consumes directly:
//   bits  [mb][mh][rb64][tile][qb 2][lane 64] u32: bit 16 kb + r of lane (ql, hi) = "row 64 rb64 + 32 qb + ql attends key
//         64 tile + 32 kb + (r & 3) + 8 (r >> 2) + 4 hi" -- i.e. bit j is the mask of score register r of score block (kb, qb)
//         of that lane (fa_common.h acc_row); keys >= Skv and rows >= Sq are 0;
//   wflag [mb][mh][rb64][tile] u8: 1 = no (row < Sq, key < Skv) element attends, 2 = every element of a whole 64 x 64 tile attends
//         (the tile runs the plain tile body), 0 = mixed (the masking body with bits);
//   list  [mb][mh][qblk][T][2] u32 + cnt [mb][mh][qblk]: the key tiles a 256-row block visits, ascending -- those in which not all
//         four of its waves are fully masked -- as { tile | class(wave 0) << 16 | class(wave 1) << 18 | ..., next listed tile | the one
//         after it << 16 } (past the end: the last tile again), so that the kernel needs nothing but its current entry (mask_list_kernel).
// One read of every distinct mask byte (broadcast dims are not expanded); the bit image is 1/8 of the mask (2 MB for [1,1,4096,4096]).
struct MaskPackArgs {
    const void* mask;
    int64_t ms[4];
    uint32_t Sq, Skv;
    uint32_t* bits;
    uint8_t* wflag;
    uint32_t Bm, Hm, nrb64, T;
    uint64_t total;  // wave-tiles = Bm Hm nrb64 T (0: nothing to pack)
    int causal;      // the launch's causal flag folded into the bits (key <= row): the mask kernels have no causal instantiation and need none --
                     // tiles above the diagonal come out "masked" and never enter a block's list
    bool vec16;      // contiguous 16-byte aligned rows, Skv % 16 == 0
    bool done;       // (launcher) the pack rode along with the V cast pass
};
// bits of the lane's word (bit 16 kb + 4 g + e = key 64 tile + 32 kb + 8 g + 4 hi + e) that a causal launch keeps: key <= row
__device__ __forceinline__ uint32_t causal_word(uint32_t row, uint32_t tile, uint32_t hi) {
    const uint32_t k0 = tile * 64;
    if (k0 + 63 <= row) return 0xffffffffu;
    if (k0 > row) return 0u;
    uint32_t w = 0;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) w |= (k0 + 32 * kb + 8 * g + 4 * hi + e <= row) ? 1u << (16 * kb + 4 * g + e) : 0u;
    return w;
}

template <bool VEC16>
__device__ __forceinline__ void mask_pack_body(const MaskPackArgs& p, const uint32_t block) {
    uint32_t* const bits = p.bits;
    uint8_t* const wflag = p.wflag;
    const uint32_t Hm = p.Hm, nrb64 = p.nrb64, T = p.T;
    __shared__ __attribute__((aligned(16))) uint8_t stage[4][64 * 80];  // per wave: a 64 x 64 byte tile, rows padded to 80 bytes (bank spread)
    const uint32_t lane = threadIdx.x & 63, ql = lane & 31, hi = lane >> 5, wv = threadIdx.x >> 6;
    const uint64_t wid = (uint64_t)block * 4 + wv;
    const uint64_t total = p.total;
    if (wid >= total) return;  // (whole waves: no barrier below, every wave works on its own LDS area)
    const uint32_t tile = (uint32_t)(wid % T);
    const uint32_t rb = (uint32_t)((wid / T) % nrb64);
    const uint32_t slab = (uint32_t)(wid / ((uint64_t)T * nrb64));
    const uint32_t hm = slab % Hm, bm = slab / Hm;
    const uint8_t* base = (const uint8_t*)p.mask + (int64_t)bm * p.ms[0] + (int64_t)hm * p.ms[1];
    bool all_open = (uint64_t)tile * 64 + 64 <= p.Skv, any_open = false;
    uint32_t w[2];
    if constexpr (VEC16) {
        // contiguous, 16-byte aligned rows with Skv % 16 == 0: the tile comes in as 16-byte loads in row order (a wave-load covers
        // 16 rows x 64 bytes, coalesced), goes through the wave's LDS area and comes out in the lane order the attention kernel wants
        uint8_t* st = stage[wv];
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const uint32_t r = 16 * ps + (lane >> 2), c = lane & 3, row = rb * 64 + r, key0 = tile * 64 + 16 * c;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (row < p.Sq && key0 < p.Skv) v = *(const u32x4*)(base + (int64_t)row * p.ms[2] + key0);
            *(u32x4*)(st + r * 80 + 16 * c) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const uint32_t r = 32 * qb + ql;
            uint32_t word = 0;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint32_t four = *(const uint32_t*)(st + r * 80 + 32 * kb + 8 * g + 4 * hi);
#pragma unroll
                    for (int e = 0; e < 4; ++e) word |= ((four >> (8 * e)) & 0xffu) ? 1u << (16 * kb + 4 * g + e) : 0u;
                }
            if (p.causal) word &= causal_word(rb * 64 + r, tile, hi);
            if (rb * 64 + r < p.Sq) {
                all_open = all_open && word == 0xffffffffu;
                any_open = any_open || word != 0;
            }
            w[qb] = word;
        }
    } else {
        const bool vec4 = p.ms[3] == 1 && ((p.ms[0] | p.ms[1] | p.ms[2]) & 3) == 0 && ((uintptr_t)p.mask & 3) == 0;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const uint32_t row = rb * 64 + 32 * qb + ql;
            uint32_t word = 0;
            if (row < p.Sq) {
                const uint8_t* rp = base + (int64_t)row * p.ms[2];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const uint32_t key0 = tile * 64 + 32 * kb + 8 * g + 4 * hi;
                        uint32_t four = 0;  // byte e = mask[row][key0 + e]
                        if (vec4 && key0 + 4 <= p.Skv) {
                            four = *(const uint32_t*)(rp + key0);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (key0 + e < p.Skv) four |= (uint32_t)rp[(int64_t)(key0 + e) * p.ms[3]] << (8 * e);
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) word |= ((four >> (8 * e)) & 0xffu) ? 1u << (16 * kb + 4 * g + e) : 0u;
                    }
                if (p.causal) word &= causal_word(row, tile, hi);
                all_open = all_open && word == 0xffffffffu;
                any_open = any_open || word != 0;
            }
            w[qb] = word;
        }
    }
    uint32_t* dst = bits + ((wid * 2) * 64 + lane);
    dst[0] = w[0];
    dst[64] = w[1];
    const bool open = __builtin_amdgcn_ballot_w64(any_open) != 0, full = __builtin_amdgcn_ballot_w64(!all_open) == 0;
    if (lane == 0) wflag[wid] = !open ? 1 : (full ? 2 : 0);
}

template <bool VEC16>
__global__ __launch_bounds__(256) void mask_pack_kernel(MaskPackArgs p) {
    mask_pack_body<VEC16>(p, blockIdx.x);
}

// V of the default bf16 forward (P V product in fp16, FwdParams::pv16): bf16 rows (any batch / head / row strides, head_dim
// contiguous) -> a dense fp16 image [B, H, S, D] of V * 2^-e, e ONE power of two per (batch, head) slab taken from the slab's
// largest |v| (it lands in [2^15, 2^16): bf16's largest significand there is 65280, fp16 ends at 65504), and 2^e left in the
// slab's header for the attention kernels, which fold it into the per-row 1 / l of their epilogue.  bf16 has fp32's exponent
// range, fp16 five bits of it: after the shift every value down to amax * 2^-30 is exact in fp16 (8-bit significands), smaller
// ones round into fp16's subnormals -- errors below amax * 2^-40, seven orders of magnitude under what the fp16 rounding of P
// costs.  So there is no input this pass turns into +-inf (round 4 did: a V value >= 65536 made the call's outputs non-finite
// and raised a status word for LATER calls) and nothing sticky.
// HBM-bound: one read + one write of V.  A workgroup owns U * 256 / (D / 8) consecutive rows of one slab (U = 16: 256 rows =
// 64 KB at head_dim 128); every thread has U independent 16-byte loads in flight and keeps them in registers across the slab's
// amax exchange -- the whole tensor is requested before the first byte is converted.
//
// The exchange (FUSED): a workgroup publishes its amax as a FLAG word of its own (write-through store: no atomic), then wave 0
// polls the slab's <= 64 flag words with one load per round until all are up; leaving, it counts itself out with the pass's one
// atomic, and the LAST workgroup to leave writes 2^e and zeroes the slab's words for the next launch (same stream: the next
// launch starts after this one ended).  (First form, measured: arrival counted with two atomics per workgroup on the slab's line,
// 64 workgroups per slab: 68 us instead of 11 -- same-line atomics from eight XCDs serialise at a few hundred ns each.)
// Safe without a co-operative launch because workgroups are dispatched in linear order (slab-major here) and a slab is at most
// 64 workgroups: the oldest unfinished slab is always resident as a whole (2048 workgroup slots; <= 8 per XCD per slab).
// Slabs of more than 64 chunks take two launches (vamax_rows_kernel, then this kernel with FUSED = false).
// Header, 512 bytes per slab: words [0, 64) flags (0x80000000 | amax bits), [64] departed, [65] 2^e as fp32, [66] amax of the
// two-launch form; all zero between launches except [65].
constexpr uint32_t VH_WORDS = 128, VH_DEPART = 64, VH_SCALE = 65, VH_AMAX = 66;
__device__ __forceinline__ int vscale_exponent(unsigned amax_bits) {
    // amax_bits: |v| as bf16 bits; inf / NaN count as the largest finite exponent (they go out as inf / NaN whatever e is)
    if (amax_bits == 0) return 0;
    int E = (int)(amax_bits >> 7);
    E = E > 254 ? 254 : E;
    const int e = (E ? E - 127 : -126) - 15;  // (bf16's largest significand times 2^15 is 65280: inside fp16)
    return e < -100 ? -100 : e;  // (2^e and 2^-e stay normal fp32 numbers; |v| < 2^-115 is then merely less well placed)
}
__device__ __forceinline__ unsigned bf16x8_amax(const unsigned (&r)[4], unsigned amax) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned a = r[j] & 0x7fff7fffu, m2 = (a & 0xffffu) > (a >> 16) ? (a & 0xffffu) : (a >> 16);
        amax = amax > m2 ? amax : m2;
    }
    return amax;
}
__device__ __forceinline__ unsigned wave_umax(unsigned x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned o2 = (unsigned)__shfl_xor((int)x, off, 64);
        x = x > o2 ? x : o2;
    }
    return x;
}
__device__ __forceinline__ unsigned block_amax(unsigned amax, unsigned* wmax) {  // (every thread gets the workgroup's value)
    amax = wave_umax(amax);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = amax;
    __syncthreads();
    unsigned m = wmax[0];
    for (int w = 1; w < 4; ++w) m = m > wmax[w] ? m : wmax[w];
    return m;
}

struct CastRowsArgs {
    const uint16_t* src;
    int64_t sb, sh, ss;
    _Float16* dst;
    uint32_t H, S, D8, chunks;
    uint32_t* hdr;
    uint32_t wait_ticks;  // of the 100 MHz s_memrealtime clock (s_memtime counts shader clocks: ~1.7 GHz here)
};
template <int U, bool FUSED>
__device__ __forceinline__ void cast_rows_body(const CastRowsArgs& a, const uint32_t block) {
    const uint16_t* __restrict__ src = a.src;
    _Float16* __restrict__ dst = a.dst;
    uint32_t* __restrict__ hdr = a.hdr;
    const int64_t sb = a.sb, sh = a.sh, ss = a.ss;
    const uint32_t H = a.H, S = a.S, D8 = a.D8, chunks = a.chunks;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __shared__ unsigned wmax[4], slab_amax;
    const uint32_t bh = block / chunks, chunk = block - bh * chunks, b = bh / H, h = bh - b * H;  // slab-major linear order
    const uint32_t rpw = 256u / D8;                       // rows one pass of the workgroup covers (D8 divides 256: head_dim 64 ... 256 x8; else see launcher)
    const uint32_t tr = threadIdx.x / D8, c = threadIdx.x - tr * D8;
    const uint32_t row0 = chunk * (U * rpw) + tr;
    const uint16_t* sp = src + (int64_t)b * sb + (int64_t)h * sh + 8 * c;
    _Float16* dp = dst + ((int64_t)bh * S) * (8 * D8) + 8 * c;
    uint32_t* const hw = hdr + VH_WORDS * (size_t)bh;
    u32x4 raw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t r = row0 + u * rpw;
        raw[u] = r < S && tr < rpw ? __builtin_nontemporal_load((const u32x4*)(sp + (int64_t)r * ss)) : u32x4{0, 0, 0, 0};
    }
    unsigned amax;
    if constexpr (FUSED) {
        amax = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned r4[4] = {raw[u][0], raw[u][1], raw[u][2], raw[u][3]};
            amax = bf16x8_amax(r4, amax);
        }
        amax = block_amax(amax, wmax);
        if (threadIdx.x < 64) {
            // relaxed agent-scope accesses only (performed past the XCDs' L2s); the exchange carries nothing but these words
            if (threadIdx.x == 0) __hip_atomic_store(hw + chunk, 0x80000000u | amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // The wait is an optimisation, not a dependency: nothing promises that the slab's other workgroups are resident (a stream
            // with a small CU mask, many streams' cast passes at once), so it is bounded -- a workgroup that is not served in time reads
            // the whole slab for its amax itself (the same number: a max), below.
            const uint64_t t_in = __builtin_amdgcn_s_memrealtime();
            unsigned f;
            bool served = true;
            for (;;) {
                f = threadIdx.x < chunks ? __hip_atomic_load(hw + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0x80000000u;
                if (__builtin_amdgcn_ballot_w64((f & 0x80000000u) == 0) == 0) break;
                if (__builtin_amdgcn_s_memrealtime() - t_in >= a.wait_ticks) { served = false; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            f = wave_umax(f & 0x7fffffffu);
            if (threadIdx.x == 0) slab_amax = served ? f : 0xffffffffu;
        }
        __syncthreads();
        if (slab_amax == 0xffffffffu) {  // (workgroup-uniform)
            for (uint32_t r = tr; r < S && tr < rpw; r += rpw) {
                const u32x4 v = *(const u32x4*)(sp + (int64_t)r * ss);
                const unsigned r4[4] = {v[0], v[1], v[2], v[3]};
                amax = bf16x8_amax(r4, amax);
            }
            __syncthreads();  // wmax is used a second time
            amax = block_amax(amax, wmax);
        } else {
            amax = slab_amax;
        }
    } else {
        amax = hw[VH_AMAX];  // vamax_rows_kernel, the launch before this one
    }
    const int e = vscale_exponent(amax);
    const float mul = __uint_as_float((unsigned)(127 - e) << 23);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t r = row0 + u * rpw;
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float lo = __uint_as_float(raw[u][j] << 16) * mul, hi = __uint_as_float(raw[u][j] & 0xffff0000u) * mul;
            asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(o[j]) : "v"(lo), "v"(hi));
        }
        if (r < S && tr < rpw) *(u32x4*)(dp + (int64_t)r * (8 * D8)) = o;
    }
    // leave: the last workgroup of the slab writes 2^e and zeroes the exchange words (every workgroup has read them by now)
    if (threadIdx.x < 64) {
        unsigned d = 0;
        if (threadIdx.x == 0) d = __hip_atomic_fetch_add(hw + VH_DEPART, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        d = (unsigned)__builtin_amdgcn_readfirstlane((int)d);
        if (d == chunks - 1) {
            if (FUSED && threadIdx.x < chunks) __hip_atomic_store(hw + threadIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (threadIdx.x == 0) {
                hw[VH_SCALE] = (unsigned)(127 + e) << 23;
                __hip_atomic_store(hw + VH_DEPART, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!FUSED) __hip_atomic_store(hw + VH_AMAX, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

template <int U, bool FUSED>
__global__ __launch_bounds__(256) void cast_rows_bf16_f16_kernel(CastRowsArgs a) {
    cast_rows_body<U, FUSED>(a, blockIdx.x);
}

// the V cast pass and the bool mask's re-pack in one launch: blocks [0, cast_blocks) cast (slab-major, first: their exchange relies on it),
// the rest pack mask tiles
template <int U, bool VEC16>
__global__ __launch_bounds__(256) void cast_rows_and_mask_pack_kernel(CastRowsArgs a, uint32_t cast_blocks, MaskPackArgs mk) {
    if (blockIdx.x < cast_blocks) cast_rows_body<U, true>(a, blockIdx.x);
    else mask_pack_body<VEC16>(mk, blockIdx.x - cast_blocks);
}

// slabs too long for the in-kernel exchange: the amax of every slab first (one more read of V; such slabs are >= 16 MB of attention work each)
template <int U>
__global__ __launch_bounds__(256) void vamax_rows_kernel(const uint16_t* __restrict__ src, int64_t sb, int64_t sh, int64_t ss, uint32_t H, uint32_t S,
                                                         uint32_t D8, uint32_t chunks, uint32_t* __restrict__ hdr) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __shared__ unsigned wmax[4];
    const uint32_t bh = blockIdx.x / chunks, chunk = blockIdx.x - bh * chunks, b = bh / H, h = bh - b * H;
    const uint32_t rpw = 256u / D8, tr = threadIdx.x / D8, c = threadIdx.x - tr * D8;
    const uint32_t row0 = chunk * (U * rpw) + tr;
    const uint16_t* sp = src + (int64_t)b * sb + (int64_t)h * sh + 8 * c;
    unsigned amax = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t r = row0 + u * rpw;
        const u32x4 v = r < S && tr < rpw ? *(const u32x4*)(sp + (int64_t)r * ss) : u32x4{0, 0, 0, 0};  // (no non-temporal hint: the cast kernel reads it again)
        const unsigned r4[4] = {v[0], v[1], v[2], v[3]};
        amax = bf16x8_amax(r4, amax);
    }
    amax = block_amax(amax, wmax);
    if (threadIdx.x == 0 && amax) (void)__hip_atomic_fetch_max(hdr + VH_WORDS * (size_t)bh + VH_AMAX, amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int U>
static hipError_t launch_cast_rows_u(const void* src, const int64_t* strides, void* dst, uint32_t B, uint32_t H, uint32_t S, uint32_t D, uint32_t* hdr,
                                     hipStream_t stream, MaskPackArgs* mk) {
    const uint32_t D8 = D / 8, rpw = 256u / D8;  // (a head_dim that does not divide 2048 leaves 256 % D8 threads idle)
    const uint64_t chunks = ((uint64_t)S + U * rpw - 1) / (U * rpw), grid = (uint64_t)B * H * chunks;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    const bool fused = chunks <= 64 && !tuning().cast_two_pass.load(std::memory_order_relaxed);
    const CastRowsArgs a = {(const uint16_t*)src, strides[0], strides[1], strides[2], (_Float16*)dst, H, S, D8, (uint32_t)chunks, hdr,
                            (uint32_t)std::min<int64_t>(std::max(tuning().cast_wait_us.load(std::memory_order_relaxed), 0), 1000000) * 100u};
    if (fused) {
        if (mk && mk->total) {
            // ... and the bool mask's re-pack for the one-wave-per-SIMD kernel in the SAME launch: its workgroups come behind the cast's
            // (whose exchange needs them dispatched first, in order), run under the cast's memory time, and the call is one launch shorter
            const unsigned g2 = (unsigned)((mk->total + 3) / 4);
            if (grid + g2 > 0x7fffffffull) return hipErrorInvalidValue;
            if (mk->vec16) hipLaunchKernelGGL((cast_rows_and_mask_pack_kernel<U, true>), dim3((unsigned)grid + g2), dim3(256), 0, stream, a, (unsigned)grid, *mk);
            else hipLaunchKernelGGL((cast_rows_and_mask_pack_kernel<U, false>), dim3((unsigned)grid + g2), dim3(256), 0, stream, a, (unsigned)grid, *mk);
            mk->done = true;
        } else {
            hipLaunchKernelGGL((cast_rows_bf16_f16_kernel<U, true>), dim3((unsigned)grid), dim3(256), 0, stream, a);
        }
    } else {
        hipLaunchKernelGGL(vamax_rows_kernel<U>, dim3((unsigned)grid), dim3(256), 0, stream, (const uint16_t*)src, strides[0], strides[1], strides[2], H, S,
                           D8, (uint32_t)chunks, hdr);
        hipLaunchKernelGGL((cast_rows_bf16_f16_kernel<U, false>), dim3((unsigned)grid), dim3(256), 0, stream, a);
    }
    return hipGetLastError();
}

static hipError_t launch_cast_rows_any(const void* src, const int64_t* strides, void* dst, uint32_t B, uint32_t H, uint32_t S, uint32_t D,
                                       uint32_t* hdr, hipStream_t stream, MaskPackArgs* mk) {
    if (!src || !dst || !hdr || (D & 7) || D > 2048 || strides[3] != 1 || (strides[0] | strides[1] | strides[2]) % 8 || ((uintptr_t)src & 15)) return hipErrorInvalidValue;
    if ((int64_t)B * H * S * D == 0) return hipSuccess;
    // 16 loads per thread (64 KB per workgroup at head_dim 128) while that still leaves a workgroup per CU, else 4
    const uint32_t rpw = 256u / (D / 8);
    const uint64_t wg16 = (uint64_t)B * H * (((uint64_t)S + 16 * rpw - 1) / (16 * rpw));
    const int lab = tuning().cast_u.load(std::memory_order_relaxed);
    const uint64_t chunks16 = ((uint64_t)S + 16 * rpw - 1) / (16 * rpw);
    // (32: slabs of 65 ... 128 chunks of 16 passes still take the one-launch form)
    if (lab == 32 || (!lab && chunks16 > 64 && chunks16 <= 128)) return launch_cast_rows_u<32>(src, strides, dst, B, H, S, D, hdr, stream, mk);
    if (lab == 4 || (!lab && wg16 < (uint64_t)device_cu_count())) return launch_cast_rows_u<4>(src, strides, dst, B, H, S, D, hdr, stream, mk);
    return launch_cast_rows_u<16>(src, strides, dst, B, H, S, D, hdr, stream, mk);
}

hipError_t launch_bwd16_rowc(const float* lse, const float* dvec, float* rowc, int64_t n, const float* d_mul, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const unsigned grid = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(bwd16_rowc_kernel, dim3(grid), dim3(256), 0, stream, lse, dvec, rowc, n, d_mul);
    return hipGetLastError();
}

hipError_t launch_dequant(const DequantParams& p, hipStream_t stream) {
    if (!p.src || (!p.dst && !p.dst16) || p.H_src == 0 || p.H_dst % p.H_src) return hipErrorInvalidValue;
    const int64_t n = (int64_t)p.B * p.H_dst * p.S * p.D;
    if (n == 0) return hipSuccess;
    // (the amax launch: one same-address atomic per workgroup at most -- few workgroups)
    const unsigned cap = p.amax_word ? 512u : 4096u;
    const unsigned grid = (unsigned)((n + 255) / 256 < cap ? (n + 255) / 256 : cap);
    hipLaunchKernelGGL(dequant_kernel, dim3(grid), dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_group_sum(const float* src, void* dst, uint32_t B, uint32_t H, uint32_t Hkv, int64_t slab, hipStream_t stream,
                            int out_prec) {
    const int64_t n = (int64_t)B * Hkv * slab;
    if (n == 0) return hipSuccess;
    if (slab % 4 || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) {
        // slabs that are not a whole number of 16-byte groups (Skv * D % 4 != 0: head_dim 30 with an odd Skv, head_dim 2 ...) or
        // unaligned pointers: one element per thread (the exact engine accepts any head_dim <= 256 with grouped K / V)
        const unsigned grid1 = (unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
        hipLaunchKernelGGL(group_sum_scalar_kernel, dim3(grid1), dim3(256), 0, stream, src, dst, B, H, Hkv, slab, out_prec);
        return hipGetLastError();
    }
    const unsigned grid = (unsigned)((n / 4 + 255) / 256 < 8192 ? (n / 4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(group_sum_kernel, dim3(grid), dim3(256), 0, stream, src, dst, B, H, Hkv, slab, out_prec);
    return hipGetLastError();
}

// ------------------------------------------------------------------ mask tile flags (tile early-exit of fa_fwd16)
// One wave per (mask batch, mask head, 32-row block, 64-key tile): lane = key of the tile, 32 rows each.  Reads every
// distinct mask element once (broadcast dims are not expanded).  FwdParams::mask_flags documents the byte.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
template <bool VEC16>
__global__ __launch_bounds__(256) void mask_flags_kernel(FwdParams p, uint8_t* flags, uint32_t Bm, uint32_t Hm) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wid = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t total = (uint64_t)Bm * Hm * p.mf_nrb * p.mf_ntiles;
    if (wid >= total) return;
    const uint32_t tile = (uint32_t)(wid % p.mf_ntiles);
    const uint32_t rb = (uint32_t)((wid / p.mf_ntiles) % p.mf_nrb);
    const uint32_t slab = (uint32_t)(wid / ((uint64_t)p.mf_ntiles * p.mf_nrb));
    const uint32_t hm = slab % Hm, bm = slab / Hm;
    const uint32_t key = tile * 64 + lane;
    bool any_open = false, any_term = false;  // an element that attends / an element whose term is not +0
    if (VEC16) {
        // contiguous, 16-byte aligned rows: a lane owns one 16-byte segment (16 / es keys) of a row; a wave-load covers
        // 16 / es rows of the tile, 2 es loads cover its 32 rows.  Skv % 16 == 0 on this path: a segment is all in or out
        const int es = p.mask_kind == MK_BOOL ? 1 : (p.mask_kind == MK_F32 ? 4 : 2);
        const uint32_t spr = 4 * es, rpl = 64 / spr;  // segments per row, rows per wave-load
        const uint32_t key0 = tile * 64 + (lane % spr) * (16 / es);
        if (key0 < p.Skv) {
            const char* base = (const char*)p.mask + ((int64_t)bm * p.ms[0] + (int64_t)hm * p.ms[1] + key0) * es;
            for (uint32_t i = 0; i < 32 / rpl; ++i) {
                const uint32_t row = rb * 32 + lane / spr + rpl * i;
                if (row < p.Sq) {
                    const u32x4_t w = *(const u32x4_t*)(base + (int64_t)row * p.ms[2] * es);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (p.mask_kind == MK_BOOL) {
                            // per byte: non-zero attends (term +0), zero is masked (term -inf)
                            const uint32_t nz = ((w[j] & 0x7f7f7f7fu) + 0x7f7f7f7fu | w[j]) & 0x80808080u;  // 0x80 per non-zero byte
                            any_open |= nz != 0;
                            any_term |= nz != 0x80808080u;
                        } else if (p.mask_kind == MK_F32) {
                            // "masked" = the TERM the kernel adds is -inf (fa_common.h mask_term: value x log2 e): -inf itself, and finite values whose product
                            // overflows -- torch.finfo(torch.float32).min, the transformers idiom: its tiles are skipped like -inf ones (third session of round 6;
                            // the scalar path below has always classified the term)
                            any_open |= __uint_as_float(w[j]) * UMFA_LOG2E != -INFINITY;
                            any_term |= (w[j] & 0x7fffffffu) != 0;     // not +-0
                        } else if (p.mask_kind == MK_BF16) {
                            any_open |= __uint_as_float(w[j] << 16) * UMFA_LOG2E != -INFINITY || __uint_as_float(w[j] & 0xffff0000u) * UMFA_LOG2E != -INFINITY;  // (finfo(bfloat16).min too)
                            any_term |= (w[j] & 0x7fff7fffu) != 0;
                        } else {
                            const uint32_t ninf = 0xfc00u;  // fp16: every finite value has a finite term
                            const uint32_t lo = w[j] & 0xffffu, hi16 = w[j] >> 16;
                            any_open |= lo != ninf || hi16 != ninf;
                            any_term |= (w[j] & 0x7fff7fffu) != 0;
                        }
                    }
                }
            }
        }
    } else if (key < p.Skv) {
        const int64_t base = (int64_t)bm * p.ms[0] + (int64_t)hm * p.ms[1] + (int64_t)key * p.ms[3];
        for (uint32_t i = 0; i < 32; ++i) {
            const uint32_t row = rb * 32 + i;
            if (row >= p.Sq) break;
            const float t = mask_term(p.mask, base + (int64_t)row * p.ms[2], p.mask_kind);
            any_open |= t != -INFINITY;
            any_term |= t != 0.0f;
        }
    }
    const bool open = __builtin_amdgcn_ballot_w64(any_open) != 0, term = __builtin_amdgcn_ballot_w64(any_term) != 0;
    if (lane == 0) flags[wid] = !open ? 1 : (!term ? 2 : 0);
}

// the flag array's geometry (FwdParams::mf_*) for `flags` filled elsewhere: mask_classify_f32_kernel writes the same bytes on its way through an fp32 mask
void mask_flags_describe(FwdParams& p, const uint8_t* flags) {
    const uint32_t Hm = p.ms[1] != 0 ? p.H : 1;
    p.mf_nrb = (p.Sq + 31) / 32;
    p.mf_ntiles = (p.Skv + 63) / 64;
    p.mf_bs = p.ms[0] != 0 ? Hm : 0;
    p.mf_hs = p.ms[1] != 0 ? 1 : 0;
    p.mask_flags = flags;
}

hipError_t launch_mask_flags(FwdParams& p, uint8_t* flags, hipStream_t stream) {
    const uint32_t Bm = p.ms[0] != 0 ? p.B : 1, Hm = p.ms[1] != 0 ? p.H : 1;
    p.mf_nrb = (p.Sq + 31) / 32;
    p.mf_ntiles = (p.Skv + 63) / 64;
    p.mf_bs = p.ms[0] != 0 ? Hm : 0;
    p.mf_hs = p.ms[1] != 0 ? 1 : 0;
    const uint64_t total = (uint64_t)Bm * Hm * p.mf_nrb * p.mf_ntiles;
    const int es = p.mask_kind == MK_BOOL ? 1 : (p.mask_kind == MK_F32 ? 4 : 2);
    const bool vec16 = p.ms[3] == 1 && p.Skv % 16 == 0 && ((uintptr_t)p.mask & 15) == 0 &&
                       (p.ms[2] * es) % 16 == 0 && (p.ms[1] * es) % 16 == 0 && (p.ms[0] * es) % 16 == 0;
    if (vec16) hipLaunchKernelGGL(mask_flags_kernel<true>, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, stream, p, flags, Bm, Hm);
    else hipLaunchKernelGGL(mask_flags_kernel<false>, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, stream, p, flags, Bm, Hm);
    p.mask_flags = flags;
    return hipGetLastError();
}

// ---- realigned copy of a mask for the 128-row kernel (very end of round 6).  fa_fwd16 reads a mask four keys at a time (one dword / 8 / 16 bytes per register group) when
// rows are contiguous and aligned to four elements and Skv is a multiple of four; ANY other mask -- an odd sequence length is enough -- it reads per score with scalar loads:
// B4 H16 S1111 with an fp16 bias 464 us where S 1112 takes ~130.  One pass copies such a mask into rows padded to a multiple of four keys (pad: -inf / false), dense over its
// own batch / head / row dimensions (broadcast ones stay broadcast); FwdParams::mask_padded tells the kernel that a group starting below Skv may be read whole.
template <int ES>
__global__ __launch_bounds__(256) void mask_realign_kernel(const void* __restrict__ src, int64_t s0, int64_t s1, int64_t s2, int64_t s3, uint32_t Hm, uint32_t Sm, uint32_t Skv,
                                                           uint32_t G, void* __restrict__ dst, uint64_t total, uint32_t pad, const uint32_t* guard, uint32_t guard_want) {
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (t >= total) return;
    if (guard != nullptr && *guard != guard_want) return;  // (FwdParams::guard: the copy serves the 128-row route of an fp32 mask's guarded pair -- not taken: nothing to do)
    const uint32_t g = (uint32_t)(t % G);
    const uint64_t r = t / G;
    const uint32_t row = (uint32_t)(r % Sm);
    const uint64_t slab = r / Sm;
    const uint32_t hm = (uint32_t)(slab % Hm), bm = (uint32_t)(slab / Hm);
    const int64_t base = (int64_t)bm * s0 + (int64_t)hm * s1 + (int64_t)row * s2;
    uint32_t v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const uint32_t key = 4 * g + e;
        if (key < Skv) {
            const int64_t at = base + (int64_t)key * s3;
            v[e] = ES == 1 ? ((const uint8_t*)src)[at] : ES == 2 ? ((const uint16_t*)src)[at] : ((const uint32_t*)src)[at];
        } else {
            v[e] = pad;
        }
    }
    const uint64_t o = (r * G + g) * 4;
    if constexpr (ES == 1) ((uint32_t*)dst)[o / 4] = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
    else if constexpr (ES == 2) ((uint2*)dst)[o / 4] = make_uint2(v[0] | (v[1] << 16), v[2] | (v[3] << 16));
    else ((uint4*)dst)[o / 4] = make_uint4(v[0], v[1], v[2], v[3]);
}

static int mask_elem_bytes(int kind) { return kind == MK_BOOL ? 1 : kind == MK_F32 ? 4 : 2; }
// fa_fwd16 would read this mask per score (the negation of its `mvec`)
bool mask_rows_scalar(const FwdParams& p) {
    if (p.mask_kind != MK_BOOL && p.mask_kind != MK_F16 && p.mask_kind != MK_BF16 && p.mask_kind != MK_F32) return false;
    if (!p.mask || p.mask_padded) return false;
    const int es = mask_elem_bytes(p.mask_kind);
    return !(p.ms[3] == 1 && (p.Skv & 3) == 0 && ((p.ms[0] | p.ms[1] | p.ms[2]) & 3) == 0 && ((uintptr_t)p.mask & (uintptr_t)(4 * es - 1)) == 0);
}
size_t mask_realign_bytes(const FwdParams& p) {
    const uint64_t Bm = p.ms[0] ? p.B : 1, Hm = p.ms[1] ? p.H : 1, Sm = p.ms[2] ? p.Sq : 1, G = (p.Skv + 3) / 4;
    return (size_t)(Bm * Hm * Sm * G * 4 * mask_elem_bytes(p.mask_kind));
}
// dst: 16-byte aligned, mask_realign_bytes(p) bytes.  On return p describes the copy.
hipError_t launch_mask_realign(FwdParams& p, void* dst, hipStream_t stream) {
    if (!mask_rows_scalar(p) || !dst || ((uintptr_t)dst & 15)) return hipErrorInvalidValue;
    const uint32_t Bm = p.ms[0] ? p.B : 1, Hm = p.ms[1] ? p.H : 1, Sm = p.ms[2] ? p.Sq : 1, G = (p.Skv + 3) / 4;
    const uint64_t total = (uint64_t)Bm * Hm * Sm * G;
    if (total == 0 || (total + 255) / 256 > 0x7fffffffull) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((total + 255) / 256));
    const uint32_t pad = p.mask_kind == MK_BOOL ? 0u : p.mask_kind == MK_F16 ? 0xfc00u : p.mask_kind == MK_BF16 ? 0xff80u : 0xff800000u;
    if (p.mask_kind == MK_BOOL) hipLaunchKernelGGL(mask_realign_kernel<1>, grid, dim3(256), 0, stream, p.mask, p.ms[0], p.ms[1], p.ms[2], p.ms[3], Hm, Sm, p.Skv, G, dst, total, pad, p.guard, p.guard_want);
    else if (p.mask_kind == MK_F32) hipLaunchKernelGGL(mask_realign_kernel<4>, grid, dim3(256), 0, stream, p.mask, p.ms[0], p.ms[1], p.ms[2], p.ms[3], Hm, Sm, p.Skv, G, dst, total, pad, p.guard, p.guard_want);
    else hipLaunchKernelGGL(mask_realign_kernel<2>, grid, dim3(256), 0, stream, p.mask, p.ms[0], p.ms[1], p.ms[2], p.ms[3], Hm, Sm, p.Skv, G, dst, total, pad, p.guard, p.guard_want);
    const int64_t row = 4ll * G;
    p.mask = dst;
    p.ms[0] = p.ms[0] ? (int64_t)Hm * Sm * row : 0;
    p.ms[1] = p.ms[1] ? (int64_t)Sm * row : 0;
    p.ms[2] = p.ms[2] ? row : 0;
    p.ms[3] = 1;
    p.mask_padded = 1;
    return hipGetLastError();
}

// Is the pre-pass worth its read of the mask?  Byte masks: always (at worst +15 % for a dense random per-head mask, 2-4x
// for banded / padded ones).  Additive float masks are usually dense biases with nothing to skip: only when the distinct
// mask bytes stay below twice the Q + K + V + O traffic (e.g. one [Sq, Skv] bias shared by the heads).
// one wave per (mask batch, mask head, 256-row block): compact the visited tiles
// fp32 masks (xflag != NULL): workgroup 0 also folds the wave-tiles' exactness bytes (mask_classify_f32_kernel) into the verdict word the two guarded attention
// launches read (FwdParams::guard): 1 if any tile's fp16 copy is not exact, else 0 -- a plain store, every launch (a few KiB of bytes for every mask the route
// admits: mask_flags_worthwhile bounds the mask by twice the call's tensor traffic)
__global__ __launch_bounds__(64) void mask_list_kernel(const uint8_t* wflag, uint32_t* list, uint32_t* cnt, uint32_t nrb64, uint32_t nqb, uint32_t T,
                                                       const uint8_t* __restrict__ xflag, uint64_t xtotal, uint32_t* __restrict__ guard) {
    const uint32_t lane = threadIdx.x, qblk = blockIdx.x % nqb, slab = blockIdx.x / nqb;
    if (xflag != nullptr && blockIdx.x == 0) {
        uint32_t a = 0;
        const uint64_t n16 = xtotal / 16;
        for (uint64_t i0 = 0; i0 < n16; i0 += 256) {  // four 16-byte loads per lane in flight
            u32x4_t w4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint64_t i = i0 + 64 * u + lane;
                w4[u] = i < n16 ? ((const u32x4_t*)xflag)[i] : u32x4_t{0, 0, 0, 0};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) a |= w4[u][0] | w4[u][1] | w4[u][2] | w4[u][3];
        }
        for (uint64_t i = n16 * 16 + lane; i < xtotal; i += 64) a |= xflag[i];
        const bool any = __builtin_amdgcn_ballot_w64(a != 0) != 0;
        if (lane == 0) *guard = any ? 1u : 0u;
    }
    const uint8_t* wf = wflag + (uint64_t)slab * nrb64 * T;
    uint32_t* out = list + (uint64_t)blockIdx.x * T * 2;
    uint32_t n = 0;
    for (uint32_t t0 = 0; t0 < T; t0 += 64) {
        const uint32_t t = t0 + lane;
        uint32_t cls = 0;
        bool visit = false;
        if (t < T) {
#pragma unroll
            for (uint32_t wv = 0; wv < 4; ++wv) {
                const uint32_t rb = 4 * qblk + wv;
                const uint32_t c = rb < nrb64 ? wf[(uint64_t)rb * T + t] : 1u;  // waves past Sq: nothing to do
                cls |= c << (16 + 2 * wv);
                visit = visit || c != 1u;
            }
        }
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(visit);
        if (visit) out[2 * (n + __builtin_popcountll(bal & ((1ull << lane) - 1ull)))] = t | cls;
        n += (uint32_t)__builtin_popcountll(bal);
    }
    if (n == 0 && lane == 0) out[0] = 0u | (0x55u << 16);  // a block that sees nothing: one tile, every wave fully masked -> O = 0, LSE = -inf
    if (n == 0) n = 1;
    if (lane == 0) cnt[blockIdx.x] = n;
    __threadfence_block();  // (one wave: its own stores above are visible to its loads below once they have completed)
    __builtin_amdgcn_s_waitcnt(0);
    for (uint32_t j = lane; j < n; j += 64) {
        const uint32_t j1 = j + 1 < n ? j + 1 : n - 1, j2 = j + 2 < n ? j + 2 : n - 1;
        const uint32_t e1 = out[2 * j1];
        out[2 * j + 1] = (e1 & 0xffffu) | ((out[2 * j2] & 0xffffu) << 16);
        // bits 24 ... 31: the four waves' classes of the NEXT listed tile (the additive-mask kernel fetches a tile's mask image a tile ahead, and only
        // for waves that will read it; behind the last entry: "open", nothing to fetch).  Only bits 0 ... 23 of an entry are read back above.
        out[2 * j] |= (j + 1 < n ? (e1 >> 16) & 0xffu : 0xaau) << 24;
    }
}

size_t mask_pack_bytes(const FwdParams& p) {
    const uint64_t Bm = p.ms[0] ? p.B : 1, Hm = p.ms[1] ? p.H : 1, nrb64 = (p.Sq + 63) / 64, nqb = (p.Sq + 255) / 256, T = (p.Skv + 63) / 64;
    const uint64_t slabs = Bm * Hm;
    return (size_t)(slabs * nrb64 * T * 512 + ((slabs * nrb64 * T + 255) & ~255ull) + slabs * nqb * T * 8 + ((slabs * nqb * 4 + 255) & ~255ull) + 1024 + 2304);
}

// the scratch layout of the packed mask + the arguments of the pack kernel; fills p.mk_*
static hipError_t mask_pack_prepare(FwdParams& p, void* scratch, MaskPackArgs& a, uint32_t*& list, uint32_t*& cnt, uint32_t& nqb, uint64_t& slabs) {
    if ((p.mask_kind != MK_BOOL && p.mask_kind != MK_F16) || !p.mask || !scratch) return hipErrorInvalidValue;
    const uint32_t Bm = p.ms[0] ? p.B : 1, Hm = p.ms[1] ? p.H : 1, nrb64 = (p.Sq + 63) / 64, T = (p.Skv + 63) / 64;
    nqb = (p.Sq + 255) / 256;
    slabs = (uint64_t)Bm * Hm;
    const uint64_t total = slabs * nrb64 * T;
    if (total == 0 || total > 0x7fffffffull * 4 || T > 0xffffu) return hipErrorInvalidValue;
    char* base = (char*)scratch;
    uint32_t* bits = (uint32_t*)base;
    uint8_t* wflag = (uint8_t*)(base + total * 512);
    list = (uint32_t*)(base + total * 512 + ((total + 255) & ~255ull));
    cnt = (uint32_t*)((char*)list + slabs * nqb * T * 8);
    a.mask = p.mask;
    for (int i = 0; i < 4; ++i) a.ms[i] = p.ms[i];
    a.Sq = p.Sq; a.Skv = p.Skv;
    a.causal = p.causal ? 1 : 0;
    a.bits = bits; a.wflag = wflag;
    a.Bm = Bm; a.Hm = Hm; a.nrb64 = nrb64; a.T = T;
    a.total = total;
    a.vec16 = p.ms[3] == 1 && (p.Skv & 15) == 0 && ((p.ms[0] | p.ms[1] | p.ms[2]) & 15) == 0 && ((uintptr_t)p.mask & 15) == 0;
    a.done = false;
    p.mk_bits = bits; p.mk_list = list; p.mk_cnt = cnt;
    p.mk_bs = p.ms[0] ? Hm : 0; p.mk_hs = p.ms[1] ? 1 : 0;  // slab index of (b, h) = b * mk_bs + h * mk_hs
    p.mk_nrb64 = nrb64; p.mk_T = T;
    p.mk_prefix = nullptr;  // (the attention kernel scans the shared blocks' list lengths itself since round 5)
    if (fwd_w64_grid(p) > 512) return hipErrorInvalidValue;  // (that scan: two list lengths per thread)
    return hipSuccess;
}

static hipError_t mask_pack_finish(const MaskPackArgs& a, uint32_t* list, uint32_t* cnt, uint32_t nqb, uint64_t slabs, hipStream_t stream,
                                   const uint8_t* xflag = nullptr, uint32_t* guard = nullptr) {
    if (!a.done) {
        if (a.vec16) hipLaunchKernelGGL(mask_pack_kernel<true>, dim3((unsigned)((a.total + 3) / 4)), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL(mask_pack_kernel<false>, dim3((unsigned)((a.total + 3) / 4)), dim3(256), 0, stream, a);
    }
    hipLaunchKernelGGL(mask_list_kernel, dim3((unsigned)(slabs * nqb)), dim3(64), 0, stream, a.wflag, list, cnt, a.nrb64, nqb, a.T, xflag, (uint64_t)a.total, guard);
    return hipGetLastError();
}

hipError_t launch_mask_pack(FwdParams& p, void* scratch, hipStream_t stream) {
    MaskPackArgs a;
    uint32_t *list, *cnt, nqb;
    uint64_t slabs;
    if (hipError_t e = mask_pack_prepare(p, scratch, a, list, cnt, nqb, slabs); e != hipSuccess) return e;
    return mask_pack_finish(a, list, cnt, nqb, slabs, stream);
}

// ---- additive fp16 / bf16 masks on the one-wave-per-SIMD structure (MASKA, round 6): tile classes only -- the kernel DMAs the mask tensor itself.
// One wave per (mask batch, mask head, 64-row block, 64-key tile): a lane owns one 16-byte segment (8 keys) of a row, a wave-load covers 8 rows, 8 loads
// the tile.  class 1 = every element -inf (a tile all four waves of a block call masked is never listed), 2 = every element +-0 (the plain tile
// body, no mask read), 0 = mixed.  Preconditions (fwd_w64_supported): Sq, Skv multiples of 64, keys contiguous, 16-byte aligned rows.
// BF16: the mask is bf16 -- the same pass writes the dense fp16 copy the attention kernel reads instead (v_fma_mix_f32 takes f16 halves): exact for every
// value fp16 holds (bf16's 8-bit significands fit), finite values beyond +-65504 clamped there (exp(x - max) of such a term is 0 or the row's only
// survivor either way), magnitudes below 2^-24 flushed (e^x = 1 to fp32 precision), -inf / NaN kept.
// BF16 (third session): as for fp32 masks one WORKGROUP per (mask batch, mask head, 256-row block, 64-key tile) -- nqb blocks per slab -- whose four waves are the block's
// four wave-tiles: they exchange their classes in LDS and the copy is written only where the bias kernel will read it (see mask_classify_f32_body).  fp16 masks (no copy):
// four consecutive tiles of one 64-row block per workgroup, as before.
// SRC: 0 = fp16 mask on whole tiles, read in place by the attention kernel (classes only); 1 = bf16 mask (-> fp16 copy); 3 = fp16 mask of a RAGGED shape (Sq or Skv not a
// multiple of 64; end of round 6): the attention kernel's mask DMA wants whole 64 x 64 tiles, so the pass writes a copy PADDED to whole tiles -- keys past Skv and rows past Sq
// at -inf (a padded key must not attend; a padded row's answers are never stored) -- as it does for every bf16 / fp32 mask now.  Chunks of 16 bytes are entirely inside or
// outside the mask (the route asks for Skv % 8 == 0 with 16-bit masks, % 4 with fp32 ones).
template <int SRC>
__device__ __forceinline__ void mask_classify_body(const MaskPackArgs& p, uint32_t nqb, _Float16* copy, int64_t cb, int64_t ch, int64_t cr, const uint32_t block) {
    constexpr bool BF16 = SRC == 1, COPY = SRC != 0;
    constexpr uint32_t NINF2 = BF16 ? 0xff80ff80u : 0xfc00fc00u;  // two -inf of the source type
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __shared__ uint32_t wave_class16[4];
    uint64_t wid;
    uint32_t tile, rb, slab;
    bool live = true;
    if constexpr (COPY) {
        tile = block % p.T;
        const uint32_t qblk = (block / p.T) % nqb;
        slab = block / (p.T * nqb);
        rb = 4 * qblk + wv;
        live = rb < p.nrb64;
        if (!live) wave_class16[wv] = 1u;  // (a wave past Sq: nothing there)
        wid = ((uint64_t)slab * p.nrb64 + (live ? rb : 0u)) * p.T + tile;
    } else {
        wid = (uint64_t)block * 4 + wv;
        if (wid >= p.total) return;
        tile = (uint32_t)(wid % p.T);
        rb = (uint32_t)((wid / p.T) % p.nrb64);
        slab = (uint32_t)(wid / ((uint64_t)p.T * p.nrb64));
    }
    const uint32_t hm = slab % p.Hm, bm = slab / p.Hm;
    const uint32_t row0 = rb * 64 + (lane >> 3), key0 = tile * 64 + (lane & 7) * 8;
    const char* base = (const char*)p.mask + ((int64_t)bm * p.ms[0] + (int64_t)hm * p.ms[1] + (int64_t)row0 * p.ms[2] + key0) * 2;
    u32x4_t w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bool in = live && (!COPY || (key0 < p.Skv && (p.ms[2] == 0 || row0 + 8u * (uint32_t)i < p.Sq)));
        if (!COPY || p.vec16) {
            w[i] = in ? *(const u32x4_t*)(base + (int64_t)(8 * i) * p.ms[2] * 2) : u32x4_t{NINF2, NINF2, NINF2, NINF2};
        } else {
            // rows that are not 16-byte aligned, or an Skv that is not a multiple of 8 (any length, any view): element by element, each key checked against Skv
            const uint16_t* e = (const uint16_t*)(base + (int64_t)(8 * i) * p.ms[2] * 2);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t lo = (in && key0 + 2u * j < p.Skv) ? e[2 * j] : (NINF2 & 0xffffu), hi = (in && key0 + 2u * j + 1u < p.Skv) ? e[2 * j + 1] : (NINF2 & 0xffffu);
                w[i][j] = lo | (hi << 16);
            }
        }
    }
    if constexpr (BF16) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float lo = __uint_as_float(w[i][j] << 16), hi = __uint_as_float(w[i][j] & 0xffff0000u);
                // finite values clamped to +-65504 (one v_med3_f32), inf and NaN kept as they are, one packed conversion per pair: ~5 vector instructions per
                // element (the first form -- two compares, two selects, a scalar conversion and the re-pack per element -- cost 14 and made the pass compute-bound,
                // like the fp32 one's first build: profiles/r6/lab_notes.md section 16)
                // ... and a finite value whose log2-domain term overflows in fp32 (x log2 e = -inf: torch.finfo(torch.bfloat16).min, the transformers idiom for "masked") IS
                // -inf to the 128-row kernel, which reads the bf16 tensor itself (fa_common.h mask_term) -- so it is -inf in the copy too: the same answers on both routes (rows
                // with every key masked: O = 0, LSE = -inf), and tiles that hold nothing else are never staged.  (Before the third session it was clamped to -65504: finite.)
                auto cl = [](float x) -> float {
                    return fabsf(x) < INFINITY ? (x * UMFA_LOG2E == -INFINITY ? -INFINITY : __builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f)) : x;
                };
                const float y0 = cl(lo), y1 = cl(hi);
                uint32_t pk;
                asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(y0), "v"(y1));
                w[i][j] = pk;
            }
        }
    }
    // "every element is -inf" / "every element is +-0" off the running AND / OR of the packed words (two instructions per pair)
    uint32_t or16 = 0u, and16 = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            or16 |= w[i][j];
            and16 &= w[i][j];
        }
    const bool any_open = !(and16 == 0xfc00fc00u && or16 == 0xfc00fc00u), any_term = (or16 & 0x7fff7fffu) != 0;
    const bool open = __builtin_amdgcn_ballot_w64(any_open) != 0, term = __builtin_amdgcn_ballot_w64(any_term) != 0;
    const uint32_t my_class = !open ? 1u : (!term ? 2u : 0u);
    if constexpr (COPY) {
        if (live && lane == 0) wave_class16[wv] = my_class;
        __syncthreads();
        const bool listed = tile == 0 || wave_class16[0] != 1u || wave_class16[1] != 1u || wave_class16[2] != 1u || wave_class16[3] != 1u;
        if (live && my_class != 2u && listed) {
            const bool writer = p.ms[2] != 0 || rb == 0;  // (a mask without a row dimension: ONE row in the copy, written by the wave of row block 0 -- its lanes 0 ... 7)
            _Float16* dst = copy + (int64_t)bm * cb + (int64_t)hm * ch + (int64_t)row0 * cr + key0;
            if (writer) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (p.ms[2] != 0 || (i == 0 && lane < 8)) *(u32x4_t*)(dst + (int64_t)(8 * i) * cr) = w[i];
            }
        }
    }
    if (live && lane == 0) p.wflag[wid] = (uint8_t)my_class;
}
template <int SRC>
__global__ __launch_bounds__(256) void mask_classify_kernel(MaskPackArgs p, uint32_t nqb, _Float16* copy, int64_t cb, int64_t ch, int64_t cr) {
    mask_classify_body<SRC>(p, nqb, copy, cb, ch, cr, blockIdx.x);
}

// ---- fp32 additive masks on the same structure (end of round 6; the reference's additive masks are fp32 wherever its own callers build them:
// metal_sdpa_backend.cpp:3210-3231, MFABridge.swift:157-242).  The bias kernels read fp16 (v_fma_mix_f32 takes f16 halves; a 64 x 64 fp32 tile would not fit the
// wave's staging ring), so the classification pass writes a dense fp16 copy as for bf16 masks -- but fp32 values need not fit fp16, and rounding a term of
// magnitude m moves the logit by up to m 2^-11: 8e-3 at m = 16, past the 1e-3 bound.  So the pass also records, per wave-tile, whether fp16 holds every value
// EXACTLY (xflag; "exactly" = the round trip through fp16 returns the value -- every fp16 value, +-inf --, or the magnitude is below 2^-24 (flushed
// without moving e^x in fp32), or the value is -inf to every kernel anyway: see the kernel), mask_list_kernel's first workgroup folds the bytes into one verdict word, and the two attention launches the runtime enqueues -- the bias
// kernel on the copy, the 128-row kernel on the caller's tensor -- are guarded by it (FwdParams::guard): exactly one of them runs.  0 / -inf masks, masks
// built in fp16 / bf16 and widened, small-integer and dyadic biases take the fast kernel; anything else keeps today's kernel and today's numbers.
// One WORKGROUP per (mask batch, mask head, 256-row block, 64-key tile), its four waves = the block's four 64-row wave-tiles (what the attention kernel's four
// waves see of the tile): a lane owns one 16-byte segment (4 keys) of a row, a wave-load covers 4 rows, 16 loads the wave-tile.
__device__ __forceinline__ void mask_classify_f32_body(const MaskPackArgs& p, uint32_t nqb, _Float16* copy, int64_t cb, int64_t ch, int64_t cr, uint8_t* xflag, uint8_t* flags128,
                                                       const uint32_t block) {
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    __shared__ uint32_t wave_class[4];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t tile = block % p.T;
    const uint32_t qblk = (block / p.T) % nqb;
    const uint32_t slab = block / (p.T * nqb);
    const uint32_t rb = 4 * qblk + wv;
    if (rb >= p.nrb64) wave_class[wv] = 1u;  // (a wave past Sq: nothing there)
    const bool live = rb < p.nrb64;
    const uint64_t wid = ((uint64_t)slab * p.nrb64 + (live ? rb : 0u)) * p.T + tile;
    const uint32_t hm = slab % p.Hm, bm = slab / p.Hm;
    const uint32_t row0 = rb * 64 + (lane >> 4), key0 = tile * 64 + (lane & 15) * 4;
    const char* base = (const char*)p.mask + ((int64_t)bm * p.ms[0] + (int64_t)hm * p.ms[1] + (int64_t)row0 * p.ms[2] + key0) * 4;
    u32x4_t w[16];  // the whole tile in flight: 16 loads of 4 rows each
#pragma unroll
    for (int i = 0; i < 16; ++i) {  // (ragged shapes: chunks outside the mask -- keys past Skv, rows past Sq -- count and are copied as -inf; see mask_classify_body)
        const bool in = live && key0 < p.Skv && (p.ms[2] == 0 || row0 + 4u * (uint32_t)i < p.Sq);
        if (p.vec16) {
            w[i] = in ? *(const u32x4_t*)(base + (int64_t)(4 * i) * p.ms[2] * 4) : u32x4_t{0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u};
        } else {  // (unaligned rows / an Skv that is not a multiple of 4: element by element)
            const uint32_t* e = (const uint32_t*)(base + (int64_t)(4 * i) * p.ms[2] * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) w[i][j] = (in && key0 + (uint32_t)j < p.Skv) ? e[j] : 0xff800000u;
        }
    }
    // the 128-row route's tile flags (FwdParams::mask_flags: per 32-row block and 64-key tile, from the fp32 VALUES as mask_flags_kernel reads them) come from
    // the same read: rows 0 ... 31 of the tile are loads 0 ... 7, rows 32 ... 63 loads 8 ... 15.
    // ~8 vector instructions per element (the first build's ~25 made the pass compute-bound: 4096 elements per wave, four waves per SIMD): "every element is -inf" /
    // "every element is +-0" are read off the running OR and AND of the words; exactness is the round trip through fp16 (round-to-nearest-even, overflow -> +-inf,
    // so finite values beyond +-65504 fail it) OR a magnitude fp16 flushes harmlessly OR a value that IS -inf to every kernel of this library -- a finite x whose
    // log2-domain term x log2 e overflows in fp32 (torch.finfo(torch.float32).min, the "large negative" idiom at its largest; fa_common.h mask_term): its copy is
    // -inf (the conversion's own overflow).  NaN fails the round trip: the 128-row kernel gets it.
    constexpr float NINF_LIM = -0x1.62e42ep+127f;  // x < this  =>  x * UMFA_LOG2E == -inf in fp32 (0x1.62e42fefp+127 = FLT_MAX ln 2, rounded towards zero: a few ulps of slack stay finite-and-inexact)
    uint32_t or32[2] = {0u, 0u}, or16 = 0u, and16 = 0xffffffffu;
    bool open32[2] = {false, false}, inexact = false;
    u32x2_t o[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t w0 = w[i][2 * j], w1 = w[i][2 * j + 1];
            const float x0 = __uint_as_float(w0), x1 = __uint_as_float(w1);
            uint32_t pk;
            asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(x0), "v"(x1));
            const float b0 = (float)__builtin_bit_cast(_Float16, (uint16_t)(pk & 0xffffu)), b1 = (float)__builtin_bit_cast(_Float16, (uint16_t)(pk >> 16));
            const bool ok0 = b0 == x0 || fabsf(x0) <= 0x1p-24f || x0 < NINF_LIM, ok1 = b1 == x1 || fabsf(x1) <= 0x1p-24f || x1 < NINF_LIM;
            inexact = inexact || !ok0 || !ok1;
            or32[i >> 3] |= w0 | w1;
            open32[i >> 3] = open32[i >> 3] || x0 * UMFA_LOG2E != -INFINITY || x1 * UMFA_LOG2E != -INFINITY;  // (the 128-row kernel's definition of "masked": the term is -inf)
            or16 |= pk;
            and16 &= pk;
            o[i][j] = pk;
        }
    }
    const bool any_open = !(and16 == 0xfc00fc00u && or16 == 0xfc00fc00u), any_term = (or16 & 0x7fff7fffu) != 0;
    bool term32[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) term32[hf] = (or32[hf] & 0x7fffffffu) != 0;
    const bool open = __builtin_amdgcn_ballot_w64(any_open) != 0, term = __builtin_amdgcn_ballot_w64(any_term) != 0;
    const bool bad = __builtin_amdgcn_ballot_w64(inexact) != 0;
    // Who reads the copy: a wave whose class for a LISTED tile is not "all zero" (fa_fwd16_w64_kernel.inc: mk_cls != 2 -- a wave-tile at -inf throughout
    // reads its -inf like any other), and a tile is listed when any of the block's four waves is not masked there (mask_list_kernel).  Nothing else is
    // written: the masked tiles of a 0 / -inf document or window mask cost their read, not a write.
    const uint32_t my_class = !open ? 1u : (!term ? 2u : 0u);
    if (live && lane == 0) wave_class[wv] = my_class;
    __syncthreads();
    // (tile 0 as well: a block that sees nothing anywhere lists its tile 0, every wave masked -- mask_list_kernel -- and its waves read their -inf there)
    const bool listed = tile == 0 || wave_class[0] != 1u || wave_class[1] != 1u || wave_class[2] != 1u || wave_class[3] != 1u;
    if (live && my_class != 2u && listed) {
        const bool writer = p.ms[2] != 0 || rb == 0;  // (a mask without a row dimension: ONE row in the copy, written by the wave of row block 0 -- its lanes 0 ... 15)
        _Float16* dst = copy + (int64_t)bm * cb + (int64_t)hm * ch + (int64_t)row0 * cr + key0;
        if (writer) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (p.ms[2] != 0 || (i == 0 && lane < 16)) *(u32x2_t*)(dst + (int64_t)(4 * i) * cr) = o[i];
        }
    }
    bool f_open[2], f_term[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        f_open[hf] = __builtin_amdgcn_ballot_w64(open32[hf]) != 0;
        f_term[hf] = __builtin_amdgcn_ballot_w64(term32[hf]) != 0;
    }
    if (live && lane == 0) {
        p.wflag[wid] = (uint8_t)my_class;
        xflag[wid] = bad ? 1 : 0;  // (always written: nothing to clean between launches, nothing atomic)
        if (flags128) {
            const uint32_t nrb32 = (p.Sq + 31) / 32;  // (= FwdParams::mf_nrb; a ragged Sq may leave the last 64-row block one 32-row block only)
            uint8_t* f = flags128 + ((uint64_t)slab * nrb32 + 2 * rb) * p.T + tile;
            f[0] = !f_open[0] ? 1 : (!f_term[0] ? 2 : 0);
            if (2 * rb + 1 < nrb32) f[p.T] = !f_open[1] ? 1 : (!f_term[1] ? 2 : 0);
        }
    }
}
__global__ __launch_bounds__(256) void mask_classify_f32_kernel(MaskPackArgs p, uint32_t nqb, _Float16* copy, int64_t cb, int64_t ch, int64_t cr, uint8_t* xflag, uint8_t* flags128) {
    mask_classify_f32_body(p, nqb, copy, cb, ch, cr, xflag, flags128, blockIdx.x);
}

// The classification pass in the V cast pass's launch (bf16 operands: every additive-mask call on the bias kernels has one): blocks [0, cast_blocks) cast -- slab-major,
// first: their exchange relies on it --, the rest classify.  The two passes read different tensors and are each short of the chip's bandwidth on their own (12 us of cast,
// 8 ... 26 us of classification at FLUX size); in one launch they overlap and the call is a launch shorter -- as the bool masks' re-pack has ridden there since round 5.
// KIND: 0 fp16 mask (classes only), 1 bf16 (+ the fp16 copy), 2 fp32 (+ the copy, exactness bytes, the 128-row kernel's tile flags), 3 fp16 of a ragged shape (+ the padded copy)
struct MaskClassifyExtra {
    uint32_t nqb;
    _Float16* copy;
    int64_t cb, ch, cr;
    uint8_t* xflag;
    uint8_t* flags128;
};
template <int U, int KIND>
__global__ __launch_bounds__(256) void cast_rows_and_mask_classify_kernel(CastRowsArgs a, uint32_t cast_blocks, MaskPackArgs mk, MaskClassifyExtra x) {
    if (blockIdx.x < cast_blocks) cast_rows_body<U, true>(a, blockIdx.x);
    else if constexpr (KIND == 2) mask_classify_f32_body(mk, x.nqb, x.copy, x.cb, x.ch, x.cr, x.xflag, x.flags128, blockIdx.x - cast_blocks);
    else mask_classify_body<KIND>(mk, x.nqb, x.copy, x.cb, x.ch, x.cr, blockIdx.x - cast_blocks);
}
// false: the cast takes its two-launch form (or the grids do not fit one launch): the caller launches the two passes one after the other
template <int U>
static bool launch_cast_and_classify_u(const void* src, const int64_t* strides, void* dst, uint32_t B, uint32_t H, uint32_t S, uint32_t D, uint32_t* hdr, hipStream_t stream,
                                       const MaskPackArgs& mk, const MaskClassifyExtra& x, int kind, unsigned classify_grid) {
    const uint32_t D8 = D / 8, rpw = 256u / D8;
    const uint64_t chunks = ((uint64_t)S + U * rpw - 1) / (U * rpw), grid = (uint64_t)B * H * chunks;
    if (chunks > 64 || tuning().cast_two_pass.load(std::memory_order_relaxed) || grid + classify_grid > 0x7fffffffull) return false;
    const CastRowsArgs a = {(const uint16_t*)src, strides[0], strides[1], strides[2], (_Float16*)dst, H, S, D8, (uint32_t)chunks, hdr,
                            (uint32_t)std::min<int64_t>(std::max(tuning().cast_wait_us.load(std::memory_order_relaxed), 0), 1000000) * 100u};
    const dim3 g((unsigned)grid + classify_grid);
    if (kind == 2) hipLaunchKernelGGL((cast_rows_and_mask_classify_kernel<U, 2>), g, dim3(256), 0, stream, a, (unsigned)grid, mk, x);
    else if (kind == 3) hipLaunchKernelGGL((cast_rows_and_mask_classify_kernel<U, 3>), g, dim3(256), 0, stream, a, (unsigned)grid, mk, x);
    else if (kind == 1) hipLaunchKernelGGL((cast_rows_and_mask_classify_kernel<U, 1>), g, dim3(256), 0, stream, a, (unsigned)grid, mk, x);
    else hipLaunchKernelGGL((cast_rows_and_mask_classify_kernel<U, 0>), g, dim3(256), 0, stream, a, (unsigned)grid, mk, x);
    return true;
}

static inline size_t up256(uint64_t n) { return (size_t)((n + 255) & ~255ull); }

// bytes behind the pack area of an additive mask's scratch block: bf16 masks -- the dense fp16 copy [Bm, Hm, Sq or 1, Skv]; fp32 masks -- that copy, the
// wave-tiles' exactness bytes, the verdict word (a 256-byte block of its own) and the 128-row route's tile flags (mask_flags_bytes); 0 for fp16 masks
// 16-byte chunks of the mask's rows are aligned and entirely inside or outside the mask (the classification passes' vector loads; the bias kernels' DMA of an fp16 mask in place)
static bool mask_rows_vector(const FwdParams& p) {
    const int64_t chunk = p.mask_kind == MK_F32 ? 4 : 8;
    return p.ms[3] == 1 && ((uintptr_t)p.mask & 15) == 0 && (p.ms[0] % chunk) == 0 && (p.ms[1] % chunk) == 0 && (p.ms[2] % chunk) == 0 && (p.Skv % chunk) == 0;
}
// does the bias route read a COPY of this additive mask (written by the classification pass, fp16, padded to whole tiles)?  bf16 / fp32: always; fp16: when the attention kernel's
// DMA cannot read the caller's tensor in place -- a ragged shape, unaligned rows
bool mask_needs_copy(const FwdParams& p) {
    if (p.mask_kind == MK_BF16 || p.mask_kind == MK_F32) return true;
    if (p.mask_kind != MK_F16 || p.mask_padded) return false;
    return p.Sq % 64 != 0 || p.Skv % 64 != 0 || !mask_rows_vector(p);
}

size_t mask_copy_bytes(const FwdParams& p) {
    if (!mask_needs_copy(p)) return 0;
    // the copy is padded to whole 64 x 64 tiles (the attention kernel's mask DMA reads whole tiles)
    const uint64_t Bm = p.ms[0] ? p.B : 1, Hm = p.ms[1] ? p.H : 1, Sm = p.ms[2] ? ((p.Sq + 63) / 64) * 64ull : 1, Skp = ((p.Skv + 63) / 64) * 64ull;
    const size_t copy = up256(Bm * Hm * Sm * Skp * 2);
    if (p.mask_kind != MK_F32) return copy;
    // (... and, when the 128-row kernel would read the caller's tensor per score -- rows not aligned to four elements --, its realigned copy: launch_mask_realign under the guard)
    return copy + up256(Bm * Hm * ((p.Sq + 63) / 64) * ((p.Skv + 63) / 64)) + 256 + up256(mask_flags_bytes(p)) + (mask_rows_scalar(p) ? up256(mask_realign_bytes(p)) : 0);
}

// scratch = [the pack layout of mask_pack_bytes | the fp16 copy of a bf16 mask].  On return p describes what the attention kernel reads: an fp16 mask.
// fp32 masks: scratch = [pack | fp16 copy | exactness bytes | verdict word (256 bytes) | the 128-row route's tile flags]; p.guard = the verdict word, guard_want = 0
// (the bias kernel's side: runtime.hip enqueues the 128-row kernel behind it with guard_want = 1 and its flags at (uint8_t*)p.guard + 256).
// cast != NULL: the V cast pass of the same call (launch_cast_rows_bf16_to_f16's arguments) rides in the classification's launch when its one-launch form applies;
// else it is launched here, in front of the classification
hipError_t launch_mask_classify(FwdParams& p, void* scratch, hipStream_t stream, const CastRowsCall* cast) {
    if (p.mask_kind != MK_F16 && p.mask_kind != MK_BF16 && p.mask_kind != MK_F32) return hipErrorInvalidValue;
    const bool bf = p.mask_kind == MK_BF16, f32 = p.mask_kind == MK_F32;
    const bool f16c = p.mask_kind == MK_F16 && mask_needs_copy(p);  // fp16 mask of a ragged shape / with unaligned rows: the padded copy
    const bool vec_rows = mask_rows_vector(p);
    const size_t pack_bytes = (mask_pack_bytes(p) + 255) & ~(size_t)255;
    MaskPackArgs a;
    uint32_t *list, *cnt, nqb;
    uint64_t slabs;
    const int kind_in = p.mask_kind;
    if (bf || f32) p.mask_kind = MK_F16;  // (the layout of the pack area does not depend on the kind; mask_pack_prepare takes bool / fp16)
    if (hipError_t e = mask_pack_prepare(p, scratch, a, list, cnt, nqb, slabs); e != hipSuccess) { p.mask_kind = kind_in; return e; }
    p.mk_bits = nullptr;
    a.vec16 = vec_rows;  // (for the classification bodies: vector loads of the source rows)
    const unsigned grid = (unsigned)((a.total + 3) / 4);
    const uint8_t* xflag_f32 = nullptr;
    MaskClassifyExtra x = {nqb, nullptr, 0, 0, 0, nullptr, nullptr};
    unsigned cgrid = grid;
    int kind = bf ? 1 : f32 ? 2 : f16c ? 3 : 0;
    bool classify = true;  // (false: the mask is too large to be read twice -- every wave-tile is called mixed without looking)
    if (bf || f32 || f16c) {
        _Float16* copy = (_Float16*)((char*)scratch + pack_bytes);
        const int64_t Skp = (int64_t)a.T * 64, Sm = p.ms[2] ? (int64_t)a.nrb64 * 64 : 1, cr = p.ms[2] ? Skp : 0, chd = Sm * Skp, cbt = (int64_t)a.Hm * chd;  // padded to whole tiles
        x.copy = copy; x.cb = cbt; x.ch = chd; x.cr = cr;
        {
            const uint64_t wgs = slabs * nqb * a.T;  // one workgroup per (256-row block, key tile)
            if (wgs > 0x7fffffffull) { p.mask_kind = kind_in; return hipErrorInvalidValue; }
            cgrid = (unsigned)wgs;
        }
        if (f32) {
            uint8_t* xflag = (uint8_t*)copy + up256((uint64_t)a.Bm * a.Hm * Sm * Skp * 2);
            uint32_t* guard = (uint32_t*)(xflag + up256(a.total));
            // ... and the 128-row route's tile flags behind the verdict word: [Bm Hm][2 nrb64][T] bytes = the layout of launch_mask_flags (Sq is a multiple of 64 here)
            x.xflag = xflag; x.flags128 = (uint8_t*)guard + 256;
            xflag_f32 = xflag;
            p.guard = guard;
            p.guard_want = 0;
        }
        p.mask = copy;
        p.mask_padded = 1;
        p.ms[0] = p.ms[0] ? cbt : 0; p.ms[1] = p.ms[1] ? chd : 0; p.ms[2] = cr; p.ms[3] = 1;
    } else if (!mask_flags_worthwhile(p)) {
        classify = false;
    }
    bool cast_done = false;
    if (cast && classify) {
        // the same choice of loads per thread as launch_cast_rows_any
        const uint32_t rpw = 256u / (cast->D / 8);
        const uint64_t wg16 = (uint64_t)cast->B * cast->H * (((uint64_t)cast->S + 16 * rpw - 1) / (16 * rpw)), chunks16 = ((uint64_t)cast->S + 16 * rpw - 1) / (16 * rpw);
        const int lab = tuning().cast_u.load(std::memory_order_relaxed);
        const bool ok_args = cast->src && cast->dst && cast->hdr && !(cast->D & 7) && cast->D <= 2048 && cast->strides[3] == 1 &&
                             !((cast->strides[0] | cast->strides[1] | cast->strides[2]) % 8) && !((uintptr_t)cast->src & 15) && (int64_t)cast->B * cast->H * cast->S * cast->D != 0;
        if (ok_args) {
            if (lab == 32 || (!lab && chunks16 > 64 && chunks16 <= 128))
                cast_done = launch_cast_and_classify_u<32>(cast->src, cast->strides, cast->dst, cast->B, cast->H, cast->S, cast->D, cast->hdr, stream, a, x, kind, cgrid);
            else if (lab == 4 || (!lab && wg16 < (uint64_t)device_cu_count()))
                cast_done = launch_cast_and_classify_u<4>(cast->src, cast->strides, cast->dst, cast->B, cast->H, cast->S, cast->D, cast->hdr, stream, a, x, kind, cgrid);
            else
                cast_done = launch_cast_and_classify_u<16>(cast->src, cast->strides, cast->dst, cast->B, cast->H, cast->S, cast->D, cast->hdr, stream, a, x, kind, cgrid);
        }
    }
    if (cast && !cast_done) {
        if (hipError_t e = launch_cast_rows_any(cast->src, cast->strides, cast->dst, cast->B, cast->H, cast->S, cast->D, cast->hdr, stream, nullptr); e != hipSuccess) { p.mask_kind = kind_in; return e; }
    }
    if (!cast_done || !classify) {
        if (!classify) {
            // a mask whose distinct bytes exceed twice the call's Q + K + V + O traffic (a dense per-head bias: 805 MB at the FLUX shape) is read ONCE, by the
            // attention kernel: every wave-tile is called mixed (class 0, every tile listed) without looking -- the 128-row kernel's rule for its flags pass
            // (mask_flags_worthwhile); a bias has nothing to skip, and the pass would cost what the attention itself costs
            if (hipError_t e = hipMemsetAsync(a.wflag, 0, a.total, stream); e != hipSuccess) return e;
        } else if (f32) {
            hipLaunchKernelGGL(mask_classify_f32_kernel, dim3(cgrid), dim3(256), 0, stream, a, nqb, x.copy, x.cb, x.ch, x.cr, x.xflag, x.flags128);
        } else if (bf) {
            hipLaunchKernelGGL(mask_classify_kernel<1>, dim3(cgrid), dim3(256), 0, stream, a, nqb, x.copy, x.cb, x.ch, x.cr);
        } else if (f16c) {
            hipLaunchKernelGGL(mask_classify_kernel<3>, dim3(cgrid), dim3(256), 0, stream, a, nqb, x.copy, x.cb, x.ch, x.cr);
        } else {
            hipLaunchKernelGGL(mask_classify_kernel<0>, dim3(grid), dim3(256), 0, stream, a, nqb, (_Float16*)nullptr, 0, 0, 0);
        }
    }
    a.done = true;  // (no bit image to pack)
    return mask_pack_finish(a, list, cnt, nqb, slabs, stream, xflag_f32, const_cast<uint32_t*>(p.guard));
}

hipError_t launch_cast_rows_bf16_to_f16(const void* src, const int64_t* strides, void* dst, uint32_t B, uint32_t H, uint32_t S, uint32_t D,
                                        uint32_t* hdr, hipStream_t stream) {
    return launch_cast_rows_any(src, strides, dst, B, H, S, D, hdr, stream, nullptr);
}

hipError_t launch_cast_rows_and_mask_pack(const void* src, const int64_t* strides, void* dst, uint32_t B, uint32_t H, uint32_t S, uint32_t D,
                                          uint32_t* hdr, FwdParams& p, void* mask_scratch, hipStream_t stream) {
    MaskPackArgs a;
    uint32_t *list, *cnt, nqb;
    uint64_t slabs;
    if (hipError_t e = mask_pack_prepare(p, mask_scratch, a, list, cnt, nqb, slabs); e != hipSuccess) return e;
    if (hipError_t e = launch_cast_rows_any(src, strides, dst, B, H, S, D, hdr, stream, &a); e != hipSuccess) return e;
    return mask_pack_finish(a, list, cnt, nqb, slabs, stream);  // (the pack itself rode along unless the cast took its two-launch form)
}

bool mask_flags_worthwhile(const FwdParams& p) {
    if (p.mask_kind == MK_WINDOW) return false;  // its tile flags are arithmetic, inside the kernel
    if (p.mask_kind == MK_BOOL) return true;
    const uint64_t Bm = p.ms[0] != 0 ? p.B : 1, Hm = p.ms[1] != 0 ? p.H : 1;
    const uint64_t es = p.mask_kind == MK_F32 ? 4 : 2, eb = p.in_prec == P_FP32 ? 4 : 2;
    const uint64_t mask_bytes = Bm * Hm * p.Sq * p.Skv * es;
    const uint64_t qkvo = (uint64_t)p.B * p.H * p.D * ((uint64_t)p.Sq * (eb + 4) + 2ull * p.Skv * eb);
    // ... 2 x: set for dense per-head biases, where the pass is pure cost.  A float mask with a BATCH dimension and no head dimension ([B, 1, Sq, Skv]: what a padding /
    // document / causal + padding mask looks like once it is additive -- the transformers idiom, 0 / finfo.min) is rarely a bias: up to 8 x.  Measured at 3.2 and 6.4 x
    // (profiles/r6/mask_pass_rule_probe.jsonl, rule 2 -> 8): documents -40 ... -75 %, per-sample key padding -12 ... -45 %, a dense bias of that shape +10 ... +17 % on the
    // bias kernels and +16 ... +35 % on the 128-row kernel.  Lab option mask_pass_ratio overrides both.
    const int lab = tuning().mask_pass_ratio.load(std::memory_order_relaxed);
    const uint64_t limit = lab > 0 ? (uint64_t)lab : (p.ms[0] != 0 && p.ms[1] == 0 ? 8u : 2u);
    return mask_bytes <= limit * qkvo;
}

size_t mask_flags_bytes(const FwdParams& p) {
    const uint64_t Bm = p.ms[0] != 0 ? p.B : 1, Hm = p.ms[1] != 0 ? p.H : 1;
    return (size_t)(Bm * Hm * ((p.Sq + 31) / 32) * ((p.Skv + 63) / 64));
}

}  // namespace umfa
