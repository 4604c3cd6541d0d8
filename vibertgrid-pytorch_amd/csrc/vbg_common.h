// Shared device/host helpers for libvbg (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VBG_OK 0
#define VBG_EARG (-1)

// every C-ABI entry point: 0 ok, negative = argument error, positive = hipError_t
#define VBG_CHECK_ARG(cond) do { if (!(cond)) return VBG_EARG; } while (0)
// clear any stale sticky error left by unrelated runtime calls on this thread, then launch
#define VBG_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)
#define VBG_LAUNCH_RET() do { hipError_t e__ = hipGetLastError(); return e__ == hipSuccess ? VBG_OK : (int)e__; } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace vbg {

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- wave / block reductions (wave = 64 lanes) ------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// block reduce for blockDim.x <= 1024 (multiple of 64); `sh` has >= 16 floats; result broadcast
__device__ __forceinline__ float block_sum(float v, float* sh) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += sh[i];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* sh) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    float r = sh[0];
    for (int i = 1; i < nw; ++i) r = fmaxf(r, sh[i]);
    return r;
}

// ---- counter-based RNG for dropout: one 32-bit draw per (seed, stream, index) ----------
// (splitmix64-style finaliser; deterministic, so backward regenerates the forward mask)
__host__ __device__ __forceinline__ uint32_t rng_u32(uint64_t seed, uint64_t stream, uint64_t idx) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (stream + 1) + idx * 0xBF58476D1CE4E5B9ull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(z >> 16);
}
// keep with probability 1-p: threshold on a 32-bit draw
__host__ __device__ __forceinline__ bool rng_keep(uint64_t seed, uint64_t stream, uint64_t idx, uint32_t drop_thresh) {
    return rng_u32(seed, stream, idx) >= drop_thresh;
}
static inline uint32_t drop_threshold(float p) {
    double t = (double)p * 4294967296.0;
    if (t <= 0) return 0u;
    if (t >= 4294967295.0) return 4294967295u;
    return (uint32_t)t;
}

// erf to < 1 ulp-and-a-bit without a branch: both pieces of N. Juffa's single-precision erf (minimax polynomial in x^2 below 0.9277,
// 1 - exp(polynomial) above) are evaluated and one is selected -- 21 VALU instructions against the 33 + two exec-mask switches the
// library's erff costs once a wave's lanes straddle |x| = 1 (they always do in a GELU epilogue); exp through v_exp_f32 (the argument
// is <= 0: the absolute error it adds to erf stays below 4e-8).  Checked against erf in fp64 over [-6, 6] and N(0, 1.5) samples:
// 0.94 ulp with an exact exp (tests/test_gpu_kernels.py::test_gelu_erf_epilogues).
__device__ __forceinline__ float vbg_erff(float a) {
    const float t = fabsf(a), s = a * a;
    float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = fmaf(r, s, u);
    r = fmaf(r, t, -1.06777877e-1f);
    r = fmaf(r, t, -6.34846687e-1f);
    r = fmaf(r, t, -1.28717512e-1f);
    r = fmaf(r, t, -t);
    const float big = 1.0f - __expf(r);
    float q = -5.96761703e-4f;
    q = fmaf(q, s, 4.99119423e-3f);
    q = fmaf(q, s, -2.67681349e-2f);
    q = fmaf(q, s, 1.12819925e-1f);
    q = fmaf(q, s, -3.76125336e-1f);
    q = fmaf(q, s, 1.28379166e-1f);
    q = fmaf(q, t, t);
    return copysignf(t > 0.927734375f ? big : q, a);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + vbg_erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + vbg_erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

}  // namespace vbg

// ---- amax slots (include/vbg.h VBG_AMAX_WORDS / VBG_AMAX_STRIDE) and the power-of-two scale derived from one ------------------------
#if defined(__HIPCC__)
#ifndef VBG_AMAX_WORDS
#define VBG_AMAX_WORDS 64
#define VBG_AMAX_STRIDE 32
#endif
// the value of an amax slot: the max over its 64 words, uniform across the wave (every wave of the caller may call it)
__device__ __forceinline__ unsigned vbg_amax_read(const unsigned* slot) {
    unsigned v = slot[(threadIdx.x & (VBG_AMAX_WORDS - 1)) * VBG_AMAX_STRIDE];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o, 64));
    return __builtin_amdgcn_readfirstlane(v);
}
// power-of-two scale that brings a tensor whose largest magnitude has the bit pattern `amax_bits` to [2^13, 2^14): (scale, 1 / scale).
// Exact in fp32; amax = 0 or outside 2^-100 ... 2^113: no scaling.
__device__ __forceinline__ float2 vbg_pow2_scale(unsigned amax_bits) {
    const int e = (int)((amax_bits >> 23) & 0xffu);            // biased exponent of amax
    if (e < 27 || e > 240) return make_float2(1.f, 1.f);
    return make_float2(__uint_as_float((unsigned)(267 - e) << 23), __uint_as_float((unsigned)(e - 13) << 23));
}
// block-wide max of non-negative floats into an amax slot: one atomic per block at most, on the block's own word (L2 line)
__device__ __forceinline__ void vbg_amax_publish(float mx, unsigned* amax, float* sh16) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) sh16[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) mx = fmaxf(mx, sh16[w]);
        const unsigned bits = __float_as_uint(mx);
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        unsigned* word = amax + (lin & (VBG_AMAX_WORDS - 1)) * VBG_AMAX_STRIDE;
        if (bits > __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(word, bits);
    }
}
#endif
