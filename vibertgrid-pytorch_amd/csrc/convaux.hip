// NHWC helpers around the implicit-GEMM convolutions of the ResNet-FPN trunk and heads
// (model/ResNetFPN_ViBERTgrid.py:106-184, 478-508; model/semantic_segmentation_head.py:66-78;
// model/field_type_classification_head.py:64-75): BatchNorm statistics / apply / backward
// (Sync-able: statistics are plain sums the caller may all-reduce), 3x3/s2 max-pool, nearest
// up/down-sampling of the FPN, layout changes, input normalisation + bilinear resize
// (pipeline/transform.py:104-157), stem im2col.  All HBM-bound streaming kernels.
#include "vbg_common.h"
#include "../../include/vbg.h"

namespace vbg {

static inline int ew_grid(long long n, int block) {
    long long g = (n + block - 1) / block;
    if (g > 256 * 8) g = 256 * 8;
    if (g < 1) g = 1;
    return (int)g;
}

// ------------------------------------------------------------------------------------------
// BatchNorm.  x is [M, C] (NHWC rows).
// ------------------------------------------------------------------------------------------
// Reduction layout shared by bn_stats / bn_bwd_reduce: a block covers up to 64 channel QUADS (float4) x RL row lanes
// (256 threads); every load is a float4 so a wave touches >= 1 KB of contiguous NHWC rows; per-thread partials in fp32
// over its rows, cross-lane reduction through LDS in fp64, one fp64 atomic per channel per block.
// Same-address fp64 atomics serialise at ~0.1 us each on MI355X (measured: 2048 row-blocks -> 190 us for a 134 MB
// tensor), so (a) the block sums land in one of VBG_BN_SLOTS slot rows (slot = row-block % slots; the consumers fold
// the slots) and (b) the host picks rows-per-block so the grid has ~512 blocks (tools/bn_bench.py sweep: 5.5 TB/s on the
// 134 MB stem map; more blocks only add atomics): <= 16 atomics per address, and small maps (layer4: 2048 rows) still
// spread over > 100 blocks instead of 16.
constexpr int BN_SLOTS = 32;
template <bool BWD>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const float* __restrict__ a, const float* __restrict__ y,
                                                        const float* __restrict__ x, long long M, int C,
                                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                                        int relu, int rows_per_block, double* out) {
    __shared__ double sh[2][256][4];
    const int C4 = C >> 2;
    const int QB = min(C4, 64);                     // quads handled per block
    const int RL = 256 / QB;                        // row lanes
    const int q = threadIdx.x % QB, rl = threadIdx.x / QB;
    const int cq = blockIdx.x * 64 + q;             // channel quad
    const long long r0 = (long long)blockIdx.y * rows_per_block;
    const long long r1 = min(M, r0 + rows_per_block);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cq < C4 && rl < RL) {
        float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), is = mu;
        if (BWD) { mu = reinterpret_cast<const float4*>(mean)[cq]; is = reinterpret_cast<const float4*>(invstd)[cq]; }
#pragma unroll 4
        for (long long r = r0 + rl; r < r1; r += RL) {
            const long long o = r * C4 + cq;
            float4 v = reinterpret_cast<const float4*>(a)[o];
            if (!BWD) {
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                t.x += v.x * v.x; t.y += v.y * v.y; t.z += v.z * v.z; t.w += v.w * v.w;
            } else {
                if (relu) {
                    const float4 yv = reinterpret_cast<const float4*>(y)[o];
                    v.x = yv.x > 0.f ? v.x : 0.f; v.y = yv.y > 0.f ? v.y : 0.f;
                    v.z = yv.z > 0.f ? v.z : 0.f; v.w = yv.w > 0.f ? v.w : 0.f;
                }
                const float4 xv = reinterpret_cast<const float4*>(x)[o];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                t.x += v.x * ((xv.x - mu.x) * is.x); t.y += v.y * ((xv.y - mu.y) * is.y);
                t.z += v.z * ((xv.z - mu.z) * is.z); t.w += v.w * ((xv.w - mu.w) * is.w);
            }
        }
    }
    sh[0][threadIdx.x][0] = s.x; sh[0][threadIdx.x][1] = s.y; sh[0][threadIdx.x][2] = s.z; sh[0][threadIdx.x][3] = s.w;
    sh[1][threadIdx.x][0] = t.x; sh[1][threadIdx.x][1] = t.y; sh[1][threadIdx.x][2] = t.z; sh[1][threadIdx.x][3] = t.w;
    __syncthreads();
    if (rl == 0 && cq < C4) {
        double a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0};
        for (int l = 0; l < RL; ++l)
#pragma unroll
            for (int j = 0; j < 4; ++j) { a0[j] += sh[0][l * QB + q][j]; a1[j] += sh[1][l * QB + q][j]; }
        double* o = out + (size_t)(blockIdx.y % BN_SLOTS) * 2 * C;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsafeAtomicAdd(o + cq * 4 + j, a0[j]);
            unsafeAtomicAdd(o + C + cq * 4 + j, a1[j]);
        }
    }
}

// rows per block so that the reduction grid has about 512 blocks (a multiple of the row-lane count, >= 2 rows per lane)
static inline int bn_rows_per_block(long long M, int C) {
    const int C4 = C / 4, QB = C4 < 64 ? C4 : 64, RL = 256 / QB, CG = cdiv(C4, 64);
    long long r = cdiv(M * CG, 512);
    if (r < 2 * RL) r = 2 * RL;
    r = cdiv(r, RL) * RL;
    return (int)r;
}

// sum of one column over the slot rows; with `clear` the slots are zeroed behind the read, so a persistent workspace is ready
// for the next reduction without a fill launch
__device__ __forceinline__ double fold_slots(double* __restrict__ slots, int nslots, int C, int idx, bool clear) {
    double v[4] = {0.0, 0.0, 0.0, 0.0};          // independent loads, all in flight together
    int s = 0;
#pragma unroll 2
    for (; s + 3 < nslots; s += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += slots[(size_t)(s + j) * 2 * C + idx];
    }
    for (; s < nslots; ++s) v[0] += slots[(size_t)s * 2 * C + idx];
    if (clear)
        for (s = 0; s < nslots; ++s) slots[(size_t)s * 2 * C + idx] = 0.0;
    return (v[0] + v[1]) + (v[2] + v[3]);
}

__global__ void bn_finalize_kernel(double* __restrict__ stats, int nslots, int clear, double count, const double* __restrict__ count_dev, int C,
                                   float eps, float momentum, float* mean, float* invstd, float* running_mean,
                                   float* running_var) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (count_dev) count = count_dev[0];
    const double m = fold_slots(stats, nslots, C, c, clear) / count;
    double var = fold_slots(stats, nslots, C, C + c, clear) / count - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

// block-wide max of non-negative floats into a device SLOT of VBG_AMAX_WORDS = 64 words holding bit patterns (non-negative floats order
// like their bits); the slot's value is the max over its words, which lie 128 bytes apart.  A block publishes into word blockIdx % 64:
// atomics on one L2 line serialise at ~10 ns each (2048 blocks on ONE word, or on 64 adjacent words: +16-19 us per launch, measured
// in the step); 32 per line on 64 lines do not show.
__device__ __forceinline__ void amax_publish(float mx, unsigned* amax) {
    __shared__ float amax_sh[16];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) amax_sh[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) mx = fmaxf(mx, amax_sh[w]);
        const unsigned bits = __float_as_uint(mx);
        unsigned* word = amax + (blockIdx.x & (VBG_AMAX_WORDS - 1)) * VBG_AMAX_STRIDE;
        if (bits > __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(word, bits);
    }
}

__global__ void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ res, long long M, int C4,
                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
                                float* __restrict__ y, unsigned* amax) {
    const long long total = M * C4;
    const long long stride = (long long)gridDim.x * blockDim.x;
    float mx = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c4 = (int)(i % C4);
        const float4 xv = reinterpret_cast<const float4*>(x)[i];
        const float4 mu = reinterpret_cast<const float4*>(mean)[c4];
        const float4 is = reinterpret_cast<const float4*>(invstd)[c4];
        const float4 ga = reinterpret_cast<const float4*>(gamma)[c4];
        const float4 be = reinterpret_cast<const float4*>(beta)[c4];
        float4 o;
        o.x = (xv.x - mu.x) * is.x * ga.x + be.x; o.y = (xv.y - mu.y) * is.y * ga.y + be.y;
        o.z = (xv.z - mu.z) * is.z * ga.z + be.z; o.w = (xv.w - mu.w) * is.w * ga.w + be.w;
        if (res) {
            const float4 rv = reinterpret_cast<const float4*>(res)[i];
            o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
        }
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        reinterpret_cast<float4*>(y)[i] = o;
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    }
    if (amax) amax_publish(mx, amax);                            // (uniform)
}

__global__ void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ x,
                                    long long M, int C4, int C, const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const float* __restrict__ gamma, const double* __restrict__ sums, double count,
                                    const double* __restrict__ count_dev, int relu, float* __restrict__ dx,
                                    float* __restrict__ dres, unsigned* amax) {
    if (count_dev) count = count_dev[0];
    float mx = 0.f;
    const float icnt = (float)(1.0 / count);
    const long long total = M * C4;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c4 = (int)(i % C4), c = c4 * 4;
        float4 g = reinterpret_cast<const float4*>(dy)[i];
        if (relu) {
            const float4 yv = reinterpret_cast<const float4*>(y)[i];
            g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f; g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
        }
        const float4 xv = reinterpret_cast<const float4*>(x)[i];
        const float4 mu = reinterpret_cast<const float4*>(mean)[c4], is = reinterpret_cast<const float4*>(invstd)[c4];
        const float4 ga = reinterpret_cast<const float4*>(gamma)[c4];
        float4 o;
        o.x = ga.x * is.x * (g.x - (float)sums[c + 0] * icnt - (xv.x - mu.x) * is.x * ((float)sums[C + c + 0] * icnt));
        o.y = ga.y * is.y * (g.y - (float)sums[c + 1] * icnt - (xv.y - mu.y) * is.y * ((float)sums[C + c + 1] * icnt));
        o.z = ga.z * is.z * (g.z - (float)sums[c + 2] * icnt - (xv.z - mu.z) * is.z * ((float)sums[C + c + 2] * icnt));
        o.w = ga.w * is.w * (g.w - (float)sums[c + 3] * icnt - (xv.w - mu.w) * is.w * ((float)sums[C + c + 3] * icnt));
        reinterpret_cast<float4*>(dx)[i] = o;
        if (dres) reinterpret_cast<float4*>(dres)[i] = g;
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    }
    if (amax) amax_publish(mx, amax);                            // (uniform)
}

// ---- the same two passes with the statistics FOLDED IN THEIR PROLOGUE (round 5) ----------------------------------------------------------
// bn_finalize / bn_param_grad are 7 us launches of a few hundred threads between two streaming kernels (80 per step at resnet-34: 0.56 ms of
// launch-shaped time).  Here a block owns 64 channels x a chunk of rows (16 channel quads x 16 row lanes): its first 128 threads fold the
// slot rows of their (channel, statistic) pair -- 32 independent L2 loads, summed in fold_slots' order, so mean / invstd / the sums are the
// bits the separate launches produce --, the block of row chunk 0 publishes mean / invstd / running statistics (forward) or adds the affine
// gradients (backward), and everybody streams its rows with the per-channel constants in registers.  The fold costs a block 32 KB of L2
// reads, so a block takes >= 64 rows (>= 32 KB of its own traffic) -- small maps launch few blocks, which they can afford.  The slot
// rows are NOT cleared (other blocks are still reading them): the caller hands in zeroed rows from a pool (vbg.ops.bn_zero_slots).
__device__ __forceinline__ void amax_publish_at(float mx, unsigned* amax, unsigned block_id) {
    __shared__ float amax_sh2[16];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) amax_sh2[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) mx = fmaxf(mx, amax_sh2[w]);
        const unsigned bits = __float_as_uint(mx);
        unsigned* word = amax + (block_id & (VBG_AMAX_WORDS - 1)) * VBG_AMAX_STRIDE;
        if (bits > __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(word, bits);
    }
}

__global__ __launch_bounds__(256) void bn_apply_fold_kernel(const float* __restrict__ x, const float* __restrict__ res, long long M, int C,
                                                            double* __restrict__ slots, int nslots, double count,
                                                            const double* __restrict__ count_dev, float eps, float momentum,
                                                            float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                            float* __restrict__ running_mean, float* __restrict__ running_var,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
                                                            int rows_per_block, float* __restrict__ y, unsigned* amax) {
    __shared__ double fold[128];
    __shared__ __attribute__((aligned(16))) float cst[4][64];          // mean, invstd, gamma, beta of the block's 64 channels
    const int tid = threadIdx.x, c0 = blockIdx.y * 64;
    if (count_dev) count = count_dev[0];                 // (SyncBatchNorm: the all-reduced row count, nslots = 1: the all-reduced sums)
    if (tid < 128) fold[tid] = fold_slots(slots, nslots, C, (tid >> 6) * C + c0 + (tid & 63), false);
    __syncthreads();
    if (tid < 64) {
        const int c = c0 + tid;
        const double m = fold[tid] / count;
        double var = fold[64 + tid] / count - m * m;
        if (var < 0.0) var = 0.0;
        const float mf = (float)m, isf = (float)(1.0 / sqrt(var + (double)eps));
        cst[0][tid] = mf; cst[1][tid] = isf; cst[2][tid] = gamma[c]; cst[3][tid] = beta[c];
        if (blockIdx.x == 0) {
            mean_out[c] = mf;
            invstd_out[c] = isf;
            if (running_mean) {
                const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mf;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
            }
        }
    }
    __syncthreads();
    const int q = tid & 15, rl = tid >> 4, C4 = C >> 2, cq = blockIdx.y * 16 + q;
    const float4 mu = reinterpret_cast<const float4*>(cst[0])[q], is = reinterpret_cast<const float4*>(cst[1])[q];
    const float4 ga = reinterpret_cast<const float4*>(cst[2])[q], be = reinterpret_cast<const float4*>(cst[3])[q];
    const long long r0 = (long long)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float mx = 0.f;
#pragma unroll 4
    for (long long r = r0 + rl; r < r1; r += 16) {
        const long long i = r * C4 + cq;
        const float4 xv = reinterpret_cast<const float4*>(x)[i];
        float4 o;
        o.x = (xv.x - mu.x) * is.x * ga.x + be.x; o.y = (xv.y - mu.y) * is.y * ga.y + be.y;
        o.z = (xv.z - mu.z) * is.z * ga.z + be.z; o.w = (xv.w - mu.w) * is.w * ga.w + be.w;
        if (res) {
            const float4 rv = reinterpret_cast<const float4*>(res)[i];
            o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
        }
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        reinterpret_cast<float4*>(y)[i] = o;
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    }
    if (amax) amax_publish_at(mx, amax, blockIdx.x + blockIdx.y * gridDim.x);                  // (uniform)
}

__global__ __launch_bounds__(256) void bn_bwd_apply_fold_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                const float* __restrict__ x, long long M, int C,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma, double* __restrict__ slots, int nslots,
                                                                double count, int relu, int rows_per_block, float* __restrict__ dx,
                                                                float* __restrict__ dres, float* dgamma, float* dbeta, unsigned* amax) {
    __shared__ double fold[128];
    __shared__ __attribute__((aligned(16))) float cst[5][64];          // mean, invstd, gamma, sum g / n, sum g xhat / n
    const int tid = threadIdx.x, c0 = blockIdx.y * 64;
    if (tid < 128) fold[tid] = fold_slots(slots, nslots, C, (tid >> 6) * C + c0 + (tid & 63), false);
    __syncthreads();
    const float icnt = (float)(1.0 / count);
    if (tid < 64) {
        const int c = c0 + tid;
        const double sg = fold[tid], sgx = fold[64 + tid];
        cst[0][tid] = mean[c]; cst[1][tid] = invstd[c]; cst[2][tid] = gamma[c];
        cst[3][tid] = (float)sg * icnt; cst[4][tid] = (float)sgx * icnt;
        if (blockIdx.x == 0) {
            if (dbeta) dbeta[c] += (float)sg;
            if (dgamma) dgamma[c] += (float)sgx;
        }
    }
    __syncthreads();
    const int q = tid & 15, rl = tid >> 4, C4 = C >> 2, cq = blockIdx.y * 16 + q;
    const float4 mu = reinterpret_cast<const float4*>(cst[0])[q], is = reinterpret_cast<const float4*>(cst[1])[q];
    const float4 ga = reinterpret_cast<const float4*>(cst[2])[q], s0 = reinterpret_cast<const float4*>(cst[3])[q];
    const float4 s1 = reinterpret_cast<const float4*>(cst[4])[q];
    const long long r0 = (long long)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float mx = 0.f;
#pragma unroll 4
    for (long long r = r0 + rl; r < r1; r += 16) {
        const long long i = r * C4 + cq;
        float4 g = reinterpret_cast<const float4*>(dy)[i];
        if (relu) {
            const float4 yv = reinterpret_cast<const float4*>(y)[i];
            g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f; g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
        }
        const float4 xv = reinterpret_cast<const float4*>(x)[i];
        float4 o;
        o.x = ga.x * is.x * (g.x - s0.x - (xv.x - mu.x) * is.x * s1.x);
        o.y = ga.y * is.y * (g.y - s0.y - (xv.y - mu.y) * is.y * s1.y);
        o.z = ga.z * is.z * (g.z - s0.z - (xv.z - mu.z) * is.z * s1.z);
        o.w = ga.w * is.w * (g.w - s0.w - (xv.w - mu.w) * is.w * s1.w);
        reinterpret_cast<float4*>(dx)[i] = o;
        if (dres) reinterpret_cast<float4*>(dres)[i] = g;
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    }
    if (amax) amax_publish_at(mx, amax, blockIdx.x + blockIdx.y * gridDim.x);                  // (uniform)
}

// amax[0] = max(amax[0], bits of max |x|)
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, long long n, unsigned* amax) {
    const long long n4 = n / 4, stride = (long long)gridDim.x * blockDim.x;
    float mx = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - n4 * 4)) mx = fmaxf(mx, fabsf(x[n4 * 4 + threadIdx.x]));
    amax_publish(mx, amax);
}

__global__ void bn_param_grad_kernel(double* __restrict__ slots, int nslots, int clear, int C, double* __restrict__ folded, float* dgamma,
                                     float* dbeta, double tail = -1.0) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && folded && tail >= 0.0) folded[2 * C] = tail;        // (the local row count behind the sums: one buffer, one all-reduce)
    if (c >= C) return;
    const double sg = fold_slots(slots, nslots, C, c, clear), sgx = fold_slots(slots, nslots, C, C + c, clear);
    if (folded) { folded[c] = sg; folded[C + c] = sgx; }
    if (dbeta) dbeta[c] += (float)sg;
    if (dgamma) dgamma[c] += (float)sgx;
}

// ------------------------------------------------------------------------------------------
__global__ void maxpool_fwd_kernel(const float* __restrict__ x, int B, int H, int W, int C, int Ho, int Wo,
                                   float* __restrict__ y, int* __restrict__ argmax) {
    const long long total = (long long)B * Ho * Wo * C;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i % C);
        long long t = i / C;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const int b = (int)(t / Ho);
        float best = -INFINITY;
        int bi = -1;
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = oy * 2 - 1 + dy;
            if (iy < 0 || iy >= H) continue;
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = ox * 2 - 1 + dx;
                if (ix < 0 || ix >= W) continue;
                const float v = x[(((long long)b * H + iy) * W + ix) * C + c];
                if (v > best || bi < 0) { best = v; bi = iy * W + ix; }
            }
        }
        y[i] = best;
        argmax[i] = bi;
    }
}

// gather form: an input pixel belongs to at most 2 x 2 windows of the 3x3 / stride 2 / pad 1 pooling (rows floor(y/2) and, for odd
// y, floor(y/2) + 1); it receives the gradient of every window whose recorded argmax it is.  No atomics, every dx element is
// written exactly once (the destination needs no zero fill).
__global__ void maxpool_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ argmax, int B, int Ho, int Wo, int C4, int H,
                                   int W, float* __restrict__ dx) {
    const long long total = (long long)B * H * W * C4;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c4 = (int)(i % C4);
        long long t = i / C4;
        const int x = (int)(t % W); t /= W;
        const int y = (int)(t % H);
        const int b = (int)(t / H);
        const int me = y * W + x;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int oy = y >> 1; oy <= ((y + 1) >> 1); ++oy) {
            if (oy >= Ho) continue;
            for (int ox = x >> 1; ox <= ((x + 1) >> 1); ++ox) {
                if (ox >= Wo) continue;
                const long long o = (((long long)b * Ho + oy) * Wo + ox) * C4 + c4;
                const int4 am = reinterpret_cast<const int4*>(argmax)[o];
                const float4 d = reinterpret_cast<const float4*>(dy)[o];
                g.x += am.x == me ? d.x : 0.f; g.y += am.y == me ? d.y : 0.f;
                g.z += am.z == me ? d.z : 0.f; g.w += am.w == me ? d.w : 0.f;
            }
        }
        reinterpret_cast<float4*>(dx)[i] = g;
    }
}

__global__ void upsample2_add_kernel(const float* __restrict__ lo, const float* __restrict__ skip, int B, int H, int W, int C4,
                                     float* __restrict__ y) {
    const long long total = (long long)B * H * W * C4;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c4 = (int)(i % C4);
        long long t = i / C4;
        const int x = (int)(t % W); t /= W;
        const int yy = (int)(t % H);
        const int b = (int)(t / H);
        const float4 a = reinterpret_cast<const float4*>(lo)[(((long long)b * (H / 2) + (yy >> 1)) * (W / 2) + (x >> 1)) * C4 + c4];
        const float4 s = reinterpret_cast<const float4*>(skip)[i];
        reinterpret_cast<float4*>(y)[i] = make_float4(a.x + s.x, a.y + s.y, a.z + s.z, a.w + s.w);
    }
}

// AvgPool2d(2, 2) of the ResNet-D projection shortcut (model/ResNetFPN_ViBERTgrid.py:224): floor output size, odd trailing
// row / column dropped; BWD spreads dy/4 over the 2x2 window (zero for a dropped trailing row / column)
template <bool BWD>
__global__ void avgpool2_kernel(const float* __restrict__ src, int B, int H, int W, int C4, float* __restrict__ dst) {
    const int Ho = H / 2, Wo = W / 2;
    const long long total = (long long)B * (BWD ? H : Ho) * (BWD ? W : Wo) * C4;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c4 = (int)(i % C4);
        long long t = i / C4;
        if (!BWD) {
            const int x = (int)(t % Wo); t /= Wo;
            const int y = (int)(t % Ho);
            const int b = (int)(t / Ho);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const float4 v = reinterpret_cast<const float4*>(src)[(((long long)b * H + 2 * y + dy) * W + 2 * x + dx) * C4 + c4];
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                }
            reinterpret_cast<float4*>(dst)[i] = make_float4(acc.x * 0.25f, acc.y * 0.25f, acc.z * 0.25f, acc.w * 0.25f);
        } else {
            const int x = (int)(t % W); t /= W;
            const int y = (int)(t % H);
            const int b = (int)(t / H);
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((y >> 1) < Ho && (x >> 1) < Wo) {
                g = reinterpret_cast<const float4*>(src)[(((long long)b * Ho + (y >> 1)) * Wo + (x >> 1)) * C4 + c4];
                g.x *= 0.25f; g.y *= 0.25f; g.z *= 0.25f; g.w *= 0.25f;
            }
            reinterpret_cast<float4*>(dst)[i] = g;
        }
    }
}

__global__ void sumpool_kernel(const float* __restrict__ hi, int B, int H, int W, int C4, int f, float* lo, int accumulate) {
    const int Hl = H / f, Wl = W / f;
    const long long total = (long long)B * Hl * Wl * C4;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c4 = (int)(i % C4);
        long long t = i / C4;
        const int x = (int)(t % Wl); t /= Wl;
        const int y = (int)(t % Hl);
        const int b = (int)(t / Hl);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int dy = 0; dy < f; ++dy)
            for (int dx = 0; dx < f; ++dx) {
                const float4 v = reinterpret_cast<const float4*>(hi)[(((long long)b * H + y * f + dy) * W + x * f + dx) * C4 + c4];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        if (accumulate) {
            const float4 o = reinterpret_cast<float4*>(lo)[i];
            acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
        }
        reinterpret_cast<float4*>(lo)[i] = acc;
    }
}

// [B, R, Cc] -> [B, Cc, R] through a padded 32x32 LDS tile (both sides coalesced)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ x, int R, int Cc, float* __restrict__ y) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* xb = x + (long long)b * R * Cc;
    float* yb = y + (long long)b * R * Cc;
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        tile[j][tx] = (r < R && c < Cc) ? xb[(long long)r * Cc + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;
        if (r < R && c < Cc) yb[(long long)c * R + r] = tile[tx][j];
    }
}

__global__ void upsample_nhwc_to_nchw_kernel(const float* __restrict__ x, int B, int h, int w, int C, int f, float* __restrict__ y) {
    const int H = h * f, W = w * f;
    const long long total = (long long)B * C * H * W;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int X = (int)(i % W);
        long long t = i / W;
        const int Y = (int)(t % H); t /= H;
        const int c = (int)(t % C);
        const int b = (int)(t / C);
        y[i] = x[(((long long)b * h + Y / f) * w + X / f) * C + c];
    }
}

// ------------------------------------------------------------------------------------------
// input transform: (img - mean) / std, bilinear resize (align_corners=False, scale = in/out),
// written NHWC into batch slot b.  One thread per output pixel (3 channels).
// ------------------------------------------------------------------------------------------
__global__ void normalize_resize_kernel(const float* __restrict__ img, int h, int w, int oh, int ow, float m0, float m1, float m2,
                                        float s0, float s1, float s2, float* __restrict__ out, int W) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= oh * ow) return;
    const int oy = i / ow, ox = i - oy * ow;
    const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
    float v[3];
    if (oh == h && ow == w) {
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = (img[((long long)c * h + oy) * w + ox] - mean[c]) / sd[c];
    } else {
        const float sy = (float)h / (float)oh, sx = (float)w / (float)ow;
        float fy = ((float)oy + 0.5f) * sy - 0.5f; if (fy < 0.f) fy = 0.f;
        float fx = ((float)ox + 0.5f) * sx - 0.5f; if (fx < 0.f) fx = 0.f;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + ((y0 < h - 1) ? 1 : 0), x1 = x0 + ((x0 < w - 1) ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* p = img + (long long)c * h * w;
            const float a = (p[(long long)y0 * w + x0] - mean[c]) / sd[c], b = (p[(long long)y0 * w + x1] - mean[c]) / sd[c];
            const float d = (p[(long long)y1 * w + x0] - mean[c]) / sd[c], e = (p[(long long)y1 * w + x1] - mean[c]) / sd[c];
            v[c] = hy * (hx * a + lx * b) + ly * (hx * d + lx * e);
        }
    }
    float* o = out + ((long long)oy * W + ox) * 3;
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
}

__global__ void rescale_boxes_kernel(const long long* __restrict__ in, int n, float rh, float rw, int* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float r = ((i & 1) == 0) ? rh : rw;       // cols 0,2 <- height ratio; 1,3 <- width ratio (reference swap)
    out[i] = (int)__fmul_rn((float)in[i], r);
}

__global__ void im2col_kernel(const float* __restrict__ x, int B, int H, int W, int C, int kh, int kw, int stride, int pad,
                              int Ho, int Wo, int Kpad, float* __restrict__ out) {
    // one thread per (output pixel, 4 consecutive k): the pixel is decoded once, the row is written as float4 (Kpad % 4 == 0)
    const int KQ = Kpad >> 2, K = kh * kw * C;
    const long long total = (long long)B * Ho * Wo * KQ;
    const long long gstride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
        const long long pix = i / KQ;
        const int k0 = (int)(i - pix * KQ) * 4;
        const int p = (int)pix;                          // < 2^31 output pixels
        const int ox = p % Wo, t = p / Wo, oy = t % Ho, b = t / Ho;
        const float* img = x + (long long)b * H * W * C;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + j;
            v[j] = 0.f;
            if (k < K) {
                const int tap = k / C, c = k - tap * C;
                const int dy = tap / kw, dx = tap - dy * kw;
                const int iy = oy * stride - pad + dy, ix = ox * stride - pad + dx;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) v[j] = img[(iy * W + ix) * C + c];
            }
        }
        reinterpret_cast<float4*>(out)[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

}  // namespace vbg

using namespace vbg;
#define S_ ((hipStream_t)stream)
#define ALIGNED16(p) (((uintptr_t)(p)) % 16 == 0)

extern "C" int vbg_bn_stats(const float* x, long long M, int C, double* stats_accum, void* stream) {
    VBG_CHECK_ARG(x && stats_accum && M >= 0 && C > 0 && C % 4 == 0 && ALIGNED16(x));
    VBG_CHECK_ARG(C / 4 <= 64 ? (256 % (C / 4) == 0) : (C / 4) % 64 == 0);
    if (M == 0) return VBG_OK;
    const int rpb = bn_rows_per_block(M, C);
    VBG_LAUNCH((bn_reduce_kernel<false>), dim3(cdiv(C / 4, 64), cdiv(M, rpb)), dim3(256), 0, S_, x, nullptr, nullptr, M, C, nullptr,
               nullptr, 0, rpb, stats_accum);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_bn_slots(void) { return BN_SLOTS; }

extern "C" int vbg_bn_finalize(double* stats, int nslots, int clear_slots, double count, const double* count_dev, int C, float eps, float momentum,
                               float* mean, float* invstd, float* running_mean, float* running_var, void* stream) {
    VBG_CHECK_ARG(stats && nslots >= 1 && mean && invstd && C > 0 && (count > 0 || count_dev) &&
                  ((running_mean == nullptr) == (running_var == nullptr)));
    VBG_LAUNCH(bn_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0, S_, stats, nslots, clear_slots, count, count_dev, C, eps, momentum, mean,
                       invstd, running_mean, running_var);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_bn_apply(const float* x, const float* res, long long M, int C, const float* mean, const float* invstd,
                            const float* gamma, const float* beta, int relu, float* y, unsigned* y_amax, void* stream) {
    VBG_CHECK_ARG(x && mean && invstd && gamma && beta && y && M >= 0 && C > 0 && C % 4 == 0);
    VBG_CHECK_ARG(ALIGNED16(x) && ALIGNED16(y) && ALIGNED16(mean) && ALIGNED16(invstd) && ALIGNED16(gamma) && ALIGNED16(beta) &&
                  (!res || ALIGNED16(res)));
    if (M == 0) return VBG_OK;
    VBG_LAUNCH(bn_apply_kernel, dim3(ew_grid(M * (C / 4), 256)), dim3(256), 0, S_, x, res, M, C / 4, mean, invstd, gamma,
                       beta, relu, y, y_amax);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_bn_bwd_reduce(const float* dy, const float* y, const float* x, long long M, int C, const float* mean,
                                 const float* invstd, int relu, double* sums_accum, void* stream) {
    VBG_CHECK_ARG(dy && x && mean && invstd && sums_accum && M >= 0 && C > 0 && (!relu || y) && C % 4 == 0);
    VBG_CHECK_ARG(ALIGNED16(dy) && ALIGNED16(x) && ALIGNED16(mean) && ALIGNED16(invstd) && (!relu || ALIGNED16(y)));
    VBG_CHECK_ARG(C / 4 <= 64 ? (256 % (C / 4) == 0) : (C / 4) % 64 == 0);
    if (M == 0) return VBG_OK;
    const int rpb = bn_rows_per_block(M, C);
    VBG_LAUNCH((bn_reduce_kernel<true>), dim3(cdiv(C / 4, 64), cdiv(M, rpb)), dim3(256), 0, S_, dy, y, x, M, C, mean, invstd, relu,
               rpb, sums_accum);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_bn_bwd_apply(const float* dy, const float* y, const float* x, long long M, int C, const float* mean,
                                const float* invstd, const float* gamma, const double* sums, double count,
                                const double* count_dev, int relu, float* dx, float* dres, float* dgamma_accum,
                                float* dbeta_accum, unsigned* dx_amax, void* stream) {
    VBG_CHECK_ARG(dy && x && mean && invstd && gamma && sums && dx && M >= 0 && C > 0 && (count > 0 || count_dev) && (!relu || y));
    VBG_CHECK_ARG(C % 4 == 0 && ALIGNED16(dy) && ALIGNED16(x) && ALIGNED16(dx) && ALIGNED16(mean) && ALIGNED16(invstd) &&
                  ALIGNED16(gamma) && (!relu || ALIGNED16(y)) && (!dres || ALIGNED16(dres)));
    if (M > 0) VBG_LAUNCH(bn_bwd_apply_kernel, dim3(ew_grid(M * (C / 4), 256)), dim3(256), 0, S_, dy, y, x, M, C / 4, C, mean, invstd,
                          gamma, sums, count, count_dev, relu, dx, dres, dx_amax);
    if (dgamma_accum && dbeta_accum)
        VBG_LAUNCH(bn_param_grad_kernel, dim3(cdiv(C, 256)), dim3(256), 0, S_, const_cast<double*>(sums), 1, 0, C, (double*)nullptr, dgamma_accum, dbeta_accum, -1.0);
    VBG_LAUNCH_RET();
}

// rows per block of the folding kernels: >= 64 (a block's fold reads 32 KB), about 1536 blocks on the large maps
static inline int bn_fold_rows(long long M, int C) {
    long long r = cdiv(M * (C / 64), 1536);
    if (r < 64) r = 64;
    return (int)(cdiv(r, 16) * 16);
}

extern "C" int vbg_bn_apply_fold(const float* x, const float* res, long long M, int C, double* slots, int nslots, double count,
                                 const double* count_dev, float eps, float momentum, float* mean, float* invstd, float* running_mean,
                                 float* running_var, const float* gamma, const float* beta, int relu, float* y, unsigned* y_amax, void* stream) {
    VBG_CHECK_ARG(x && y && slots && mean && invstd && gamma && beta && M > 0 && C > 0 && C % 64 == 0 && nslots >= 1 && (count > 0 || count_dev));
    VBG_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr) && ALIGNED16(x) && ALIGNED16(y) && (!res || ALIGNED16(res)));
    const int rpb = bn_fold_rows(M, C);
    VBG_LAUNCH(bn_apply_fold_kernel, dim3(cdiv(M, rpb), C / 64), dim3(256), 0, S_, x, res, M, C, slots, nslots, count, count_dev, eps, momentum, mean, invstd,
               running_mean, running_var, gamma, beta, relu, rpb, y, y_amax);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_bn_bwd_apply_fold(const float* dy, const float* y, const float* x, long long M, int C, const float* mean,
                                     const float* invstd, const float* gamma, double* slots, int nslots, double count, int relu, float* dx,
                                     float* dres, float* dgamma_accum, float* dbeta_accum, unsigned* dx_amax, void* stream) {
    VBG_CHECK_ARG(dy && x && mean && invstd && gamma && slots && dx && M > 0 && C > 0 && C % 64 == 0 && nslots >= 1 && count > 0 && (!relu || y));
    VBG_CHECK_ARG(((dgamma_accum == nullptr) == (dbeta_accum == nullptr)) && ALIGNED16(dy) && ALIGNED16(x) && ALIGNED16(dx) &&
                  (!relu || ALIGNED16(y)) && (!dres || ALIGNED16(dres)));
    const int rpb = bn_fold_rows(M, C);
    VBG_LAUNCH(bn_bwd_apply_fold_kernel, dim3(cdiv(M, rpb), C / 64), dim3(256), 0, S_, dy, y, x, M, C, mean, invstd, gamma, slots, nslots, count,
               relu, rpb, dx, dres, dgamma_accum, dbeta_accum, dx_amax);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_amax(const float* x, long long n, unsigned* amax, void* stream) {
    VBG_CHECK_ARG(n >= 0 && amax);
    if (n == 0) return VBG_OK;
    VBG_CHECK_ARG(x && ALIGNED16(x));
    VBG_LAUNCH(amax_kernel, dim3(ew_grid((n + 3) / 4, 256 * 8)), dim3(256), 0, S_, x, n, amax);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_bn_param_grad(double* slots, int nslots, int clear_slots, int C, double* folded, float* dgamma_accum,
                                 float* dbeta_accum, void* stream) {
    VBG_CHECK_ARG(slots && nslots >= 1 && C > 0 && ((dgamma_accum == nullptr) == (dbeta_accum == nullptr)) && (folded || dgamma_accum));
    VBG_LAUNCH(bn_param_grad_kernel, dim3(cdiv(C, 256)), dim3(256), 0, S_, slots, nslots, clear_slots, C, folded, dgamma_accum, dbeta_accum, -1.0);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_bn_fold_count(double* slots, int nslots, int clear_slots, int C, double* folded, double count, void* stream) {
    VBG_CHECK_ARG(slots && folded && nslots >= 1 && C > 0 && count >= 0);
    VBG_LAUNCH(bn_param_grad_kernel, dim3(cdiv(C, 256)), dim3(256), 0, S_, slots, nslots, clear_slots, C, folded, (float*)nullptr, (float*)nullptr, count);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_maxpool3x3s2_fwd(const float* x, int B, int H, int W, int C, float* y, int* argmax, void* stream) {
    VBG_CHECK_ARG(x && y && argmax && B >= 0 && H > 0 && W > 0 && C > 0);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)B * Ho * Wo * C;
    if (total == 0) return VBG_OK;
    VBG_LAUNCH(maxpool_fwd_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, S_, x, B, H, W, C, Ho, Wo, y, argmax);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_maxpool3x3s2_bwd(const float* dy, const int* argmax, int B, int Ho, int Wo, int C, int H, int W,
                                    float* dx_zeroed, void* stream) {
    VBG_CHECK_ARG(dy && argmax && dx_zeroed && B >= 0 && Ho > 0 && Wo > 0 && C > 0 && C % 4 == 0);
    VBG_CHECK_ARG(ALIGNED16(dy) && ALIGNED16(argmax) && ALIGNED16(dx_zeroed) && Ho == (H + 2 - 3) / 2 + 1 && Wo == (W + 2 - 3) / 2 + 1);
    const long long total = (long long)B * Ho * Wo * C;
    if (total == 0) return VBG_OK;
    VBG_LAUNCH(maxpool_bwd_kernel, dim3(ew_grid((long long)B * H * W * (C / 4), 256)), dim3(256), 0, S_, dy, argmax, B, Ho, Wo, C / 4, H, W,
               dx_zeroed);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_upsample2_add(const float* lo, const float* skip, int B, int H, int W, int C, float* y, void* stream) {
    VBG_CHECK_ARG(lo && skip && y && H % 2 == 0 && W % 2 == 0 && C % 4 == 0 && ALIGNED16(lo) && ALIGNED16(skip) && ALIGNED16(y));
    const long long total = (long long)B * H * W * (C / 4);
    if (total == 0) return VBG_OK;
    VBG_LAUNCH(upsample2_add_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, S_, lo, skip, B, H, W, C / 4, y);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_avgpool2_fwd(const float* x, int B, int H, int W, int C, float* y, void* stream) {
    VBG_CHECK_ARG(x && y && B >= 0 && H >= 2 && W >= 2 && C > 0 && C % 4 == 0 && ALIGNED16(x) && ALIGNED16(y));
    const long long total = (long long)B * (H / 2) * (W / 2) * (C / 4);
    if (total == 0) return VBG_OK;
    VBG_LAUNCH((avgpool2_kernel<false>), dim3(ew_grid(total, 256)), dim3(256), 0, S_, x, B, H, W, C / 4, y);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_avgpool2_bwd(const float* dy, int B, int H, int W, int C, float* dx, void* stream) {
    VBG_CHECK_ARG(dy && dx && B >= 0 && H >= 2 && W >= 2 && C > 0 && C % 4 == 0 && ALIGNED16(dy) && ALIGNED16(dx));
    const long long total = (long long)B * H * W * (C / 4);
    if (total == 0) return VBG_OK;
    VBG_LAUNCH((avgpool2_kernel<true>), dim3(ew_grid(total, 256)), dim3(256), 0, S_, dy, B, H, W, C / 4, dx);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_sumpool(const float* hi, int B, int H, int W, int C, int f, float* lo, int accumulate, void* stream) {
    VBG_CHECK_ARG(hi && lo && f >= 1 && H % f == 0 && W % f == 0 && C % 4 == 0 && ALIGNED16(hi) && ALIGNED16(lo));
    const long long total = (long long)B * (H / f) * (W / f) * (C / 4);
    if (total == 0) return VBG_OK;
    VBG_LAUNCH(sumpool_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, S_, hi, B, H, W, C / 4, f, lo, accumulate);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_nchw_to_nhwc(const float* x, int B, int C, int HW, float* y, void* stream) {
    VBG_CHECK_ARG(x && y && B >= 0 && C > 0 && HW > 0);
    if (B == 0) return VBG_OK;
    VBG_LAUNCH(transpose_kernel, dim3(cdiv(HW, 32), cdiv(C, 32), B), dim3(256), 0, S_, x, C, HW, y);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_nhwc_to_nchw(const float* x, int B, int C, int HW, float* y, void* stream) {
    VBG_CHECK_ARG(x && y && B >= 0 && C > 0 && HW > 0);
    if (B == 0) return VBG_OK;
    VBG_LAUNCH(transpose_kernel, dim3(cdiv(C, 32), cdiv(HW, 32), B), dim3(256), 0, S_, x, HW, C, y);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_upsample_nhwc_to_nchw(const float* x, int B, int h, int w, int C, int f, float* y, void* stream) {
    VBG_CHECK_ARG(x && y && f >= 1 && C > 0);
    const long long total = (long long)B * C * h * f * w * f;
    if (total == 0) return VBG_OK;
    VBG_LAUNCH(upsample_nhwc_to_nchw_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, S_, x, B, h, w, C, f, y);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_normalize_resize(const float* img, int h, int w, int oh, int ow, const float* h_mean3, const float* h_std3,
                                    float* batch_nhwc, int b, int H, int W, void* stream) {
    VBG_CHECK_ARG(img && h_mean3 && h_std3 && batch_nhwc && h > 0 && w > 0 && oh > 0 && ow > 0 && oh <= H && ow <= W && b >= 0);
    float* out = batch_nhwc + (long long)b * H * W * 3;
    VBG_LAUNCH(normalize_resize_kernel, dim3(cdiv((long)oh * ow, 256)), dim3(256), 0, S_, img, h, w, oh, ow, h_mean3[0],
                       h_mean3[1], h_mean3[2], h_std3[0], h_std3[1], h_std3[2], out, W);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_rescale_boxes(const long long* in, int S, float ratio_h, float ratio_w, int* out, void* stream) {
    VBG_CHECK_ARG(S >= 0);
    if (S == 0) return VBG_OK;
    VBG_CHECK_ARG(in && out);
    VBG_LAUNCH(rescale_boxes_kernel, dim3(cdiv(S * 4, 256)), dim3(256), 0, S_, in, S * 4, ratio_h, ratio_w, out);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_im2col(const float* x, int B, int H, int W, int C, int kh, int kw, int stride, int pad, int Kpad, float* out,
                          void* stream) {
    VBG_CHECK_ARG(x && out && Kpad >= kh * kw * C && Kpad % 4 == 0 && stride > 0 && ALIGNED16(out));
    const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    VBG_CHECK_ARG((long long)B * Ho * Wo < (1ll << 31) && (long long)H * W * C < (1ll << 31));
    const long long total = (long long)B * Ho * Wo * (Kpad / 4);
    if (total == 0) return VBG_OK;
    VBG_LAUNCH(im2col_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, S_, x, B, H, W, C, kh, kw, stride, pad, Ho, Wo, Kpad,
                       out);
    VBG_LAUNCH_RET();
}
