// Loss building blocks for the sampled / OHEM cross-entropy losses of the reference
// (pipeline/custom_loss.py:35-101 CrossEntropyLossRandomSample, :127-201 CrossEntropyLossOHEM):
// per-element CE forward/backward straight from low-resolution logits (the x4 nearest upsampling of
// model/semantic_segmentation_head.py:73 is folded into the row index, so the 268 MB/doc
// upsampled activation never exists), order-preserving category compaction, a STABLE descending
// radix sort (rocPRIM device primitive, <1 % of the step), gathers and sums.
// The host-RNG driven index selection (python `random.sample`) stays on the host, as in the reference.
#include "vbg_common.h"
#include "../../include/vbg.h"
#include <cstring>
#include <string.h>
#include <rocprim/rocprim.hpp>

namespace vbg {

static inline int ew_grid(long long n, int block) {
    long long g = (n + block - 1) / block;
    if (g > 256 * 8) g = 256 * 8;
    if (g < 1) g = 1;
    return (int)g;
}

__device__ __forceinline__ long long ce_row(long long e, int up_shift, int H, int W) {
    if (H <= 0) return e;
    const int x = (int)(e % W);
    const long long t = e / W;
    const int y = (int)(t % H);
    const long long b = t / H;
    return (b * (H >> up_shift) + (y >> up_shift)) * (W >> up_shift) + (x >> up_shift);
}

__global__ void ce_fwd_kernel(const float* __restrict__ logits, long long ld, int ncls, const int* __restrict__ elem,
                              const int* __restrict__ labels, long long n, const float* __restrict__ weight, int up_shift,
                              int H, int W, float* __restrict__ loss) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const long long e = elem ? elem[i] : i;
        const int t = labels[e];
        // a label outside [0, ncls) (num_classes / tag_to_idx misconfiguration) must not index logits or weights: the loss
        // becomes NaN -- visible to the caller like torch's device assert -- and backward skips the element
        if ((unsigned)t >= (unsigned)ncls) { loss[i] = __int_as_float(0x7fc00000); continue; }
        const float* x = logits + ce_row(e, up_shift, H, W) * ld;
        float mx = x[0];
        for (int c = 1; c < ncls; ++c) mx = fmaxf(mx, x[c]);
        float s = 0.f;
        for (int c = 0; c < ncls; ++c) s += expf(x[c] - mx);
        const float lp = (x[t] - mx) - logf(s);
        loss[i] = -(weight ? weight[t] : 1.f) * lp;
    }
}

__global__ void ce_bwd_kernel(const float* __restrict__ logits, long long ld, int ncls, const int* __restrict__ elem,
                              const int* __restrict__ labels, long long n, const float* __restrict__ weight,
                              const float* __restrict__ gdev, float gmul, int up_shift, int H, int W, float* dlogits) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const float g0 = gmul * (gdev ? gdev[0] : 1.f);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const long long e = elem ? elem[i] : i;
        const int t = labels[e];
        if ((unsigned)t >= (unsigned)ncls) continue;
        const long long row = ce_row(e, up_shift, H, W);
        const float* x = logits + row * ld;
        float mx = x[0];
        for (int c = 1; c < ncls; ++c) mx = fmaxf(mx, x[c]);
        float s = 0.f;
        for (int c = 0; c < ncls; ++c) s += expf(x[c] - mx);
        const float g = g0 * (weight ? weight[t] : 1.f), inv = 1.f / s;
        for (int c = 0; c < ncls; ++c) {
            const float p = expf(x[c] - mx) * inv;
            unsafeAtomicAdd(dlogits + row * ld + c, g * (p - (c == t ? 1.f : 0.f)));
        }
    }
}

struct EqPred {
    const int* labels; int value; int eq;
    __host__ __device__ bool operator()(const int i) const { return (labels[i] == value) == (eq != 0); }
};

__global__ void gather_f32_kernel(const float* __restrict__ src, const int* __restrict__ idx, long long n, float* __restrict__ out) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = src[idx[i]];
}
__global__ void gather_i32_kernel(const int* __restrict__ src, const int* __restrict__ idx, long long n, int* __restrict__ out) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = src[idx[i]];
}
__global__ void iota_kernel(int* out, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (int)i;
}
__global__ __launch_bounds__(256) void sum_kernel(const float* __restrict__ x, long long n, float* out) {
    __shared__ float sh[16];
    const long long stride = (long long)gridDim.x * blockDim.x;
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) s += x[i];
    s = block_sum(s, sh);
    if (threadIdx.x == 0) unsafeAtomicAdd(out, s);
}
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long long n, float* out) {
    __shared__ float sh[16];
    const long long stride = (long long)gridDim.x * blockDim.x;
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) s += x[i] * x[i];
    s = block_sum(s, sh);
    if (threadIdx.x == 0) unsafeAtomicAdd(out, s);
}

}  // namespace vbg

using namespace vbg;
#define S_ ((hipStream_t)stream)

extern "C" int vbg_ce_fwd(const float* logits, long long ld, int ncls, const int* elem, const int* labels, long long n,
                          const float* weight, int up_shift, int H, int W, float* loss, void* stream) {
    VBG_CHECK_ARG(n >= 0 && ncls > 0 && up_shift >= 0);
    if (n == 0) return VBG_OK;
    VBG_CHECK_ARG(logits && labels && loss);
    VBG_LAUNCH(ce_fwd_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, S_, logits, ld, ncls, elem, labels, n, weight, up_shift,
                       H, W, loss);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_ce_bwd(const float* logits, long long ld, int ncls, const int* elem, const int* labels, long long n,
                          const float* weight, const float* gscale_dev, float gmul, int up_shift, int H, int W,
                          float* dlogits_accum, void* stream) {
    VBG_CHECK_ARG(n >= 0 && ncls > 0 && up_shift >= 0);
    if (n == 0) return VBG_OK;
    VBG_CHECK_ARG(logits && labels && dlogits_accum);
    VBG_LAUNCH(ce_bwd_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, S_, logits, ld, ncls, elem, labels, n, weight,
                       gscale_dev, gmul, up_shift, H, W, dlogits_accum);
    VBG_LAUNCH_RET();
}

extern "C" long long vbg_compact_ws_bytes(long long n) {
    size_t bytes = 0;
    EqPred pred{nullptr, 0, 1};
    (void)rocprim::select(nullptr, bytes, rocprim::counting_iterator<int>(0), (int*)nullptr, (int*)nullptr, (size_t)n, pred);
    return (long long)bytes + 256;
}

extern "C" int vbg_compact(const int* labels, long long n, int value, int eq, int* out_idx, int* out_count_dev, void* ws,
                           long long ws_bytes, void* stream) {
    VBG_CHECK_ARG(n >= 0 && out_count_dev && n < 2147483647LL);
    if (n == 0) {
        hipError_t e = hipMemsetAsync(out_count_dev, 0, sizeof(int), S_);
        return e == hipSuccess ? VBG_OK : (int)e;
    }
    VBG_CHECK_ARG(labels && out_idx && ws);
    size_t bytes = (size_t)ws_bytes;
    EqPred pred{labels, value, eq};
    hipError_t e = rocprim::select(ws, bytes, rocprim::counting_iterator<int>(0), out_idx, out_count_dev, (size_t)n, pred, S_);
    return e == hipSuccess ? VBG_OK : (int)e;
}

extern "C" long long vbg_sort_ws_bytes(long long n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs_desc(nullptr, bytes, (const float*)nullptr, (float*)nullptr, (const int*)nullptr, (int*)nullptr,
                                   (size_t)n);
    return (long long)bytes + (long long)n * sizeof(int) + 512;
}

extern "C" int vbg_sort_desc(const float* keys, long long n, float* keys_out, int* idx_out, void* ws, long long ws_bytes,
                             void* stream) {
    VBG_CHECK_ARG(n >= 0 && n < 2147483647LL);
    if (n == 0) return VBG_OK;
    VBG_CHECK_ARG(keys && keys_out && idx_out && ws);
    // workspace layout: [iota int32 * n (256-aligned)] [rocprim temp]
    const size_t iota_bytes = (((size_t)n * sizeof(int)) + 255) / 256 * 256;
    VBG_CHECK_ARG((size_t)ws_bytes > iota_bytes);
    int* iota = (int*)ws;
    VBG_LAUNCH(iota_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, S_, iota, n);
    size_t bytes = (size_t)ws_bytes - iota_bytes;
    hipError_t e = rocprim::radix_sort_pairs_desc((char*)ws + iota_bytes, bytes, keys, keys_out, (const int*)iota, idx_out,
                                                  (size_t)n, 0, 32, S_);
    return e == hipSuccess ? VBG_OK : (int)e;
}

extern "C" int vbg_gather_f32(const float* src, const int* idx, long long n, float* out, void* stream) {
    VBG_CHECK_ARG(n >= 0);
    if (n == 0) return VBG_OK;
    VBG_CHECK_ARG(src && idx && out);
    VBG_LAUNCH(gather_f32_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, S_, src, idx, n, out);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_gather_i32(const int* src, const int* idx, long long n, int* out, void* stream) {
    VBG_CHECK_ARG(n >= 0);
    if (n == 0) return VBG_OK;
    VBG_CHECK_ARG(src && idx && out);
    VBG_LAUNCH(gather_i32_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, S_, src, idx, n, out);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_sum_f32(const float* x, long long n, float* out_accum, void* stream) {
    VBG_CHECK_ARG(n >= 0 && out_accum);
    if (n == 0) return VBG_OK;
    VBG_CHECK_ARG(x);
    VBG_LAUNCH(sum_kernel, dim3(ew_grid(n, 1024)), dim3(256), 0, S_, x, n, out_accum);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_sumsq(const float* g, long long n, float* out_accum, void* stream) {
    VBG_CHECK_ARG(n >= 0 && out_accum);
    if (n == 0) return VBG_OK;
    VBG_CHECK_ARG(g);
    VBG_LAUNCH(sumsq_kernel, dim3(ew_grid(n, 1024)), dim3(256), 0, S_, g, n, out_accum);
    VBG_LAUNCH_RET();
}
