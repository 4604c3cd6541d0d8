// Fused optimizer steps over flat fp32 ranges (torch.optim.SGD(momentum, weight_decay) for the
// CNN/head parameters and torch.optim.AdamW for BERT, as configured at train_SROIE.py:223-235 and
// stepped at pipeline/train_val_utils.py:272-284).  HBM-bound: 20 B/param (SGD-momentum),
// 28 B/param (AdamW); one launch covers a whole flat parameter bucket.
#include "vbg_common.h"
#include "../../include/vbg.h"

namespace vbg {

__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ mom, long long n4, long long n,
                           float lr, float momentum, float wd, int first, float gs) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    auto upd = [&](float& pv, float gv, float& mv) {
        float d = gv * gs + wd * pv;
        mv = first ? d : momentum * mv + d;
        pv = pv - lr * mv;
    };
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 mv = reinterpret_cast<float4*>(mom)[i];
        upd(pv.x, gv.x, mv.x); upd(pv.y, gv.y, mv.y); upd(pv.z, gv.z, mv.z); upd(pv.w, gv.w, mv.w);
        reinterpret_cast<float4*>(p)[i] = pv;
        reinterpret_cast<float4*>(mom)[i] = mv;
    }
    if (blockIdx.x == 0)
        for (long long i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) upd(p[i], g[i], mom[i]);
}

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             long long n4, long long n, float lr, float b1, float b2, float eps, float wd, float bc1,
                             float bc2_sqrt, float gs) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const float step_size = lr / bc1;
    auto upd = [&](float& pv, float gv, float& mv, float& vv) {
        gv *= gs;
        pv = pv * (1.f - lr * wd);
        mv = b1 * mv + (1.f - b1) * gv;
        vv = b2 * vv + (1.f - b2) * gv * gv;
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        pv = pv - step_size * (mv / denom);
    };
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 mv = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        upd(pv.x, gv.x, mv.x, vv.x); upd(pv.y, gv.y, mv.y, vv.y); upd(pv.z, gv.z, mv.z, vv.z); upd(pv.w, gv.w, mv.w, vv.w);
        reinterpret_cast<float4*>(p)[i] = pv;
        reinterpret_cast<float4*>(m)[i] = mv;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    if (blockIdx.x == 0)
        for (long long i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) upd(p[i], g[i], m[i], v[i]);
}

static inline int ew_grid(long long n, int block) {
    long long g = (n + block - 1) / block;
    if (g > 256 * 8) g = 256 * 8;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace vbg

using namespace vbg;
#define ALIGNED16(p) (((uintptr_t)(p)) % 16 == 0)

extern "C" int vbg_sgd_step(float* p, const float* g, float* mom, long long n, float lr, float momentum, float wd, int first_step,
                            float grad_scale, void* stream) {
    VBG_CHECK_ARG(n >= 0);
    if (n == 0) return VBG_OK;
    VBG_CHECK_ARG(p && g && mom);
    const long long n4 = (ALIGNED16(p) && ALIGNED16(g) && ALIGNED16(mom)) ? n / 4 : 0;
    VBG_LAUNCH(sgd_kernel, dim3(ew_grid(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, p, g, mom, n4, n, lr, momentum,
                       wd, first_step, grad_scale);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_adamw_step(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps,
                              float wd, int step, float grad_scale, void* stream) {
    VBG_CHECK_ARG(n >= 0 && step >= 1);
    if (n == 0) return VBG_OK;
    VBG_CHECK_ARG(p && g && m && v);
    const long long n4 = (ALIGNED16(p) && ALIGNED16(g) && ALIGNED16(m) && ALIGNED16(v)) ? n / 4 : 0;
    const double bc1 = 1.0 - pow((double)b1, (double)step), bc2 = 1.0 - pow((double)b2, (double)step);
    VBG_LAUNCH(adamw_kernel, dim3(ew_grid(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n4, n, lr, b1, b2,
                       eps, wd, (float)bc1, (float)sqrt(bc2), grad_scale);
    VBG_LAUNCH_RET();
}
