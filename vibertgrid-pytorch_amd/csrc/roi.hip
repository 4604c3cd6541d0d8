// RoIAlign on the NHWC P_fuse map (torchvision.ops.RoIAlign(output_size=7, spatial_scale=1/4,
// sampling_ratio=-1, aligned=False) called at model/grid_roi_align.py:37-41, 81).
// One block per (roi, output bin); threads run over channels so every bilinear tap is a fully
// coalesced C-wide row read of the NHWC map (HBM/L2-bound gather); the per-bin sample geometry
// is computed once per block.  Backward: separable weight tables per RoI (roi_align_bwd_sep_kernel), one atomic per patch pixel.
#include "vbg_common.h"
#include "../../include/vbg.h"

namespace vbg {

struct Tap { int y0, y1, x0, x1; float w00, w01, w10, w11; };

// torchvision bilinear_interpolate pre-calculation for one sample (fp32, no FMA contraction so the
// coordinates equal the CPU oracle's)
__device__ __forceinline__ bool make_tap(float y, float x, int H, int W, Tap& t) {
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return false;
    if (y <= 0.f) y = 0.f;
    if (x <= 0.f) x = 0.f;
    int y0 = (int)y, x0 = (int)x, y1, x1;
    if (y0 >= H - 1) { y1 = y0 = H - 1; y = (float)y0; } else y1 = y0 + 1;
    if (x0 >= W - 1) { x1 = x0 = W - 1; x = (float)x0; } else x1 = x0 + 1;
    const float ly = __fsub_rn(y, (float)y0), lx = __fsub_rn(x, (float)x0);
    const float hy = __fsub_rn(1.f, ly), hx = __fsub_rn(1.f, lx);
    t.y0 = y0; t.y1 = y1; t.x0 = x0; t.x1 = x1;
    t.w00 = __fmul_rn(hy, hx); t.w01 = __fmul_rn(hy, lx); t.w10 = __fmul_rn(ly, hx); t.w11 = __fmul_rn(ly, lx);
    return true;
}

struct BinGeo { float y_start, x_start, bin_h, bin_w; int gh, gw; float inv_count; };

__device__ __forceinline__ BinGeo roi_geo(const int* box, float scale, int out) {
    const float x1 = __fmul_rn((float)box[0], scale), y1 = __fmul_rn((float)box[1], scale);
    const float x2 = __fmul_rn((float)box[2], scale), y2 = __fmul_rn((float)box[3], scale);
    const float rw = fmaxf(__fsub_rn(x2, x1), 1.0f), rh = fmaxf(__fsub_rn(y2, y1), 1.0f);
    BinGeo g;
    g.bin_h = __fdiv_rn(rh, (float)out);
    g.bin_w = __fdiv_rn(rw, (float)out);
    g.gh = (int)ceilf(__fdiv_rn(rh, (float)out));
    g.gw = (int)ceilf(__fdiv_rn(rw, (float)out));
    g.y_start = y1; g.x_start = x1;
    const int cnt = max(g.gh * g.gw, 1);
    g.inv_count = 1.0f / (float)cnt;
    return g;
}

__device__ __forceinline__ float sample_coord(float start, int p, float bin, int i, int g) {
    // start + p*bin + (i + .5f) * bin / g
    return __fadd_rn(__fadd_rn(start, __fmul_rn((float)p, bin)), __fdiv_rn(__fmul_rn((float)i + 0.5f, bin), (float)g));
}

__global__ __launch_bounds__(256) void roi_align_fwd_kernel(const float* __restrict__ feat, int H, int W, int C,
                                                            const int* __restrict__ boxes, const int* __restrict__ box_doc,
                                                            int out, float scale, float* __restrict__ y) {
    const int r = blockIdx.y, bin = blockIdx.x;
    const int ph = bin / out, pw = bin - ph * out;
    const BinGeo g = roi_geo(boxes + 4 * (long long)r, scale, out);
    const float* fb = feat + (long long)box_doc[r] * H * W * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.f;
        for (int iy = 0; iy < g.gh; ++iy) {
            const float yy = sample_coord(g.y_start, ph, g.bin_h, iy, g.gh);
            for (int ix = 0; ix < g.gw; ++ix) {
                const float xx = sample_coord(g.x_start, pw, g.bin_w, ix, g.gw);
                Tap t;
                if (!make_tap(yy, xx, H, W, t)) continue;
                acc += t.w00 * fb[((long long)t.y0 * W + t.x0) * C + c] + t.w01 * fb[((long long)t.y0 * W + t.x1) * C + c] +
                       t.w10 * fb[((long long)t.y1 * W + t.x0) * C + c] + t.w11 * fb[((long long)t.y1 * W + t.x1) * C + c];
            }
        }
        y[((long long)r * out * out + bin) * C + c] = acc * g.inv_count;
    }
}

__global__ __launch_bounds__(256) void roi_align_bwd_kernel(const float* __restrict__ dy, int H, int W, int C,
                                                            const int* __restrict__ boxes, const int* __restrict__ box_doc,
                                                            int out, float scale, float* dfeat) {
    const int r = blockIdx.y, bin = blockIdx.x;
    const int ph = bin / out, pw = bin - ph * out;
    const BinGeo g = roi_geo(boxes + 4 * (long long)r, scale, out);
    float* fb = dfeat + (long long)box_doc[r] * H * W * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float gv = dy[((long long)r * out * out + bin) * C + c] * g.inv_count;
        for (int iy = 0; iy < g.gh; ++iy) {
            const float yy = sample_coord(g.y_start, ph, g.bin_h, iy, g.gh);
            for (int ix = 0; ix < g.gw; ++ix) {
                const float xx = sample_coord(g.x_start, pw, g.bin_w, ix, g.gw);
                Tap t;
                if (!make_tap(yy, xx, H, W, t)) continue;
                unsafeAtomicAdd(fb + ((long long)t.y0 * W + t.x0) * C + c, t.w00 * gv);
                unsafeAtomicAdd(fb + ((long long)t.y0 * W + t.x1) * C + c, t.w01 * gv);
                unsafeAtomicAdd(fb + ((long long)t.y1 * W + t.x0) * C + c, t.w10 * gv);
                unsafeAtomicAdd(fb + ((long long)t.y1 * W + t.x1) * C + c, t.w11 * gv);
            }
        }
    }
}

// Backward without one global atomic per tap.  The bilinear weights are separable -- a sample's tap weights are (hy | ly) x (hx | lx)
// and its validity is valid(y) && valid(x) -- so the gradient a RoI sends to feature pixel (Y, X) is
//     sum_bh sum_bw Wy[bh][Y] Wx[bw][X] dy[bh][bw] / count,   Wy[bh][Y] = sum over the bin's y samples of their weight on row Y,
// i.e. two small dense contractions per channel instead of 49 x (gh x gw) x 4 scattered updates: a block = one RoI, a thread = one
// channel holding the RoI's 49 bin gradients in registers; the weight tables (7 x patch rows, 7 x patch columns) are built once
// per block in LDS; every patch pixel is flushed with ONE global atomic per channel (RoIs overlap, so the flush still has to add):
// 154 M -> 42 M atomics at the cfg2 box sizes and no LDS atomics in the hot loop (an LDS-patch accumulator, the previous form of
// this kernel, lost against the direct kernel: 608 vs 329 us, LDS float atomics serialise).
constexpr int ROI_DIM_MAX = 256;            // patch rows / columns held in the weight tables
template <int OUT>
__global__ __launch_bounds__(256) void roi_align_bwd_sep_kernel(const float* __restrict__ dy, int H, int W, int C,
                                                                const int* __restrict__ boxes, const int* __restrict__ box_doc,
                                                                float scale, float* dfeat) {
    __shared__ float wy[OUT][ROI_DIM_MAX], wx[OUT][ROI_DIM_MAX];
    const int r = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    const int* box = boxes + 4 * (long long)r;
    const BinGeo g = roi_geo(box, scale, OUT);
    // rows / columns any tap of this RoI can touch: every sample coordinate lies in [start, start + size]
    const float x1 = g.x_start, y1 = g.y_start;
    const float rw = __fmul_rn(g.bin_w, (float)OUT), rh = __fmul_rn(g.bin_h, (float)OUT);
    const int y_lo = min(max((int)floorf(fmaxf(y1, 0.f)), 0), H - 1), x_lo = min(max((int)floorf(fmaxf(x1, 0.f)), 0), W - 1);
    const int y_hi = min(max((int)floorf(y1 + rh) + 2, 0), H - 1), x_hi = min(max((int)floorf(x1 + rw) + 2, 0), W - 1);
    const int ph = y_hi - y_lo + 1, pw = x_hi - x_lo + 1;              // (<= ROI_DIM_MAX: checked by the host through H, W)
    for (int i = threadIdx.x; i < OUT * ROI_DIM_MAX; i += blockDim.x) { (&wy[0][0])[i] = 0.f; (&wx[0][0])[i] = 0.f; }
    __syncthreads();
    // one thread per (bin row, y sample) and per (bin column, x sample): 1-D form of make_tap
    auto tap1 = [](float v, int size, int& i0, int& i1, float& w0, float& w1) {
        if (v < -1.0f || v > (float)size) return false;
        if (v <= 0.f) v = 0.f;
        i0 = (int)v;
        if (i0 >= size - 1) { i1 = i0 = size - 1; v = (float)i0; } else i1 = i0 + 1;
        w1 = __fsub_rn(v, (float)i0);
        w0 = __fsub_rn(1.f, w1);
        return true;
    };
    for (int i = threadIdx.x; i < OUT * g.gh; i += blockDim.x) {
        const int bh = i / g.gh, iy = i - bh * g.gh;
        int i0, i1; float w0, w1;
        if (tap1(sample_coord(g.y_start, bh, g.bin_h, iy, g.gh), H, i0, i1, w0, w1)) {
            atomicAdd(&wy[bh][min(max(i0 - y_lo, 0), ROI_DIM_MAX - 1)], w0 * g.inv_count);
            atomicAdd(&wy[bh][min(max(i1 - y_lo, 0), ROI_DIM_MAX - 1)], w1 * g.inv_count);
        }
    }
    for (int i = threadIdx.x; i < OUT * g.gw; i += blockDim.x) {
        const int bw = i / g.gw, ix = i - bw * g.gw;
        int i0, i1; float w0, w1;
        if (tap1(sample_coord(g.x_start, bw, g.bin_w, ix, g.gw), W, i0, i1, w0, w1)) {
            atomicAdd(&wx[bw][min(max(i0 - x_lo, 0), ROI_DIM_MAX - 1)], w0);
            atomicAdd(&wx[bw][min(max(i1 - x_lo, 0), ROI_DIM_MAX - 1)], w1);
        }
    }
    __syncthreads();
    if (c >= C) return;
    float gq[OUT][OUT];
#pragma unroll
    for (int bh = 0; bh < OUT; ++bh)
#pragma unroll
        for (int bw = 0; bw < OUT; ++bw) gq[bh][bw] = dy[((long long)r * OUT * OUT + bh * OUT + bw) * C + c];
    float* fb = dfeat + (long long)box_doc[r] * H * W * C + c;
    for (int Y = 0; Y < ph; ++Y) {
        float wrow[OUT];
        bool any = false;
#pragma unroll
        for (int bh = 0; bh < OUT; ++bh) { wrow[bh] = wy[bh][Y]; any |= wrow[bh] != 0.f; }
        if (!any) continue;                                            // (uniform: no sample of the RoI weighs on this row)
        float tmp[OUT];
#pragma unroll
        for (int bw = 0; bw < OUT; ++bw) {
            float a = 0.f;
#pragma unroll
            for (int bh = 0; bh < OUT; ++bh) a = fmaf(wrow[bh], gq[bh][bw], a);
            tmp[bw] = a;
        }
        float* frow = fb + (long long)(y_lo + Y) * W * C;
        for (int X = 0; X < pw; ++X) {
            float v = 0.f;
            bool anyx = false;
#pragma unroll
            for (int bw = 0; bw < OUT; ++bw) { const float w = wx[bw][X]; anyx |= w != 0.f; v = fmaf(w, tmp[bw], v); }
            if (anyx) unsafeAtomicAdd(frow + (long long)(x_lo + X) * C, v);
        }
    }
}

}  // namespace vbg

using namespace vbg;

extern "C" int vbg_roi_align_fwd(const float* feat, int B, int H, int W, int C, const int* boxes, const int* box_doc, int nroi,
                                 int out, float scale, float* y, void* stream) {
    VBG_CHECK_ARG(feat && y && B >= 0 && H > 0 && W > 0 && C > 0 && out > 0 && nroi >= 0);
    if (nroi == 0) return VBG_OK;
    VBG_CHECK_ARG(boxes && box_doc);
    VBG_LAUNCH(roi_align_fwd_kernel, dim3(out * out, nroi), dim3(C >= 256 ? 256 : (C >= 128 ? 128 : 64)), 0,
                       (hipStream_t)stream, feat, H, W, C, boxes, box_doc, out, scale, y);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_roi_align_bwd(const float* dy, int B, int H, int W, int C, const int* boxes, const int* box_doc, int nroi,
                                 int out, float scale, float* dfeat_accum, void* stream) {
    VBG_CHECK_ARG(dy && dfeat_accum && B >= 0 && H > 0 && W > 0 && C > 0 && out > 0 && nroi >= 0);
    if (nroi == 0) return VBG_OK;
    VBG_CHECK_ARG(boxes && box_doc);
    if (out == 7 && H <= ROI_DIM_MAX && W <= ROI_DIM_MAX) {
        const int nt = C >= 256 ? 256 : (C >= 128 ? 128 : 64);
        VBG_LAUNCH(roi_align_bwd_sep_kernel<7>, dim3((C + nt - 1) / nt, nroi), dim3(nt), 0, (hipStream_t)stream, dy, H, W, C, boxes, box_doc,
                   scale, dfeat_accum);
    } else {
        VBG_LAUNCH(roi_align_bwd_kernel, dim3(out * out, nroi), dim3(C >= 256 ? 256 : (C >= 128 ? 128 : 64)), 0, (hipStream_t)stream, dy, H, W,
                   C, boxes, box_doc, out, scale, dfeat_accum);
    }
    VBG_LAUNCH_RET();
}
