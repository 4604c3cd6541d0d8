// RoIAlign on the NHWC P_fuse map (torchvision.ops.RoIAlign(output_size=7, spatial_scale=1/4,
// sampling_ratio=-1, aligned=False) called at model/grid_roi_align.py:37-41, 81).
// One block per (roi, output bin); threads run over channels so every bilinear tap is a fully
// coalesced C-wide row read of the NHWC map (HBM/L2-bound gather); the per-bin sample geometry
// is computed once per block.  Backward spreads each bin gradient to the 4 taps with atomics.
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
    VBG_LAUNCH(roi_align_bwd_kernel, dim3(out * out, nroi), dim3(C >= 256 ? 256 : (C >= 128 ? 128 : 64)), 0,
                       (hipStream_t)stream, dy, H, W, C, boxes, box_doc, out, scale, dfeat_accum);
    VBG_LAUNCH_RET();
}
