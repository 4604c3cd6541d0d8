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

// Backward without one global atomic per tap.  The 49 bins x (gh x gw) samples x 4 taps of a RoI land on the same few feature
// pixels over and over (a 72 x 24 px box covers ~20 x 8 P_fuse pixels but issues 588 tap updates per channel): a block owns
// (RoI, 64-channel slice), accumulates every tap into an LDS patch [patch pixels][64 channels] with LDS atomics (fast, no memory
// traffic), and flushes each touched patch pixel ONCE with a global atomic (RoIs overlap, so the flush still has to add): 3-5x
// fewer global atomics at the cfg2 box sizes (154 M -> ~40 M).  RoIs whose patch exceeds the LDS budget use the direct kernel.
constexpr int ROI_PATCH_MAX = 256;          // pixels
constexpr int ROI_CH = 64;

__device__ __forceinline__ void roi_patch(const int* box, float scale, int H, int W, int& y_lo, int& x_lo, int& ph, int& pw) {
    const float x1 = __fmul_rn((float)box[0], scale), y1 = __fmul_rn((float)box[1], scale);
    const float x2 = __fmul_rn((float)box[2], scale), y2 = __fmul_rn((float)box[3], scale);
    const float rw = fmaxf(__fsub_rn(x2, x1), 1.0f), rh = fmaxf(__fsub_rn(y2, y1), 1.0f);
    // every sample coordinate lies in [start, start + size]; taps are floor / floor + 1 of the coordinate clamped to the map
    y_lo = min(max((int)floorf(fmaxf(y1, 0.f)), 0), H - 1);
    x_lo = min(max((int)floorf(fmaxf(x1, 0.f)), 0), W - 1);
    const int y_hi = min(max((int)floorf(y1 + rh) + 1, 0), H - 1), x_hi = min(max((int)floorf(x1 + rw) + 1, 0), W - 1);
    ph = max(y_hi - y_lo + 1, 1);
    pw = max(x_hi - x_lo + 1, 1);
}

__global__ __launch_bounds__(256) void roi_align_bwd_patch_kernel(const float* __restrict__ dy, int H, int W, int C,
                                                                  const int* __restrict__ boxes, const int* __restrict__ box_doc,
                                                                  int out, float scale, float* dfeat) {
    __shared__ float patch[ROI_PATCH_MAX * ROI_CH];
    const int r = blockIdx.y, c0 = blockIdx.x * ROI_CH;
    const int* box = boxes + 4 * (long long)r;
    int y_lo, x_lo, ph, pw;
    roi_patch(box, scale, H, W, y_lo, x_lo, ph, pw);
    const BinGeo g = roi_geo(box, scale, out);
    float* fb = dfeat + (long long)box_doc[r] * H * W * C;
    const int cl = threadIdx.x & (ROI_CH - 1), lane_bin = threadIdx.x / ROI_CH, nbl = blockDim.x / ROI_CH;
    const int c = c0 + cl;
    const bool cv = c < C;
    if (ph * pw > ROI_PATCH_MAX) {          // (uniform) too large for the LDS patch: direct global atomics
        for (int bin = lane_bin; bin < out * out; bin += nbl) {
            const int bh = bin / out, bw = bin - bh * out;
            if (!cv) continue;
            const float gv = dy[((long long)r * out * out + bin) * C + c] * g.inv_count;
            for (int iy = 0; iy < g.gh; ++iy) {
                const float yy = sample_coord(g.y_start, bh, g.bin_h, iy, g.gh);
                for (int ix = 0; ix < g.gw; ++ix) {
                    const float xx = sample_coord(g.x_start, bw, g.bin_w, ix, g.gw);
                    Tap t;
                    if (!make_tap(yy, xx, H, W, t)) continue;
                    unsafeAtomicAdd(fb + ((long long)t.y0 * W + t.x0) * C + c, t.w00 * gv);
                    unsafeAtomicAdd(fb + ((long long)t.y0 * W + t.x1) * C + c, t.w01 * gv);
                    unsafeAtomicAdd(fb + ((long long)t.y1 * W + t.x0) * C + c, t.w10 * gv);
                    unsafeAtomicAdd(fb + ((long long)t.y1 * W + t.x1) * C + c, t.w11 * gv);
                }
            }
        }
        return;
    }
    for (int i = threadIdx.x; i < ph * pw * ROI_CH; i += blockDim.x) patch[i] = 0.f;
    __syncthreads();
    for (int bin = lane_bin; bin < out * out; bin += nbl) {
        const int bh = bin / out, bw = bin - bh * out;
        const float gv = cv ? dy[((long long)r * out * out + bin) * C + c] * g.inv_count : 0.f;
        for (int iy = 0; iy < g.gh; ++iy) {
            const float yy = sample_coord(g.y_start, bh, g.bin_h, iy, g.gh);
            for (int ix = 0; ix < g.gw; ++ix) {
                const float xx = sample_coord(g.x_start, bw, g.bin_w, ix, g.gw);
                Tap t;
                if (!make_tap(yy, xx, H, W, t)) continue;
                const int py0 = t.y0 - y_lo, py1 = t.y1 - y_lo, px0 = t.x0 - x_lo, px1 = t.x1 - x_lo;
                if ((unsigned)py0 < (unsigned)ph && (unsigned)py1 < (unsigned)ph && (unsigned)px0 < (unsigned)pw && (unsigned)px1 < (unsigned)pw) {
                    atomicAdd(&patch[(py0 * pw + px0) * ROI_CH + cl], t.w00 * gv);
                    atomicAdd(&patch[(py0 * pw + px1) * ROI_CH + cl], t.w01 * gv);
                    atomicAdd(&patch[(py1 * pw + px0) * ROI_CH + cl], t.w10 * gv);
                    atomicAdd(&patch[(py1 * pw + px1) * ROI_CH + cl], t.w11 * gv);
                } else if (cv) {            // (cannot happen by the patch bound; kept as a guard against a geometry corner case)
                    unsafeAtomicAdd(fb + ((long long)t.y0 * W + t.x0) * C + c, t.w00 * gv);
                    unsafeAtomicAdd(fb + ((long long)t.y0 * W + t.x1) * C + c, t.w01 * gv);
                    unsafeAtomicAdd(fb + ((long long)t.y1 * W + t.x0) * C + c, t.w10 * gv);
                    unsafeAtomicAdd(fb + ((long long)t.y1 * W + t.x1) * C + c, t.w11 * gv);
                }
            }
        }
    }
    __syncthreads();
    if (!cv) return;
    for (int pix = lane_bin; pix < ph * pw; pix += nbl) {
        const float v = patch[pix * ROI_CH + cl];
        if (v != 0.f) unsafeAtomicAdd(fb + ((long long)(y_lo + pix / pw) * W + (x_lo + pix % pw)) * C + c, v);
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
    VBG_LAUNCH(roi_align_bwd_patch_kernel, dim3((C + ROI_CH - 1) / ROI_CH, nroi), dim3(256), 0,
                       (hipStream_t)stream, dy, H, W, C, boxes, box_doc, out, scale, dfeat_accum);
    VBG_LAUNCH_RET();
}
