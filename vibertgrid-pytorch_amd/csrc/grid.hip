// BERTgrid construction: token->segment aggregation (model/BERTgrid_generator.py:148-191), the
// bbox -> owner map and the embedding scatter (model/BERTgrid_generator.py:193-245), and the
// full-resolution label raster of the auxiliary head (model/semantic_segmentation_head.py:314-341).
//
// The reference performs S sequential slice assignments per document (last writer wins).  Here one
// pass computes, per cell, the OWNER = the last box in document order that covers it (integer math,
// bit-exact incl. python slice clipping and truncating division), and the payload is then a pure
// coalesced gather-write: HBM-bound, 12.6 MB/doc written once at 512x512.
#include "vbg_common.h"
#include "../../include/vbg.h"

namespace vbg {

// python `a[lo:hi]` clipping on an axis of length n
__host__ __device__ __forceinline__ int slice_clip(int v, int n) {
    if (v < 0) { v += n; if (v < 0) v = 0; }
    if (v > n) v = n;
    return v;
}

struct Rect { int r0, r1, c0, c1; };

__host__ __device__ __forceinline__ Rect box_rect(const int* b, int gh, int gw, int stride) {
    // int(v / stride): true division then truncation toward zero == C integer division
    Rect r;
    r.c0 = slice_clip(b[0] / stride, gw);
    r.r0 = slice_clip(b[1] / stride, gh);
    r.c1 = slice_clip(b[2] / stride, gw);
    r.r1 = slice_clip(b[3] / stride, gh);
    if (r.c1 < r.c0) r.c1 = r.c0;
    if (r.r1 < r.r0) r.r1 = r.r0;
    return r;
}

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void seg_reduce_fwd_kernel(const float* __restrict__ tok, const int* __restrict__ tok_row,
                                                             const int* __restrict__ run_start, const int* __restrict__ run_len,
                                                             int hidden, int mode, float* __restrict__ out) {
    const int s = blockIdx.x;
    const int st = run_start[s], ln = run_len[s];
    for (int c = threadIdx.x; c < hidden; c += blockDim.x) {
        float acc = tok[(long long)tok_row[st] * hidden + c];
        if (mode == 0) {
            for (int j = 1; j < ln; ++j) acc = acc + tok[(long long)tok_row[st + j] * hidden + c];   // token order
            acc = acc / (float)ln;
        }
        out[(long long)s * hidden + c] = acc;
    }
}

__global__ __launch_bounds__(256) void seg_reduce_bwd_kernel(const float* __restrict__ dout, const int* __restrict__ tok_row,
                                                             const int* __restrict__ run_start, const int* __restrict__ run_len,
                                                             int hidden, int mode, float* dtok) {
    const int s = blockIdx.x;
    const int st = run_start[s], ln = run_len[s];
    for (int c = threadIdx.x; c < hidden; c += blockDim.x) {
        const float g = dout[(long long)s * hidden + c];
        if (mode == 0) {
            const float v = g / (float)ln;
            for (int j = 0; j < ln; ++j) dtok[(long long)tok_row[st + j] * hidden + c] += v;
        } else {
            dtok[(long long)tok_row[st] * hidden + c] += g;
        }
    }
}

// ------------------------------------------------------------------------------------------
// owner map: block = 256 consecutive cells of one document; the document's rectangles are staged
// through LDS in chunks, scanned from the LAST box backwards with early exit.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void owner_map_kernel(const int* __restrict__ boxes, const int* __restrict__ box_off,
                                                        int gh, int gw, int stride, int* __restrict__ owner) {
    __shared__ Rect rects[256];
    const int b = blockIdx.y;
    const int cell = blockIdx.x * 256 + threadIdx.x;
    const int ncell = gh * gw;
    const int y = cell / gw, x = cell - y * gw;
    const int s0 = box_off[b], s1 = box_off[b + 1];
    int own = -1;
    for (int hi = s1; hi > s0; hi -= 256) {
        const int lo = max(s0, hi - 256);
        __syncthreads();
        if (lo + (int)threadIdx.x < hi) rects[threadIdx.x] = box_rect(boxes + 4 * (long long)(lo + threadIdx.x), gh, gw, stride);
        __syncthreads();
        if (own < 0 && cell < ncell) {
            for (int j = hi - lo - 1; j >= 0; --j) {
                const Rect r = rects[j];
                if (y >= r.r0 && y < r.r1 && x >= r.c0 && x < r.c1) { own = lo + j; break; }
            }
        }
    }
    if (cell < ncell) owner[(long long)b * ncell + cell] = own;
}

__global__ void grid_scatter_nhwc_kernel(const float* __restrict__ emb, const int* __restrict__ owner, long long ncell,
                                         int C4, float* __restrict__ grid) {
    const long long total = ncell * C4;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long long cell = i / C4;
        const int c4 = (int)(i - cell * C4);
        const int o = owner[cell];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (o >= 0) v = reinterpret_cast<const float4*>(emb)[(long long)o * C4 + c4];
        reinterpret_cast<float4*>(grid)[i] = v;
    }
}

__global__ void grid_scatter_nchw_kernel(const float* __restrict__ emb, const int* __restrict__ owner, int B, int C,
                                         int hw, float* __restrict__ grid) {
    const long long total = (long long)B * C * hw;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int cell = (int)(i % hw);
        const long long t = i / hw;
        const int c = (int)(t % C);
        const int b = (int)(t / C);
        const int o = owner[(long long)b * hw + cell];
        grid[i] = (o >= 0) ? emb[(long long)o * C + c] : 0.f;
    }
}

// backward: one block per box; sums dgrid over the cells of its rectangle that it still owns
__global__ __launch_bounds__(256) void grid_scatter_bwd_kernel(const float* __restrict__ dgrid, const int* __restrict__ owner,
                                                               const int* __restrict__ boxes, const int* __restrict__ box_doc,
                                                               int gh, int gw, int stride, int C, float* demb) {
    const int s = blockIdx.x;
    const int b = box_doc[s];
    const Rect r = box_rect(boxes + 4 * (long long)s, gh, gw, stride);
    const long long base = (long long)b * gh * gw;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.f;
        for (int y = r.r0; y < r.r1; ++y)
            for (int x = r.c0; x < r.c1; ++x) {
                const long long cell = base + (long long)y * gw + x;
                if (owner[cell] == s) acc += dgrid[cell * C + c];
            }
        demb[(long long)s * C + c] += acc;
    }
}

__global__ void label_raster_kernel(const int* __restrict__ owner, const int* __restrict__ seg_class, long long ncell,
                                    int* __restrict__ pos_neg, int* __restrict__ cls) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < ncell; i += stride) {
        const int o = owner[i];
        int c = 0, pn = 0;
        if (o >= 0) { c = seg_class[o]; pn = (c > 0) ? 1 : 2; }
        pos_neg[i] = pn;
        cls[i] = c;
    }
}

static inline int ew_grid(long long n, int block) {
    long long g = (n + block - 1) / block;
    if (g > 256 * 8) g = 256 * 8;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace vbg

using namespace vbg;

extern "C" int vbg_seg_reduce_fwd(const float* tok, const int* tok_row, const int* run_start, const int* run_len, int nseg,
                                  int hidden, int mode, float* out, void* stream) {
    VBG_CHECK_ARG(tok && tok_row && run_start && run_len && out && hidden > 0 && (mode == 0 || mode == 1) && nseg >= 0);
    if (nseg == 0) return VBG_OK;
    VBG_LAUNCH(seg_reduce_fwd_kernel, dim3(nseg), dim3(256), 0, (hipStream_t)stream, tok, tok_row, run_start, run_len,
                       hidden, mode, out);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_seg_reduce_bwd(const float* dout, const int* tok_row, const int* run_start, const int* run_len, int nseg,
                                  int hidden, int mode, float* dtok_accum, void* stream) {
    VBG_CHECK_ARG(dout && tok_row && run_start && run_len && dtok_accum && hidden > 0 && (mode == 0 || mode == 1) && nseg >= 0);
    if (nseg == 0) return VBG_OK;
    VBG_LAUNCH(seg_reduce_bwd_kernel, dim3(nseg), dim3(256), 0, (hipStream_t)stream, dout, tok_row, run_start, run_len,
                       hidden, mode, dtok_accum);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_owner_map(const int* boxes, const int* box_off, int B, int gh, int gw, int stride, int* owner, void* stream) {
    VBG_CHECK_ARG(box_off && owner && B >= 0 && gh >= 0 && gw >= 0 && stride > 0);
    if (B == 0 || gh == 0 || gw == 0) return VBG_OK;
    VBG_LAUNCH(owner_map_kernel, dim3(cdiv((long)gh * gw, 256), B), dim3(256), 0, (hipStream_t)stream, boxes, box_off, gh,
                       gw, stride, owner);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_grid_scatter_fwd(const float* emb, const int* owner, int B, int gh, int gw, int C, int layout, float* grid,
                                    void* stream) {
    VBG_CHECK_ARG(owner && grid && B >= 0 && gh >= 0 && gw >= 0 && C > 0 && (layout == 0 || layout == 1));
    const long long ncell = (long long)B * gh * gw;
    if (ncell == 0) return VBG_OK;
    if (layout == 0) {
        VBG_CHECK_ARG(C % 4 == 0 && ((uintptr_t)emb % 16 == 0) && ((uintptr_t)grid % 16 == 0));
        VBG_LAUNCH(grid_scatter_nhwc_kernel, dim3(ew_grid(ncell * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, emb,
                           owner, ncell, C / 4, grid);
    } else {
        VBG_LAUNCH(grid_scatter_nchw_kernel, dim3(ew_grid(ncell * C, 256)), dim3(256), 0, (hipStream_t)stream, emb, owner,
                           B, C, gh * gw, grid);
    }
    VBG_LAUNCH_RET();
}

extern "C" int vbg_grid_scatter_bwd(const float* dgrid, const int* owner, const int* boxes, const int* box_doc, int nbox, int gh,
                                    int gw, int stride, int C, float* demb_accum, void* stream) {
    VBG_CHECK_ARG(dgrid && owner && demb_accum && nbox >= 0 && gh > 0 && gw > 0 && stride > 0 && C > 0);
    if (nbox == 0) return VBG_OK;
    VBG_CHECK_ARG(boxes && box_doc);
    VBG_LAUNCH(grid_scatter_bwd_kernel, dim3(nbox), dim3(256), 0, (hipStream_t)stream, dgrid, owner, boxes, box_doc, gh, gw,
                       stride, C, demb_accum);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_label_raster(const int* owner, const int* seg_class, long long ncell, int* pos_neg, int* cls, void* stream) {
    VBG_CHECK_ARG(owner && pos_neg && cls && ncell >= 0);
    if (ncell == 0) return VBG_OK;
    VBG_LAUNCH(label_raster_kernel, dim3(ew_grid(ncell, 256)), dim3(256), 0, (hipStream_t)stream, owner, seg_class, ncell,
                       pos_neg, cls);
    VBG_LAUNCH_RET();
}
