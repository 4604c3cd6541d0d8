// Linear-chain CRF of the `crf` classifier head (reference model/crf.py:33-157, called per document at
// model/field_type_classification_head.py:690-716) and the row gather / scatter of the two-stage `full` classifier
// (model/field_type_classification_head.py:371, 389-395: `fuse_embeddings[pred_pos_neg_mask]`).
//
// CRF: one 64-thread block per document, lane i owns tag i (ntag <= 64).  The recursions are sequential in the segment
// index (S <= a few hundred), each step is ntag x ntag flops: latency-bound by design, the point is to stay on the device
// and to batch all documents of the rank into one launch.  trans[i*ntag + j] = score of the transition j -> i.
#include "vbg_common.h"
#include "../../include/vbg.h"

namespace vbg {

constexpr int CRF_MAX_TAGS = 64;

// log-sum-exp over j of (prev[j] + w[j]) the way the reference does it (model/crf.py:24-30): max + log(sum(exp(v - max)))
__device__ __forceinline__ float lse_row(const float* prev, const float* w, int n) {
    float mx = prev[0] + w[0];
    for (int j = 1; j < n; ++j) mx = fmaxf(mx, prev[j] + w[j]);
    float s = 0.f;
    for (int j = 0; j < n; ++j) s += expf(prev[j] + w[j] - mx);
    return mx + logf(s);
}

// forward algorithm + gold path score; alpha[t*ntag + i] (emission included) is kept for the backward pass
__global__ __launch_bounds__(64) void crf_nll_fwd_kernel(const float* __restrict__ em, const int* __restrict__ tags,
                                                         const int* __restrict__ doc_off, const float* __restrict__ trans, int ntag,
                                                         int start, int stop, float* __restrict__ alpha, float* __restrict__ logz,
                                                         float* __restrict__ nll) {
    __shared__ float prev[CRF_MAX_TAGS], tr[CRF_MAX_TAGS * CRF_MAX_TAGS];
    const int d = blockIdx.x, i = threadIdx.x;
    const int r0 = doc_off[d], n = doc_off[d + 1] - r0;
    for (int k = i; k < ntag * ntag; k += 64) tr[k] = trans[k];
    if (i < ntag) prev[i] = (i == start) ? 0.f : -10000.f;
    __syncthreads();
    for (int t = 0; t < n; ++t) {
        float a = 0.f;
        if (i < ntag) a = lse_row(prev, tr + i * ntag, ntag) + em[(long long)(r0 + t) * ntag + i];
        __syncthreads();
        if (i < ntag) { prev[i] = a; alpha[(long long)(r0 + t) * ntag + i] = a; }
        __syncthreads();
    }
    if (i == 0) {
        const float z = lse_row(prev, tr + stop * ntag, ntag);
        float gold = 0.f;
        int pt = start;
        for (int t = 0; t < n; ++t) {
            const int ct = tags[r0 + t];
            gold += tr[ct * ntag + pt] + em[(long long)(r0 + t) * ntag + ct];
            pt = ct;
        }
        gold += tr[stop * ntag + pt];
        logz[d] = z;
        nll[d] = n > 0 ? (z - gold) / (float)n : 0.f;
    }
}

// d nll_d / d emissions and transitions: (marginals - gold indicators) * gout[d] / n_d
__global__ __launch_bounds__(64) void crf_nll_bwd_kernel(const float* __restrict__ em, const int* __restrict__ tags,
                                                         const int* __restrict__ doc_off, const float* __restrict__ trans, int ntag,
                                                         int start, int stop, const float* __restrict__ alpha,
                                                         const float* __restrict__ logz, const float* __restrict__ gout,
                                                         float* __restrict__ dem, float* dtrans) {
    __shared__ float beta[CRF_MAX_TAGS], nb[CRF_MAX_TAGS], prev[CRF_MAX_TAGS], tr[CRF_MAX_TAGS * CRF_MAX_TAGS];
    __shared__ float dtr[CRF_MAX_TAGS * CRF_MAX_TAGS];
    const int d = blockIdx.x, i = threadIdx.x;
    const int r0 = doc_off[d], n = doc_off[d + 1] - r0;
    if (n == 0) return;
    const float z = logz[d], g = gout[d] / (float)n;
    for (int k = i; k < ntag * ntag; k += 64) { tr[k] = trans[k]; dtr[k] = 0.f; }
    if (i < ntag) beta[i] = trans[stop * ntag + i];              // beta of the last position: the transition to STOP
    __syncthreads();
    for (int t = n - 1; t >= 0; --t) {
        // state marginal of position t and its emission gradient
        if (i < ntag) {
            const float a = alpha[(long long)(r0 + t) * ntag + i];
            const float p = expf(a + beta[i] - z);
            dem[(long long)(r0 + t) * ntag + i] = g * (p - (tags[r0 + t] == i ? 1.f : 0.f));
            if (t == n - 1) dtr[stop * ntag + i] += p;          // terminal transition i -> STOP
            // alpha of the previous position (or the initial vector) for the pairwise marginals
            prev[i] = t > 0 ? alpha[(long long)(r0 + t - 1) * ntag + i] : (i == start ? 0.f : -10000.f);
            nb[i] = em[(long long)(r0 + t) * ntag + i] + beta[i];        // emission of t + everything after it, per tag of t
        }
        __syncthreads();
        if (i < ntag) {
            // pairwise marginals p(y_{t-1} = j, y_t = i): lane i owns row i of dtrans
            for (int j = 0; j < ntag; ++j) dtr[i * ntag + j] += expf(prev[j] + tr[i * ntag + j] + nb[i] - z);
        }
        __syncthreads();
        // beta of position t-1: lane j sums over the tag i of position t
        float b = 0.f;
        if (i < ntag) {
            float mx = tr[0 * ntag + i] + nb[0];
            for (int k = 1; k < ntag; ++k) mx = fmaxf(mx, tr[k * ntag + i] + nb[k]);
            float s = 0.f;
            for (int k = 0; k < ntag; ++k) s += expf(tr[k * ntag + i] + nb[k] - mx);
            b = mx + logf(s);
        }
        __syncthreads();
        if (i < ntag) beta[i] = b;
        __syncthreads();
    }
    if (i == 0) {                 // gold path indicators
        int pt = start;
        for (int t = 0; t < n; ++t) {
            const int ct = tags[r0 + t];
            dtr[ct * ntag + pt] -= 1.f;
            pt = ct;
        }
        dtr[stop * ntag + pt] -= 1.f;
    }
    __syncthreads();
    for (int k = i; k < ntag * ntag; k += 64) unsafeAtomicAdd(dtrans + k, g * dtr[k]);
}

// Viterbi decode (model/crf.py:99-145): first maximal previous tag on ties, like torch.max
__global__ __launch_bounds__(64) void crf_viterbi_kernel(const float* __restrict__ em, const int* __restrict__ doc_off,
                                                         const float* __restrict__ trans, int ntag, int start, int stop,
                                                         int* __restrict__ bptr, int* __restrict__ path, float* __restrict__ score) {
    __shared__ float prev[CRF_MAX_TAGS], tr[CRF_MAX_TAGS * CRF_MAX_TAGS];
    const int d = blockIdx.x, i = threadIdx.x;
    const int r0 = doc_off[d], n = doc_off[d + 1] - r0;
    for (int k = i; k < ntag * ntag; k += 64) tr[k] = trans[k];
    if (i < ntag) prev[i] = (i == start) ? 0.f : -10000.f;
    __syncthreads();
    for (int t = 0; t < n; ++t) {
        float best = 0.f;
        int bj = 0;
        if (i < ntag) {
            best = prev[0] + tr[i * ntag];
            for (int j = 1; j < ntag; ++j) {
                const float v = prev[j] + tr[i * ntag + j];
                if (v > best) { best = v; bj = j; }
            }
            best += em[(long long)(r0 + t) * ntag + i];
        }
        __syncthreads();
        if (i < ntag) { prev[i] = best; bptr[(long long)(r0 + t) * ntag + i] = bj; }
        __syncthreads();
    }
    if (i == 0) {
        float best = prev[0] + tr[stop * ntag];
        int cur = 0;
        for (int j = 1; j < ntag; ++j) {
            const float v = prev[j] + tr[stop * ntag + j];
            if (v > best) { best = v; cur = j; }
        }
        score[d] = best;
        for (int t = n - 1; t >= 0; --t) {
            path[r0 + t] = cur;
            cur = bptr[(long long)(r0 + t) * ntag + cur];
        }
    }
}

__global__ void gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ idx, long long n, int C, float* __restrict__ dst) {
    const long long total = n * C;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const long long r = e / C;
        dst[e] = src[(long long)idx[r] * C + (e - r * C)];
    }
}

__global__ void scatter_rows_add_kernel(const float* __restrict__ src, const int* __restrict__ idx, long long n, int C, float* dst) {
    const long long total = n * C;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const long long r = e / C;
        unsafeAtomicAdd(dst + (long long)idx[r] * C + (e - r * C), src[e]);
    }
}

static inline int rows_grid(long long n) {
    long long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace vbg

using namespace vbg;
#define S_ ((hipStream_t)stream)

extern "C" int vbg_crf_nll_fwd(const float* emissions, const int* tags, const int* doc_off, int ndoc, const float* trans, int ntag,
                               int start_tag, int stop_tag, float* alpha, float* logz, float* nll, void* stream) {
    VBG_CHECK_ARG(emissions && tags && doc_off && trans && alpha && logz && nll && ndoc >= 0);
    VBG_CHECK_ARG(ntag >= 2 && ntag <= CRF_MAX_TAGS && start_tag >= 0 && start_tag < ntag && stop_tag >= 0 && stop_tag < ntag);
    if (ndoc == 0) return VBG_OK;
    VBG_LAUNCH(crf_nll_fwd_kernel, dim3(ndoc), dim3(64), 0, S_, emissions, tags, doc_off, trans, ntag, start_tag, stop_tag, alpha, logz,
               nll);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_crf_nll_bwd(const float* emissions, const int* tags, const int* doc_off, int ndoc, const float* trans, int ntag,
                               int start_tag, int stop_tag, const float* alpha, const float* logz, const float* gout, float* demissions,
                               float* dtrans_accum, void* stream) {
    VBG_CHECK_ARG(emissions && tags && doc_off && trans && alpha && logz && gout && demissions && dtrans_accum && ndoc >= 0);
    VBG_CHECK_ARG(ntag >= 2 && ntag <= CRF_MAX_TAGS && start_tag >= 0 && start_tag < ntag && stop_tag >= 0 && stop_tag < ntag);
    if (ndoc == 0) return VBG_OK;
    VBG_LAUNCH(crf_nll_bwd_kernel, dim3(ndoc), dim3(64), 0, S_, emissions, tags, doc_off, trans, ntag, start_tag, stop_tag, alpha, logz,
               gout, demissions, dtrans_accum);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_crf_viterbi(const float* emissions, const int* doc_off, int ndoc, const float* trans, int ntag, int start_tag,
                               int stop_tag, int* backptr, int* path, float* score, void* stream) {
    VBG_CHECK_ARG(emissions && doc_off && trans && backptr && path && score && ndoc >= 0);
    VBG_CHECK_ARG(ntag >= 2 && ntag <= CRF_MAX_TAGS && start_tag >= 0 && start_tag < ntag && stop_tag >= 0 && stop_tag < ntag);
    if (ndoc == 0) return VBG_OK;
    VBG_LAUNCH(crf_viterbi_kernel, dim3(ndoc), dim3(64), 0, S_, emissions, doc_off, trans, ntag, start_tag, stop_tag, backptr, path, score);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_gather_rows(const float* src, const int* idx, long long n, int C, float* dst, void* stream) {
    VBG_CHECK_ARG(n >= 0 && C > 0);
    if (n == 0) return VBG_OK;
    VBG_CHECK_ARG(src && idx && dst);
    VBG_LAUNCH(gather_rows_kernel, dim3(rows_grid(n * C)), dim3(256), 0, S_, src, idx, n, C, dst);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_scatter_rows_add(const float* src, const int* idx, long long n, int C, float* dst_accum, void* stream) {
    VBG_CHECK_ARG(n >= 0 && C > 0);
    if (n == 0) return VBG_OK;
    VBG_CHECK_ARG(src && idx && dst_accum);
    VBG_LAUNCH(scatter_rows_add_kernel, dim3(rows_grid(n * C)), dim3(256), 0, S_, src, idx, n, C, dst_accum);
    VBG_LAUNCH_RET();
}
