// Row-wise kernels of the BERT encoder (transformers BertModel called at
// model/BERTgrid_generator.py:134): embedding gather + LayerNorm, dropout + residual + LayerNorm,
// attention softmax (+dropout), GELU / ReLU backward, bias-gradient column sums.
// All HBM-bound: one pass over the rows, float4 where the layout allows, wave64 reductions.
#include "vbg_common.h"
#include "../../include/vbg.h"

namespace vbg {

constexpr int LN_THREADS = 256;
constexpr int LN_MAXPER = 4;      // hidden <= 1024

// ------------------------------------------------------------------------------------------
// LayerNorm helpers: each thread owns columns tid, tid+256, ... (<= 4)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void ln_forward_row(const float (&x)[LN_MAXPER], int hidden, float eps, float* sh,
                                               float (&xhat)[LN_MAXPER], float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXPER; ++j) if ((int)threadIdx.x + j * LN_THREADS < hidden) s += x[j];
    const float mean = block_sum(s, sh) / (float)hidden;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXPER; ++j) if ((int)threadIdx.x + j * LN_THREADS < hidden) { const float d = x[j] - mean; q += d * d; }
    const float var = block_sum(q, sh) / (float)hidden;
    rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int j = 0; j < LN_MAXPER; ++j) xhat[j] = (x[j] - mean) * rstd;
}

__global__ __launch_bounds__(LN_THREADS) void embed_ln_fwd_kernel(
    const int* __restrict__ ids, const int* __restrict__ pos_ids, int ntok, int hidden, const float* __restrict__ word,
    const float* __restrict__ pos, const float* __restrict__ type0, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, uint32_t drop_thr, float keep_scale, uint64_t seed, uint64_t sid,
    float* __restrict__ out, float* __restrict__ xhat_out, float* __restrict__ rstd_out) {
    __shared__ float sh[16];
    const int t = blockIdx.x;
    if (t >= ntok) return;
    const long long wid = ids[t], pid = pos_ids[t];
    float x[LN_MAXPER], xh[LN_MAXPER];
#pragma unroll
    for (int j = 0; j < LN_MAXPER; ++j) {
        const int c = threadIdx.x + j * LN_THREADS;
        x[j] = (c < hidden) ? (word[wid * hidden + c] + type0[c]) + pos[pid * hidden + c] : 0.f;
    }
    float rstd;
    ln_forward_row(x, hidden, eps, sh, xh, rstd);
#pragma unroll
    for (int j = 0; j < LN_MAXPER; ++j) {
        const int c = threadIdx.x + j * LN_THREADS;
        if (c < hidden) {
            float y = xh[j] * gamma[c] + beta[c];
            if (drop_thr) y = rng_keep(seed, sid, (uint64_t)t * hidden + c, drop_thr) ? y * keep_scale : 0.f;
            out[(long long)t * hidden + c] = y;
            xhat_out[(long long)t * hidden + c] = xh[j];
        }
    }
    if (threadIdx.x == 0) rstd_out[t] = rstd;
}

// LN backward over a chunk of rows per block; returns dz for the row in `dz`
__device__ __forceinline__ void ln_backward_row(const float (&g)[LN_MAXPER], const float (&xh)[LN_MAXPER],
                                                const float (&gam)[LN_MAXPER], float rstd, int hidden, float* sh,
                                                float (&dz)[LN_MAXPER]) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXPER; ++j) {
        if ((int)threadIdx.x + j * LN_THREADS < hidden) { const float d = g[j] * gam[j]; s1 += d; s2 += d * xh[j]; }
    }
    const float m1 = block_sum(s1, sh) / (float)hidden;
    const float m2 = block_sum(s2, sh) / (float)hidden;
#pragma unroll
    for (int j = 0; j < LN_MAXPER; ++j) dz[j] = rstd * (g[j] * gam[j] - m1 - xh[j] * m2);
}

constexpr int LN_ROWS_PER_BLOCK = 16;

__global__ __launch_bounds__(LN_THREADS) void embed_ln_bwd_kernel(
    const float* __restrict__ dout, const float* __restrict__ xhat, const float* __restrict__ rstd,
    const int* __restrict__ ids, const int* __restrict__ pos_ids, int ntok, int hidden, const float* __restrict__ gamma,
    uint32_t drop_thr, float keep_scale, uint64_t seed, uint64_t sid, float* dword, float* dpos, float* dtype0,
    float* dgamma, float* dbeta) {
    __shared__ float sh[16];
    float gam[LN_MAXPER], ag[LN_MAXPER], ab[LN_MAXPER], at[LN_MAXPER];
#pragma unroll
    for (int j = 0; j < LN_MAXPER; ++j) {
        const int c = threadIdx.x + j * LN_THREADS;
        gam[j] = (c < hidden) ? gamma[c] : 0.f;
        ag[j] = ab[j] = at[j] = 0.f;
    }
    const int t0 = blockIdx.x * LN_ROWS_PER_BLOCK;
    for (int t = t0; t < min(ntok, t0 + LN_ROWS_PER_BLOCK); ++t) {
        float g[LN_MAXPER], xh[LN_MAXPER], dz[LN_MAXPER];
#pragma unroll
        for (int j = 0; j < LN_MAXPER; ++j) {
            const int c = threadIdx.x + j * LN_THREADS;
            g[j] = xh[j] = 0.f;
            if (c < hidden) {
                float v = dout[(long long)t * hidden + c];
                if (drop_thr) v = rng_keep(seed, sid, (uint64_t)t * hidden + c, drop_thr) ? v * keep_scale : 0.f;
                g[j] = v;
                xh[j] = xhat[(long long)t * hidden + c];
                ag[j] += v * xh[j];
                ab[j] += v;
            }
        }
        ln_backward_row(g, xh, gam, rstd[t], hidden, sh, dz);
        const long long wid = ids[t], pid = pos_ids[t];
#pragma unroll
        for (int j = 0; j < LN_MAXPER; ++j) {
            const int c = threadIdx.x + j * LN_THREADS;
            if (c < hidden) {
                unsafeAtomicAdd(dword + wid * hidden + c, dz[j]);
                unsafeAtomicAdd(dpos + pid * hidden + c, dz[j]);
                at[j] += dz[j];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < LN_MAXPER; ++j) {
        const int c = threadIdx.x + j * LN_THREADS;
        if (c < hidden) {
            unsafeAtomicAdd(dgamma + c, ag[j]);
            unsafeAtomicAdd(dbeta + c, ab[j]);
            unsafeAtomicAdd(dtype0 + c, at[j]);
        }
    }
}

// ---- wave-per-row LayerNorm kernels (hidden % 256 == 0, hidden <= 1024): each lane owns float4 chunks lane + 64*j,
// no block-level barrier; a block of 256 threads handles 4 rows per pass and `wrows` passes (backward keeps the
// dgamma / dbeta partial sums of its columns in registers across all its rows; the block's sums leave as ONE plain row of a partials
// workspace, ln_fold_rows_kernel adds the rows in a fixed order).
constexpr int LN_V = 4;            // max float4 per lane (hidden <= 1024)

// rows per wave of the backward kernel: about one block of four waves per CU (258 blocks at cfg2's 4128 rows; the next row's loads are
// in flight while a row is reduced and stored), never more than 8
static int ln_bwd_wrows(int rows) {
    static const int forced = [] { const char* e = getenv("VBG_LN_WROWS"); return e ? atoi(e) : 0; }();
    if (forced > 0) return forced;
    int w = rows / 1024;
    return w < 1 ? 1 : (w > 8 ? 8 : w);
}

__device__ __forceinline__ float4 drop4(float4 v, uint32_t thr, float ks, uint64_t seed, uint64_t sid, uint64_t idx) {
    if (thr) {
        v.x = rng_keep(seed, sid, idx + 0, thr) ? v.x * ks : 0.f; v.y = rng_keep(seed, sid, idx + 1, thr) ? v.y * ks : 0.f;
        v.z = rng_keep(seed, sid, idx + 2, thr) ? v.z * ks : 0.f; v.w = rng_keep(seed, sid, idx + 3, thr) ? v.w * ks : 0.f;
    }
    return v;
}

__global__ __launch_bounds__(256) void dropout_add_ln_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ res, int rows, int hidden, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, uint32_t drop_thr, float keep_scale, uint64_t seed, uint64_t sid,
    float* __restrict__ y, float* __restrict__ xhat_out, float* __restrict__ rstd_out,
    unsigned short* __restrict__ ypl, int ldp, long long plane, unsigned short* __restrict__ yq, int ldq, long long qplane) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= rows) return;
    const int nv = hidden >> 8;                      // float4 per lane
    const long long base = (long long)t * hidden;
    float4 z[LN_V];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LN_V; ++j) {
        if (j < nv) {
            const int c = (lane + 64 * j) * 4;
            float4 v = drop4(*reinterpret_cast<const float4*>(x + base + c), drop_thr, keep_scale, seed, sid, (uint64_t)base + c);
            const float4 r = *reinterpret_cast<const float4*>(res + base + c);
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
            z[j] = v;
            s += (v.x + v.y) + (v.z + v.w);
        }
    }
    const float mean = wave_sum(s) / (float)hidden;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < LN_V; ++j)
        if (j < nv) {
            const float a = z[j].x - mean, b = z[j].y - mean, c = z[j].z - mean, d = z[j].w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)hidden + eps);
#pragma unroll
    for (int j = 0; j < LN_V; ++j)
        if (j < nv) {
            const int c = (lane + 64 * j) * 4;
            const float4 g = *reinterpret_cast<const float4*>(gamma + c), b = *reinterpret_cast<const float4*>(beta + c);
            float4 xh, o;
            xh.x = (z[j].x - mean) * rstd; xh.y = (z[j].y - mean) * rstd; xh.z = (z[j].z - mean) * rstd; xh.w = (z[j].w - mean) * rstd;
            o.x = xh.x * g.x + b.x; o.y = xh.y * g.y + b.y; o.z = xh.z * g.z + b.z; o.w = xh.w * g.w + b.w;
            *reinterpret_cast<float4*>(y + base + c) = o;
            *reinterpret_cast<float4*>(xhat_out + base + c) = xh;
            if (ypl) {                                // the bf16 planes of y (the A operand of the next plane GEMM), exact 3-way split
                const float e[4] = {o.x, o.y, o.z, o.w};
                unsigned short h[4], m[4], l[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned u = __float_as_uint(e[i]);
                    const float r1 = e[i] - __uint_as_float(u & 0xffff0000u);
                    const unsigned u1 = __float_as_uint(r1);
                    const float r2 = r1 - __uint_as_float(u1 & 0xffff0000u);
                    h[i] = (unsigned short)(u >> 16); m[i] = (unsigned short)(u1 >> 16); l[i] = (unsigned short)(__float_as_uint(r2) >> 16);
                }
                unsigned short* op = ypl + (long long)t * ldp + c;
                *reinterpret_cast<uint2*>(op) = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
                *reinterpret_cast<uint2*>(op + plane) = make_uint2(m[0] | ((unsigned)m[1] << 16), m[2] | ((unsigned)m[3] << 16));
                *reinterpret_cast<uint2*>(op + 2 * plane) = make_uint2(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16));
            }
            if (yq) {                                 // ... and its fp16-pair planes (the A operand of the form-1 forward products)
                typedef float f32x2_t __attribute__((ext_vector_type(2)));
                typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
                const f32x2_t v0 = {o.x, o.y}, v1 = {o.z, o.w};
                const f16x2_t h0 = __builtin_convertvector(v0, f16x2_t), h1 = __builtin_convertvector(v1, f16x2_t);
                const f16x2_t l0 = __builtin_convertvector((v0 - __builtin_convertvector(h0, f32x2_t)) * 2048.f, f16x2_t);
                const f16x2_t l1 = __builtin_convertvector((v1 - __builtin_convertvector(h1, f32x2_t)) * 2048.f, f16x2_t);
                unsigned short* oq = yq + (long long)t * ldq + c;
                *reinterpret_cast<uint2*>(oq) = make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1));
                *reinterpret_cast<uint2*>(oq + qplane) = make_uint2(__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1));
            }
        }
    if (lane == 0) rstd_out[t] = rstd;
}

// PL: dx (the gradient of the dense output in front of this LayerNorm) leaves as bf16 planes [3][rows][ldp] instead of fp32 -- it is
// only ever a plane operand of that layer's data- and weight-gradient products -- and its column sums (the dense layer's bias
// gradient) ride along in a third array of the partials row; the workspace is then [blocks][3][hidden] and mandatory.
// PL = 0: dx as fp32 (+ its largest magnitude); 1: dx as three bf16 planes + column sums; 2 (round 4): dx as TWO fp16 planes scaled by the
// power of two of a rigorous BOUND of |dx| -- no fp32 round trip and no split pass for a gradient that is only ever a plane operand --,
// + column sums + the true largest magnitude (the next producer's bound needs it).  The bound: dz = rstd (g gamma - mean(g gamma) -
// xhat mean(g gamma xhat)) with |xhat| <= sqrt(H) and mean |xhat| <= 1, so |dx| <= keep^-1 max rstd max |gamma| max |dy| (2 + sqrt H):
// max |dy| from the amax slot of the producer of dy, max |gamma| from the wave's own registers, max rstd by every block over the rstd
// vector (16 KB, L2 resident).  Every block derives the same bound; block 0 publishes its bit pattern (the consumers' a_amax).
template <int PL>
__global__ __launch_bounds__(256) void dropout_add_ln_bwd_kernel(
    const float* __restrict__ dy, const float* __restrict__ xhat, const float* __restrict__ rstd, int rows, int hidden,
    const float* __restrict__ gamma, uint32_t drop_thr, float keep_scale, uint64_t seed, uint64_t sid,
    float* __restrict__ dx, float* __restrict__ dres, float* dgamma, float* dbeta, float* __restrict__ part, int wrows,
    unsigned short* __restrict__ dxp, int ldp, long long plane, unsigned* dx_amax, const unsigned* dy_amax = nullptr,
    unsigned* dx_bound = nullptr) {
    const int lane = threadIdx.x & 63;
    const int nv = hidden >> 8;
    const int t0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * wrows;
    float amx = 0.f;                                  // max |dx| of this thread (dx_amax: the scale of dx as a pair-plane operand)
    float4 gam[LN_V], ag[LN_V], ab[LN_V], ac[LN_V];
#pragma unroll
    for (int j = 0; j < LN_V; ++j) {
        ac[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        ag[j] = ab[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        gam[j] = (j < nv) ? *reinterpret_cast<const float4*>(gamma + (lane + 64 * j) * 4) : ag[j];
    }
    // (the first row's loads are issued in front of the bound's prologue: its L2 round trips and barrier run under their latency)
    const int tend = min(rows, t0 + wrows);
    float4 g[LN_V], xh[LN_V], gn[LN_V], xn[LN_V];
    float rs = 0.f, rsn = 0.f;
    if (t0 < tend) {
#pragma unroll
        for (int j = 0; j < LN_V; ++j)
            if (j < nv) {
                gn[j] = *reinterpret_cast<const float4*>(dy + (long long)t0 * hidden + (lane + 64 * j) * 4);
                xn[j] = *reinterpret_cast<const float4*>(xhat + (long long)t0 * hidden + (lane + 64 * j) * 4);
            }
        rsn = rstd[t0];
    }
    float qsc = 1.f;
    if constexpr (PL == 2) {
        __shared__ float sh_b[16];
        float gm = 0.f, rm = 0.f;
#pragma unroll
        for (int j = 0; j < LN_V; ++j) gm = fmaxf(gm, fmaxf(fmaxf(fabsf(gam[j].x), fabsf(gam[j].y)), fmaxf(fabsf(gam[j].z), fabsf(gam[j].w))));
        gm = wave_max(gm);
        {   // max rstd over ALL rows (16 KB at cfg2, L2 resident): float4 loads, all of a thread's loads in flight together
            const int n4 = (((uintptr_t)rstd & 15) == 0) ? rows >> 2 : 0;
            const float4* r4 = reinterpret_cast<const float4*>(rstd);
            float4 v[4];
            for (int i0 = threadIdx.x; i0 < n4; i0 += 1024) {
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = (i0 + 256 * u < n4) ? r4[i0 + 256 * u] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u = 0; u < 4; ++u) rm = fmaxf(rm, fmaxf(fmaxf(v[u].x, v[u].y), fmaxf(v[u].z, v[u].w)));
            }
            for (int i = 4 * n4 + threadIdx.x; i < rows; i += 256) rm = fmaxf(rm, rstd[i]);
        }
        rm = block_max(rm, sh_b);
        const float ady = __uint_as_float(vbg_amax_read(dy_amax));
        const float bound = ady * gm * rm * (2.f + sqrtf((float)hidden)) * keep_scale * 1.01f;
        qsc = vbg_pow2_scale(__float_as_uint(bound)).x;
        if (dx_bound && blockIdx.x == 0 && threadIdx.x == 0) dx_bound[0] = __float_as_uint(bound);
    }
    for (int t = t0; t < tend; ++t) {
        const long long base = (long long)t * hidden;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < LN_V; ++j) { g[j] = gn[j]; xh[j] = xn[j]; }
        rs = rsn;
        if (t + 1 < tend) {                              // the next row's loads fly while this one is reduced and stored
#pragma unroll
            for (int j = 0; j < LN_V; ++j)
                if (j < nv) {
                    gn[j] = *reinterpret_cast<const float4*>(dy + base + hidden + (lane + 64 * j) * 4);
                    xn[j] = *reinterpret_cast<const float4*>(xhat + base + hidden + (lane + 64 * j) * 4);
                }
            rsn = rstd[t + 1];
        }
#pragma unroll
        for (int j = 0; j < LN_V; ++j)
            if (j < nv) {
                ag[j].x += g[j].x * xh[j].x; ag[j].y += g[j].y * xh[j].y; ag[j].z += g[j].z * xh[j].z; ag[j].w += g[j].w * xh[j].w;
                ab[j].x += g[j].x; ab[j].y += g[j].y; ab[j].z += g[j].z; ab[j].w += g[j].w;
                const float a = g[j].x * gam[j].x, b = g[j].y * gam[j].y, c2 = g[j].z * gam[j].z, d = g[j].w * gam[j].w;
                s1 += (a + b) + (c2 + d);
                s2 += (a * xh[j].x + b * xh[j].y) + (c2 * xh[j].z + d * xh[j].w);
            }
        const float m1 = wave_sum(s1) / (float)hidden, m2 = wave_sum(s2) / (float)hidden;
#pragma unroll
        for (int j = 0; j < LN_V; ++j)
            if (j < nv) {
                const int c = (lane + 64 * j) * 4;
                float4 dz;
                dz.x = rs * (g[j].x * gam[j].x - m1 - xh[j].x * m2); dz.y = rs * (g[j].y * gam[j].y - m1 - xh[j].y * m2);
                dz.z = rs * (g[j].z * gam[j].z - m1 - xh[j].z * m2); dz.w = rs * (g[j].w * gam[j].w - m1 - xh[j].w * m2);
                *reinterpret_cast<float4*>(dres + base + c) = dz;
                const float4 dd = drop4(dz, drop_thr, keep_scale, seed, sid, (uint64_t)base + c);
                if constexpr (PL == 0) {
                    *reinterpret_cast<float4*>(dx + base + c) = dd;
                    amx = fmaxf(fmaxf(amx, fmaxf(fabsf(dd.x), fabsf(dd.y))), fmaxf(fabsf(dd.z), fabsf(dd.w)));
                } else if constexpr (PL == 2) {
                    ac[j].x += dd.x; ac[j].y += dd.y; ac[j].z += dd.z; ac[j].w += dd.w;
                    amx = fmaxf(fmaxf(amx, fmaxf(fabsf(dd.x), fabsf(dd.y))), fmaxf(fabsf(dd.z), fabsf(dd.w)));
                    typedef float f32x2_t __attribute__((ext_vector_type(2)));
                    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
                    const f32x2_t v0 = {dd.x * qsc, dd.y * qsc}, v1 = {dd.z * qsc, dd.w * qsc};
                    const f16x2_t h0 = __builtin_convertvector(v0, f16x2_t), h1 = __builtin_convertvector(v1, f16x2_t);
                    const f16x2_t l0 = __builtin_convertvector((v0 - __builtin_convertvector(h0, f32x2_t)) * 2048.f, f16x2_t);
                    const f16x2_t l1 = __builtin_convertvector((v1 - __builtin_convertvector(h1, f32x2_t)) * 2048.f, f16x2_t);
                    unsigned short* op = dxp + (long long)t * ldp + c;
                    *reinterpret_cast<uint2*>(op) = make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1));
                    *reinterpret_cast<uint2*>(op + plane) = make_uint2(__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1));
                } else {
                    ac[j].x += dd.x; ac[j].y += dd.y; ac[j].z += dd.z; ac[j].w += dd.w;
                    const float e[4] = {dd.x, dd.y, dd.z, dd.w};
                    unsigned short h[4], m[4], l[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const unsigned u = __float_as_uint(e[i]);
                        const float r1 = e[i] - __uint_as_float(u & 0xffff0000u);
                        const unsigned u1 = __float_as_uint(r1);
                        const float r2 = r1 - __uint_as_float(u1 & 0xffff0000u);
                        h[i] = (unsigned short)(u >> 16); m[i] = (unsigned short)(u1 >> 16); l[i] = (unsigned short)(__float_as_uint(r2) >> 16);
                    }
                    unsigned short* op = dxp + (long long)t * ldp + c;
                    *reinterpret_cast<uint2*>(op) = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
                    *reinterpret_cast<uint2*>(op + plane) = make_uint2(m[0] | ((unsigned)m[1] << 16), m[2] | ((unsigned)m[3] << 16));
                    *reinterpret_cast<uint2*>(op + 2 * plane) = make_uint2(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16));
                }
            }
    }
    // cross-wave reduction of the column partials through LDS; the block's sums then leave as row blockIdx.x of the partials workspace
    // (plain stores; 516 blocks x 2304 float atomics used to be a ~9 us tail at the L2's atomic rate), or, without a workspace (few
    // rows), as one atomic per column straight into dgamma / dbeta
    constexpr int NA = PL != 0 ? 3 : 2;
    __shared__ float red[NA][4][256 * LN_V];
    const int w = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < LN_V; ++j)
        if (j < nv) {
            const int c = (lane + 64 * j) * 4;
            *reinterpret_cast<float4*>(&red[0][w][c]) = ag[j];
            *reinterpret_cast<float4*>(&red[1][w][c]) = ab[j];
            if constexpr (PL != 0) *reinterpret_cast<float4*>(&red[2][w][c]) = ac[j];
        }
    __syncthreads();
    if (part) {
        float* pp = part + (size_t)blockIdx.x * NA * hidden;
        for (int c = threadIdx.x; c < hidden; c += 256) {
            pp[c] = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
            pp[hidden + c] = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
            if constexpr (PL != 0) pp[2 * hidden + c] = (red[2][0][c] + red[2][1][c]) + (red[2][2][c] + red[2][3][c]);
        }
    } else {
        for (int c = threadIdx.x; c < hidden; c += 256) {
            unsafeAtomicAdd(dgamma + c, (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]));
            unsafeAtomicAdd(dbeta + c, (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]));
        }
    }
    if constexpr (PL != 1) {
        if (dx_amax) {                                // (uniform)
            __syncthreads();
            vbg_amax_publish(amx, dx_amax, &red[0][0][0]);
        }
    }
}

// adds the rows of the partials workspace ([nrows][na][hidden], one row per block of the backward kernel) into dgamma / dbeta (/ dbias):
// a block is 64 columns x 16 row groups (one wave each), every wave has its ~nrows / 16 loads in flight together, the groups meet in LDS
// and are added in group order -- the sum does not depend on who ran when
__global__ __launch_bounds__(1024) void ln_fold_rows_kernel(const float* __restrict__ part, int nrows, int hidden, int na, float* dgamma,
                                                            float* dbeta, float* dbias) {
    __shared__ float sh[16][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
    const size_t ld = (size_t)na * hidden;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    int r = grp;
    for (; r + 48 < nrows; r += 64) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += part[(size_t)(r + 16 * j) * ld + col];
    }
    for (; r < nrows; r += 16) v[0] += part[(size_t)r * ld + col];
    sh[grp][threadIdx.x & 63] = (v[0] + v[1]) + (v[2] + v[3]);
    __syncthreads();
    if (grp == 0) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += sh[j][threadIdx.x];
        if (col < hidden) dgamma[col] += t; else if (col < 2 * hidden) dbeta[col - hidden] += t; else dbias[col - 2 * hidden] += t;
    }
}

// ------------------------------------------------------------------------------------------
// attention softmax: one wave per (group, row); L <= 512 -> 8 elements per lane
// ------------------------------------------------------------------------------------------
constexpr int SM_PER = 8;

__global__ __launch_bounds__(64) void softmax_fwd_kernel(float* s, const long long* __restrict__ off,
                                                         const int* __restrict__ len, const int* __restrict__ ldp,
                                                         int heads, int maxlen, float scale, uint32_t drop_thr,
                                                         uint64_t seed, uint64_t sid) {
    const int g = blockIdx.y, row = blockIdx.x, seq = g / heads;
    const int L = len[seq];
    if (row >= L) return;
    const int ld = ldp[seq];
    float* p = s + off[g] + (long long)row * ld;
    const int lane = threadIdx.x;
    float v[SM_PER];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < SM_PER; ++j) {
        const int c = lane + j * 64;
        v[j] = (c < L) ? p[c] * scale : -INFINITY;
        mx = fmaxf(mx, v[j]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < SM_PER; ++j) {
        v[j] = (lane + j * 64 < L) ? __expf(v[j] - mx) : 0.f;
        sum += v[j];
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    const uint64_t base = ((uint64_t)g * maxlen + row) * maxlen;
#pragma unroll
    for (int j = 0; j < SM_PER; ++j) {
        const int c = lane + j * 64;
        if (c < ld) {
            float o = 0.f;
            if (c < L) {
                o = v[j] * inv;
                if (drop_thr && !rng_keep(seed, sid, base + c, drop_thr)) o = -o;
            }
            p[c] = o;
        }
    }
}

__global__ __launch_bounds__(64) void softmax_bwd_kernel(const float* __restrict__ pbuf, float* dpbuf,
                                                         const long long* __restrict__ off, const int* __restrict__ len,
                                                         const int* __restrict__ ldp, int heads, float scale,
                                                         float keep_scale) {
    const int g = blockIdx.y, row = blockIdx.x, seq = g / heads;
    const int L = len[seq];
    if (row >= L) return;
    const int ld = ldp[seq];
    const float* p = pbuf + off[g] + (long long)row * ld;
    float* dp = dpbuf + off[g] + (long long)row * ld;
    const int lane = threadIdx.x;
    float pv[SM_PER], dv[SM_PER];
    float d = 0.f;
#pragma unroll
    for (int j = 0; j < SM_PER; ++j) {
        const int c = lane + j * 64;
        pv[j] = dv[j] = 0.f;
        if (c < L) {
            const float ps = p[c];
            pv[j] = fabsf(ps);
            // sign bit set (including -0.0) = dropped
            dv[j] = (__float_as_uint(ps) >> 31) ? 0.f : dp[c] * keep_scale;
            d += pv[j] * dv[j];
        }
    }
    d = wave_sum(d);
#pragma unroll
    for (int j = 0; j < SM_PER; ++j) {
        const int c = lane + j * 64;
        if (c < ld) dp[c] = (c < L) ? scale * pv[j] * (dv[j] - d) : 0.f;
    }
}

// ------------------------------------------------------------------------------------------
__global__ void gelu_bwd_kernel(const float* __restrict__ h, float* dg, long long n4, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 hv = reinterpret_cast<const float4*>(h)[i];
        float4 g = reinterpret_cast<float4*>(dg)[i];
        g.x *= gelu_erf_grad(hv.x); g.y *= gelu_erf_grad(hv.y); g.z *= gelu_erf_grad(hv.z); g.w *= gelu_erf_grad(hv.w);
        reinterpret_cast<float4*>(dg)[i] = g;
    }
    if (blockIdx.x == 0) for (long long i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) dg[i] *= gelu_erf_grad(h[i]);
}

__global__ void relu_bwd_kernel(const float* __restrict__ y, float* dy, long long n4, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 yv = reinterpret_cast<const float4*>(y)[i];
        float4 g = reinterpret_cast<float4*>(dy)[i];
        g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f; g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
        reinterpret_cast<float4*>(dy)[i] = g;
    }
    if (blockIdx.x == 0) for (long long i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) dy[i] = y[i] > 0.f ? dy[i] : 0.f;
}

__global__ void add_inplace_kernel(float* a, const float* __restrict__ b, long long n4, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 x = reinterpret_cast<float4*>(a)[i];
        const float4 y = reinterpret_cast<const float4*>(b)[i];
        x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
        reinterpret_cast<float4*>(a)[i] = x;
    }
    if (blockIdx.x == 0) for (long long i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) a[i] += b[i];
}

__global__ void scale_inplace_kernel(float* a, long long n, float s) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) a[i] *= s;
}

// column sums: block = 64 columns x 4 row lanes, 256 rows per block
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, long long ld, int M, int N, float* out) {
    __shared__ float sh[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int r0 = blockIdx.y * 256;
    float s = 0.f;
    if (c < N) {
        for (int r = r0 + rl; r < min(M, r0 + 256); r += 4) s += x[(long long)r * ld + c];
    }
    sh[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && c < N) unsafeAtomicAdd(out + c, sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

// vector variant (16-byte aligned rows, N % 4 == 0): a block covers up to 64 column QUADS x RL row lanes and a row range sized
// so that the grid has ~192 blocks; float4 loads (a wave reads whole 1 KB row pieces), one fp32 atomic per column per block
__global__ __launch_bounds__(256) void colsum_vec_kernel(const float* __restrict__ x, long long ld4, int M, int N4, int rows_per_block,
                                                         float* out) {
    __shared__ float4 sh[256];
    const int QB = min(N4, 64), RL = 256 / QB;
    const int q = threadIdx.x % QB, rl = threadIdx.x / QB;
    const int cq = blockIdx.x * 64 + q;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cq < N4 && rl < RL) {
#pragma unroll 4
        for (int r = r0 + rl; r < r1; r += RL) {
            const float4 v = reinterpret_cast<const float4*>(x)[(long long)r * ld4 + cq];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    if (rl == 0 && cq < N4) {
        float4 a = sh[q];
        for (int l = 1; l < RL; ++l) { const float4 v = sh[l * QB + q]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
        unsafeAtomicAdd(out + cq * 4 + 0, a.x); unsafeAtomicAdd(out + cq * 4 + 1, a.y);
        unsafeAtomicAdd(out + cq * 4 + 2, a.z); unsafeAtomicAdd(out + cq * 4 + 3, a.w);
    }
}

// fp64 column sums.  A bias gradient such as that of the 1x1 segmentation classifiers (model/semantic_segmentation_head.py:66-78) is
// the sum of ~1e5..1e6 per-pixel terms (p - onehot) / N whose partial sums wander to 1e3 x the final value (page regions of one label):
// fp32 accumulation across blocks leaves 1e-3 relative error there (the reference's torch.sum is pairwise).  Here every thread, block
// and the cross-block stage accumulate in double: ws[n] (fp64, one per column) takes one double atomic per block and column, and
// colsum_f64_finish_kernel rounds ONCE to fp32.
// small N (contiguous rows of <= NMAX floats): a thread owns whole rows (a wave reads 64 consecutive rows = one contiguous piece)
template <int NMAX>
__global__ __launch_bounds__(256) void colsum_f64_rows_kernel(const float* __restrict__ x, long long ld, int M, int N, double* ws) {
    __shared__ double sh[4][NMAX];
    double acc[NMAX];
#pragma unroll
    for (int c = 0; c < NMAX; ++c) acc[c] = 0.0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < M; r += stride) {
        const float* row = x + r * ld;
#pragma unroll
        for (int c = 0; c < NMAX; ++c)
            if (c < N) acc[c] += (double)row[c];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < NMAX; ++c) {
        double v = acc[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) sh[wave][c] = v;
    }
    __syncthreads();
    if (threadIdx.x < N) unsafeAtomicAdd(ws + threadIdx.x, sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}
// general N: block = 64 columns x 4 row lanes over a row range
__global__ __launch_bounds__(256) void colsum_f64_kernel(const float* __restrict__ x, long long ld, int M, int N, int rows_per_block, double* ws) {
    __shared__ double sh[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    double s = 0.0;
    if (c < N)
        for (int r = r0 + rl; r < r1; r += 4) s += (double)x[(long long)r * ld + c];
    sh[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && c < N) unsafeAtomicAdd(ws + c, sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}
__global__ void colsum_f64_finish_kernel(const double* __restrict__ ws, int N, float* out, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < N) out[c] = accumulate ? (float)((double)out[c] + ws[c]) : (float)ws[c];
}

// plain row softmax for the returned class probabilities ([rows, cols], cols small)
__global__ void row_softmax_kernel(const float* __restrict__ x, int rows, int cols, float* __restrict__ y) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float* p = x + (long long)r * cols;
    float mx = p[0];
    for (int c = 1; c < cols; ++c) mx = fmaxf(mx, p[c]);
    float s = 0.f;
    for (int c = 0; c < cols; ++c) s += expf(p[c] - mx);
    const float inv = 1.f / s;
    for (int c = 0; c < cols; ++c) y[(long long)r * cols + c] = expf(p[c] - mx) * inv;
}

static inline int ew_grid(long long n, int block) {
    long long g = (n + block - 1) / block;
    if (g > 256 * 8) g = 256 * 8;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace vbg

using namespace vbg;

extern "C" int vbg_embed_ln_fwd(const int* ids, const int* pos_ids, int ntok, int hidden, const float* word,
                                const float* pos, const float* type0, const float* gamma, const float* beta, float eps,
                                float drop_p, unsigned long long seed, unsigned long long sid, float* out, float* xhat,
                                float* rstd, void* stream) {
    VBG_CHECK_ARG(ids && pos_ids && word && pos && type0 && gamma && beta && out && xhat && rstd);
    VBG_CHECK_ARG(hidden > 0 && hidden <= LN_THREADS * LN_MAXPER && drop_p >= 0.f && drop_p < 1.f);
    if (ntok <= 0) return VBG_OK;
    VBG_LAUNCH(embed_ln_fwd_kernel, dim3(ntok), dim3(LN_THREADS), 0, (hipStream_t)stream, ids, pos_ids, ntok, hidden,
                       word, pos, type0, gamma, beta, eps, drop_threshold(drop_p), 1.0f / (1.0f - drop_p), seed, sid, out,
                       xhat, rstd);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_embed_ln_bwd(const float* dout, const float* xhat, const float* rstd, const int* ids, const int* pos_ids,
                                int ntok, int hidden, const float* gamma, float drop_p, unsigned long long seed,
                                unsigned long long sid, float* dword, float* dpos, float* dtype0, float* dgamma,
                                float* dbeta, void* stream) {
    VBG_CHECK_ARG(dout && xhat && rstd && ids && pos_ids && gamma && dword && dpos && dtype0 && dgamma && dbeta);
    VBG_CHECK_ARG(hidden > 0 && hidden <= LN_THREADS * LN_MAXPER && drop_p >= 0.f && drop_p < 1.f);
    if (ntok <= 0) return VBG_OK;
    VBG_LAUNCH(embed_ln_bwd_kernel, dim3(cdiv(ntok, LN_ROWS_PER_BLOCK)), dim3(LN_THREADS), 0, (hipStream_t)stream,
                       dout, xhat, rstd, ids, pos_ids, ntok, hidden, gamma, drop_threshold(drop_p), 1.0f / (1.0f - drop_p),
                       seed, sid, dword, dpos, dtype0, dgamma, dbeta);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_dropout_add_ln_fwd(const float* x, const float* res, int rows, int hidden, const float* gamma,
                                      const float* beta, float eps, float drop_p, unsigned long long seed,
                                      unsigned long long sid, float* y, float* xhat, float* rstd, void* stream) {
    VBG_CHECK_ARG(x && res && gamma && beta && y && xhat && rstd);
    VBG_CHECK_ARG(hidden > 0 && hidden % 256 == 0 && hidden <= 256 * LN_V && drop_p >= 0.f && drop_p < 1.f);
    VBG_CHECK_ARG(((uintptr_t)x | (uintptr_t)res | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)y | (uintptr_t)xhat) % 16 == 0);
    if (rows <= 0) return VBG_OK;
    VBG_LAUNCH(dropout_add_ln_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, res, rows, hidden,
               gamma, beta, eps, drop_threshold(drop_p), 1.0f / (1.0f - drop_p), seed, sid, y, xhat, rstd, (unsigned short*)nullptr, 0, 0ll,
               (unsigned short*)nullptr, 0, 0ll);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_dropout_add_ln_fwd_planes(const float* x, const float* res, int rows, int hidden, const float* gamma,
                                             const float* beta, float eps, float drop_p, unsigned long long seed,
                                             unsigned long long sid, float* y, float* xhat, float* rstd, unsigned short* y_planes, int ldp,
                                             long long plane, unsigned short* y_pair, int ldq, long long qplane, void* stream) {
    VBG_CHECK_ARG(x && res && gamma && beta && y && xhat && rstd && (y_planes || y_pair));
    VBG_CHECK_ARG(!y_pair || (ldq % 4 == 0 && ldq >= hidden && qplane % 4 == 0 && qplane >= (long long)rows * ldq && ((uintptr_t)y_pair & 7) == 0));
    VBG_CHECK_ARG(hidden > 0 && hidden % 256 == 0 && hidden <= 256 * LN_V && drop_p >= 0.f && drop_p < 1.f);
    VBG_CHECK_ARG(((uintptr_t)x | (uintptr_t)res | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)y | (uintptr_t)xhat) % 16 == 0);
    VBG_CHECK_ARG(!y_planes || (ldp % 4 == 0 && ldp >= hidden && plane % 4 == 0 && plane >= (long long)rows * ldp && ((uintptr_t)y_planes & 7) == 0));
    if (rows <= 0) return VBG_OK;
    VBG_LAUNCH(dropout_add_ln_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, res, rows, hidden,
               gamma, beta, eps, drop_threshold(drop_p), 1.0f / (1.0f - drop_p), seed, sid, y, xhat, rstd, y_planes, ldp, plane, y_pair, ldq, qplane);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_dropout_add_ln_bwd(const float* dy, const float* xhat, const float* rstd, int rows, int hidden,
                                      const float* gamma, float drop_p, unsigned long long seed, unsigned long long sid,
                                      float* dx, float* dres, float* dgamma, float* dbeta, float* slots_ws, unsigned* dx_amax, void* stream) {
    VBG_CHECK_ARG(dy && xhat && rstd && gamma && dx && dres && dgamma && dbeta);
    VBG_CHECK_ARG(hidden > 0 && hidden % 256 == 0 && hidden <= 256 * LN_V && drop_p >= 0.f && drop_p < 1.f);
    VBG_CHECK_ARG(((uintptr_t)dy | (uintptr_t)xhat | (uintptr_t)gamma | (uintptr_t)dx | (uintptr_t)dres) % 16 == 0);
    if (rows <= 0) return VBG_OK;
    const int wrows = slots_ws ? ln_bwd_wrows(rows) : 2, nblk = cdiv(rows, 4 * wrows);
    VBG_LAUNCH(dropout_add_ln_bwd_kernel<0>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, dy, xhat, rstd, rows,
               hidden, gamma, drop_threshold(drop_p), 1.0f / (1.0f - drop_p), seed, sid, dx, dres, dgamma, dbeta, slots_ws, wrows,
               (unsigned short*)nullptr, 0, 0ll, dx_amax);
    if (slots_ws) VBG_LAUNCH(ln_fold_rows_kernel, dim3(2 * hidden / 64), dim3(1024), 0, (hipStream_t)stream, slots_ws, nblk, hidden, 2, dgamma,
                             dbeta, (float*)nullptr);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_dropout_add_ln_bwd_planes(const float* dy, const float* xhat, const float* rstd, int rows, int hidden,
                                             const float* gamma, float drop_p, unsigned long long seed, unsigned long long sid,
                                             unsigned short* dx_planes, int ldp, long long plane, float* dres, float* dgamma, float* dbeta,
                                             float* dbias_accum, float* slots3_ws, void* stream) {
    VBG_CHECK_ARG(dy && xhat && rstd && gamma && dx_planes && dres && dgamma && dbeta && dbias_accum && slots3_ws);
    VBG_CHECK_ARG(hidden > 0 && hidden % 256 == 0 && hidden <= 256 * LN_V && drop_p >= 0.f && drop_p < 1.f);
    VBG_CHECK_ARG(((uintptr_t)dy | (uintptr_t)xhat | (uintptr_t)gamma | (uintptr_t)dres) % 16 == 0);
    VBG_CHECK_ARG(ldp % 4 == 0 && ldp >= hidden && plane % 4 == 0 && plane >= (long long)rows * ldp && ((uintptr_t)dx_planes & 7) == 0);
    if (rows <= 0) return VBG_OK;
    const int wrows = ln_bwd_wrows(rows), nblk = cdiv(rows, 4 * wrows);
    VBG_LAUNCH(dropout_add_ln_bwd_kernel<1>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, dy, xhat, rstd, rows,
               hidden, gamma, drop_threshold(drop_p), 1.0f / (1.0f - drop_p), seed, sid, (float*)nullptr, dres, dgamma, dbeta, slots3_ws, wrows,
               dx_planes, ldp, plane, (unsigned*)nullptr);
    VBG_LAUNCH(ln_fold_rows_kernel, dim3(3 * hidden / 64), dim3(1024), 0, (hipStream_t)stream, slots3_ws, nblk, hidden, 3, dgamma, dbeta, dbias_accum);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_dropout_add_ln_bwd_pair(const float* dy, const float* xhat, const float* rstd, int rows, int hidden,
                                           const float* gamma, float drop_p, unsigned long long seed, unsigned long long sid,
                                           unsigned short* dx_pair, int ldp, long long plane, float* dres, float* dgamma, float* dbeta,
                                           float* dbias_accum, float* slots3_ws, const unsigned* dy_amax, unsigned* dx_amax, unsigned* dx_bound,
                                           void* stream) {
    VBG_CHECK_ARG(dy && xhat && rstd && gamma && dx_pair && dres && dgamma && dbeta && dbias_accum && slots3_ws && dy_amax && dx_bound);
    VBG_CHECK_ARG(hidden > 0 && hidden % 256 == 0 && hidden <= 256 * LN_V && drop_p >= 0.f && drop_p < 1.f);
    VBG_CHECK_ARG(((uintptr_t)dy | (uintptr_t)xhat | (uintptr_t)gamma | (uintptr_t)dres) % 16 == 0);
    VBG_CHECK_ARG(ldp % 4 == 0 && ldp >= hidden && plane % 4 == 0 && plane >= (long long)rows * ldp && ((uintptr_t)dx_pair & 7) == 0);
    if (rows <= 0) return VBG_OK;
    const int wrows = ln_bwd_wrows(rows), nblk = cdiv(rows, 4 * wrows);
    VBG_LAUNCH(dropout_add_ln_bwd_kernel<2>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, dy, xhat, rstd, rows,
               hidden, gamma, drop_threshold(drop_p), 1.0f / (1.0f - drop_p), seed, sid, (float*)nullptr, dres, dgamma, dbeta, slots3_ws, wrows,
               dx_pair, ldp, plane, dx_amax, dy_amax, dx_bound);
    VBG_LAUNCH(ln_fold_rows_kernel, dim3(3 * hidden / 64), dim3(1024), 0, (hipStream_t)stream, slots3_ws, nblk, hidden, 3, dgamma, dbeta, dbias_accum);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_softmax_fwd(float* s, const long long* off, const int* len, const int* ldp, int ngroups, int heads,
                               int maxlen, float scale, float drop_p, unsigned long long seed, unsigned long long sid,
                               void* stream) {
    VBG_CHECK_ARG(s && off && len && ldp && heads > 0 && ngroups >= 0 && ngroups % heads == 0);
    VBG_CHECK_ARG(maxlen >= 0 && maxlen <= 64 * SM_PER && drop_p >= 0.f && drop_p < 1.f);
    if (ngroups == 0 || maxlen == 0) return VBG_OK;
    VBG_LAUNCH(softmax_fwd_kernel, dim3(maxlen, ngroups), dim3(64), 0, (hipStream_t)stream, s, off, len, ldp, heads,
                       maxlen, scale, drop_threshold(drop_p), seed, sid);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_softmax_bwd(const float* p, float* dp, const long long* off, const int* len, const int* ldp, int ngroups,
                               int heads, int maxlen, float scale, float drop_p, void* stream) {
    VBG_CHECK_ARG(p && dp && off && len && ldp && heads > 0 && ngroups >= 0 && ngroups % heads == 0);
    VBG_CHECK_ARG(maxlen >= 0 && maxlen <= 64 * SM_PER && drop_p >= 0.f && drop_p < 1.f);
    if (ngroups == 0 || maxlen == 0) return VBG_OK;
    VBG_LAUNCH(softmax_bwd_kernel, dim3(maxlen, ngroups), dim3(64), 0, (hipStream_t)stream, p, dp, off, len, ldp,
                       heads, scale, 1.0f / (1.0f - drop_p));
    VBG_LAUNCH_RET();
}

extern "C" int vbg_gelu_bwd(const float* h, float* dg, long long n, void* stream) {
    VBG_CHECK_ARG(h && dg && n >= 0);
    if (n == 0) return VBG_OK;
    VBG_LAUNCH(gelu_bwd_kernel, dim3(ew_grid(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, h, dg, n / 4, n);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_relu_bwd(const float* y, float* dy, long long n, void* stream) {
    VBG_CHECK_ARG(y && dy && n >= 0);
    if (n == 0) return VBG_OK;
    VBG_LAUNCH(relu_bwd_kernel, dim3(ew_grid(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, y, dy, n / 4, n);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_add_inplace(float* a, const float* b, long long n, void* stream) {
    VBG_CHECK_ARG(a && b && n >= 0);
    if (n == 0) return VBG_OK;
    VBG_LAUNCH(add_inplace_kernel, dim3(ew_grid(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, a, b, n / 4, n);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_scale_inplace(float* x, long long n, float s, void* stream) {
    VBG_CHECK_ARG(x && n >= 0);
    if (n == 0) return VBG_OK;
    VBG_LAUNCH(scale_inplace_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, x, n, s);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_colsum(const float* x, long long ld, int M, int N, float* out, int accumulate, void* stream) {
    VBG_CHECK_ARG(x && out && M >= 0 && N >= 0);
    if (N == 0) return VBG_OK;
    hipStream_t s = (hipStream_t)stream;
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)N, s);
        if (e != hipSuccess) return (int)e;
    }
    if (M == 0) return VBG_OK;
    const int N4 = N / 4;
    const bool quads_ok = N % 4 == 0 && (N4 <= 64 ? (256 % N4 == 0) : true);
    if (quads_ok && ld % 4 == 0 && ((uintptr_t)x % 16 == 0)) {
        const int QB = N4 < 64 ? N4 : 64, RL = 256 / QB, CG = cdiv(N4, 64);
        long long rpb = cdiv((long long)M * CG, 192);       // ~192 blocks: the same-address atomics, not the loads, set the time (sweep: 128..2048)
        if (rpb < 4 * RL) rpb = 4 * RL;
        rpb = cdiv(rpb, RL) * RL;
        VBG_LAUNCH(colsum_vec_kernel, dim3(CG, cdiv(M, rpb)), dim3(256), 0, s, x, ld / 4, M, N4, (int)rpb, out);
    } else {
        VBG_LAUNCH(colsum_kernel, dim3(cdiv(N, 64), cdiv(M, 256)), dim3(256), 0, s, x, ld, M, N, out);
    }
    VBG_LAUNCH_RET();
}

extern "C" int vbg_colsum_f64(const float* x, long long ld, int M, int N, float* out, int accumulate, double* ws, void* stream) {
    VBG_CHECK_ARG(x && out && ws && M >= 0 && N >= 0 && ld >= N);
    if (N == 0) return VBG_OK;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(ws, 0, sizeof(double) * (size_t)N, s);
    if (e != hipSuccess) return (int)e;
    if (M > 0) {
        const int blocks = (int)((M + 1023) / 1024 < 128 ? (M + 1023) / 1024 : 128);     // >= 4 rows per thread, <= 128 double atomics per column
        if (N <= 4) VBG_LAUNCH(colsum_f64_rows_kernel<4>, dim3(blocks), dim3(256), 0, s, x, ld, M, N, ws);
        else if (N <= 8) VBG_LAUNCH(colsum_f64_rows_kernel<8>, dim3(blocks), dim3(256), 0, s, x, ld, M, N, ws);
        else if (N <= 16) VBG_LAUNCH(colsum_f64_rows_kernel<16>, dim3(blocks), dim3(256), 0, s, x, ld, M, N, ws);
        else {
            const int CG = cdiv(N, 64);
            long long rpb = cdiv((long long)M * CG, 192);
            if (rpb < 16) rpb = 16;
            rpb = cdiv(rpb, 4) * 4;
            VBG_LAUNCH(colsum_f64_kernel, dim3(CG, cdiv(M, rpb)), dim3(256), 0, s, x, ld, M, N, (int)rpb, ws);
        }
    }
    VBG_LAUNCH(colsum_f64_finish_kernel, dim3(cdiv(N, 256)), dim3(256), 0, s, ws, N, out, accumulate);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_row_softmax(const float* x, int rows, int cols, float* y, void* stream) {
    VBG_CHECK_ARG(rows >= 0 && cols > 0);
    if (rows == 0) return VBG_OK;
    VBG_CHECK_ARG(x && y);
    VBG_LAUNCH(row_softmax_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, (hipStream_t)stream, x, rows, cols, y);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_ln_bwd_ws_rows(int rows) { return rows > 0 ? cdiv(rows, 4 * ln_bwd_wrows(rows)) : 0; }

extern "C" int vbg_version(void) { return VBG_VERSION; }
