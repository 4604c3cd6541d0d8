// fp32 MFMA GEMM / implicit-GEMM convolution family for gfx950 (MI355X).
//
// One kernel template computes  C[M,N] (+)= opA[M,K] * opB[K,N]  with exact-fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain, 157 TF peak) for every contraction on the
// ViBERTgrid hot path: BERT linears and attention products (reference: transformers BertModel
// called at model/BERTgrid_generator.py:134), the ResNet-FPN convolutions as implicit GEMM over
// NHWC activations (model/ResNetFPN_ViBERTgrid.py:478-508, 612-648), the concat-free early /
// late / P_fuse fusions (K segments read from several tensors in place; :315-321, :502-506,
// model/field_type_classification_head.py:181-188) and all their dgrad / wgrad products.
//
// Tiling: 256 threads = 4 waves (2x2), block tile BMxBN in {128x128, 64x64}, BK = 16, operands
// staged through LDS as [k][row] so each MFMA operand read is a conflict-free ds_read_b32 over 32
// consecutive rows; register prefetch of tile t+1 overlaps the MFMAs of tile t; LDS double
// buffered (one barrier per k-tile).  fp32 MFMA issues one 32x32x2 every 64 cycles per SIMD, so
// LDS bandwidth (16 B/clk/CU needed) is never the limiter; the loaders are kept simple.
#include "vbg_common.h"
#include "../../include/vbg.h"

namespace vbg {

__device__ __forceinline__ float4 ldv4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// keep the first `nvalid` of 4 elements (branch-free: v_cndmask)
__device__ __forceinline__ float4 mask4(float4 v, int nvalid) {
    v.x = nvalid > 0 ? v.x : 0.f; v.y = nvalid > 1 ? v.y : 0.f;
    v.z = nvalid > 2 ? v.z : 0.f; v.w = nvalid > 3 ? v.w : 0.f;
    return v;
}

// VEC = both operands 16-byte aligned with leading dimensions % 4 == 0: every global load is an unconditional
// float4 from a CLAMPED (always legal) address, out-of-range elements are zeroed by selects -> the k-loop has no
// divergent branches, so the next tile's loads stay in flight behind the MFMAs.  !VEC = general scalar path.
template <int BM, int BN, int BK, int AK, int BKD, bool VEC>
__global__ __launch_bounds__(256) void gemm_kernel(const vbg_gemm_desc p) {
    constexpr int KF = BK / 4;                 // float4 chunks per row of a K-contiguous tile
    constexpr bool A_KC = (AK == VBG_OP_DENSE_K || AK == VBG_OP_CONV_K);
    constexpr bool B_KC = (BKD == VBG_OP_DENSE_K);
    constexpr int SA = A_KC ? BM + 1 : BM + 4;
    constexpr int SB = B_KC ? BN + 1 : BN + 4;
    constexpr int NA = BM * KF / 256;
    constexpr int NB = BN * KF / 256;
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int TM = WM / 32, TN = WN / 32;
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (SA + SB)];
    float* const As = smem;
    float* const Bs = smem + 2 * BK * SA;

    const int tid = threadIdx.x;
    const int z = blockIdx.z;
    const int grp = z / p.splitk, split = z - grp * p.splitk;
    int M = p.M, N = p.N, K = p.K;
    const float* A = p.A;
    const float* B = p.B;
    float* C = p.C;
    float* C2 = p.C2;
    const float* bias = p.bias;
    if (p.grp) {
        const long long* g = p.grp + 8 * (long long)grp;
        M = (int)g[0]; N = (int)g[1]; K = (int)g[2];
        A += g[3]; B += g[4]; C += g[5];
        if (C2) C2 += g[5];
        if (bias) bias += g[6];
    }
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    if (m0 >= M || n0 >= N) return;
    const int nkt = (K + BK - 1) / BK;
    const int per = (nkt + p.splitk - 1) / p.splitk;
    const int kt0 = split * per;
    const int kt1 = min(nkt, kt0 + per);
    if (kt0 >= kt1) return;

    // ---------------- per-thread loader state -----------------------------------------
    // A, K-contiguous kinds: float4 #i covers row (f>>2), k offset (f&3)*4
    // A/B, row-contiguous kinds: float4 #i covers rows (f % (BR/4))*4.., k index f / (BR/4)
    int a_n[NA], a_y[NA], a_x[NA];          // row -> (image, y, x) for conv / shifted segments
    bool a_rv[NA];
    if constexpr (A_KC) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int f = tid + i * 256;
            const int gm = m0 + f / KF;
            a_rv[i] = gm < M;
            a_n[i] = a_y[i] = a_x[i] = 0;
            if (AK == VBG_OP_CONV_K || p.a_H > 0) {
                const int Hr = (AK == VBG_OP_CONV_K) ? p.geo.Hr : p.a_H;
                const int Wr = (AK == VBG_OP_CONV_K) ? p.geo.Wr : p.a_W;
                const int g = a_rv[i] ? gm : 0;
                a_x[i] = g % Wr;
                const int t = g / Wr;
                a_y[i] = t % Hr;
                a_n[i] = t / Hr;
            }
        }
    }
    // B CONV_R: columns (tap, ci) fixed per thread
    int b_dy[NB], b_dx[NB], b_ci[NB];
    if constexpr (BKD == VBG_OP_CONV_R) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int f = tid + i * 256;
            const int c = n0 + (f % (BN / 4)) * 4;
            const int tap = c / p.geo.Cs;
            b_ci[i] = c - tap * p.geo.Cs;
            b_dy[i] = tap / p.geo.kw;
            b_dx[i] = tap - b_dy[i] * p.geo.kw;
        }
    }

    float4 ra[NA], rb[NB];
    int ra_n[NA], rb_n[NB];        // VEC path: #valid elements of each float4; the zero-masking (and the A prologue) is
                                   // applied when the registers are written to LDS, AFTER the MFMAs of the current tile,
                                   // so nothing consumes the loads early and they stay in flight behind the compute

    auto ld4 = [](const float* ptr, bool vec, int nvalid) -> float4 {
        // nvalid: how many of the 4 consecutive elements are in range (<=0: none)
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (nvalid >= 4 && vec) {
            v = *reinterpret_cast<const float4*>(ptr);
        } else if (nvalid > 0) {
            v.x = ptr[0];
            if (nvalid > 1) v.y = ptr[1];
            if (nvalid > 2) v.z = ptr[2];
            if (nvalid > 3) v.w = ptr[3];
        }
        return v;
    };

    auto load_tiles = [&](int kt) {
        const int k0 = kt * BK;
        if constexpr (VEC) {
            // ------------------------------ A (vector, branch-free) ------------------------------
            if constexpr (AK == VBG_OP_DENSE_K) {
                int seg = 0;
                while (seg + 1 < p.a_nseg && k0 >= p.a_seg_kend[seg]) ++seg;
                const float* base = p.a_seg_ptr[seg] + (A - p.A);
                const long long ld = p.a_seg_ld[seg];
                const int kbeg = (seg == 0) ? 0 : p.a_seg_kend[seg - 1];
                const int kspan = ((p.a_nseg == 1) ? K : p.a_seg_kend[seg]) - kbeg;
                const int klast = ((kspan + 3) & ~3) - 4;
                const int sh = p.a_seg_shift[seg];
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    const int f = tid + i * 256;
                    const int kk = k0 - kbeg + (f % KF) * 4;
                    long long row = min(m0 + f / KF, M - 1);
                    if (sh > 0) row = ((long long)a_n[i] * (p.a_H >> sh) + (a_y[i] >> sh)) * (p.a_W >> sh) + (a_x[i] >> sh);
                    ra[i] = ldv4(base + row * ld + min(kk, klast)); ra_n[i] = kspan - kk;
                }
            } else if constexpr (AK == VBG_OP_CONV_K) {
                const int Cs = p.geo.Cs;
                const int tap = k0 / Cs;
                const int c0 = k0 - tap * Cs;
                const int dy = tap / p.geo.kw, dx = tap - dy * p.geo.kw;
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    const int f = tid + i * 256;
                    int sy, sx;
                    bool ok = a_rv[i];
                    if (!p.geo.dgrad) {
                        sy = a_y[i] * p.geo.stride - p.geo.pad + dy;
                        sx = a_x[i] * p.geo.stride - p.geo.pad + dx;
                    } else {
                        const int ty = a_y[i] + p.geo.pad - dy, tx = a_x[i] + p.geo.pad - dx;
                        sy = ty / p.geo.stride; sx = tx / p.geo.stride;
                        ok = ok && ty >= 0 && tx >= 0 && (sy * p.geo.stride == ty) && (sx * p.geo.stride == tx);
                    }
                    ok = ok && sy >= 0 && sy < p.geo.Hs && sx >= 0 && sx < p.geo.Ws;
                    const long long off = (((long long)a_n[i] * p.geo.Hs + sy) * p.geo.Ws + sx) * Cs + c0 + (f % KF) * 4;
                    ra[i] = ldv4(A + (ok ? off : 0)); ra_n[i] = ok ? 4 : 0;
                }
            } else {
                const int rlast = ((M + 3) & ~3) - 4;
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    const int f = tid + i * 256;
                    const int r = m0 + (f % (BM / 4)) * 4;
                    const int k = k0 + f / (BM / 4);
                    ra[i] = ldv4(A + (long long)min(k, K - 1) * p.lda + min(r, rlast)); ra_n[i] = (k < K) ? (M - r) : 0;
                }
            }
            // ------------------------------ B (vector, branch-free) ------------------------------
            if constexpr (BKD == VBG_OP_DENSE_K) {
                const int klast = ((K + 3) & ~3) - 4;
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int f = tid + i * 256;
                    const int kk = k0 + (f % KF) * 4;
                    const int col = min(n0 + f / KF, N - 1);
                    rb[i] = ldv4(B + (long long)col * p.ldb + min(kk, klast)); rb_n[i] = K - kk;
                }
            } else if constexpr (BKD == VBG_OP_DENSE_R) {
                const int clast = ((N + 3) & ~3) - 4;
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int f = tid + i * 256;
                    const int c = n0 + (f % (BN / 4)) * 4;
                    const int k = k0 + f / (BN / 4);
                    rb[i] = ldv4(B + (long long)min(k, K - 1) * p.ldb + min(c, clast)); rb_n[i] = (k < K) ? (N - c) : 0;
                }
            } else if constexpr (BKD == VBG_OP_WT_R) {
                const int Cout = p.geo.Cs;
                const int taps = p.geo.kh * p.geo.kw;
                const int tap = k0 / Cout;
                const int co0 = k0 - tap * Cout;
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int f = tid + i * 256;
                    const int c = n0 + (f % (BN / 4)) * 4;
                    const int co = co0 + f / (BN / 4);
                    rb[i] = ldv4(B + ((long long)co * taps + tap) * N + min(c, N - 4)); rb_n[i] = N - c;
                }
            } else {
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int f = tid + i * 256;
                    const int c = n0 + (f % (BN / 4)) * 4;
                    const int pix = k0 + f / (BN / 4);
                    const int pc = min(pix, K - 1);
                    const int px = pc % p.geo.Wr;
                    const int t = pc / p.geo.Wr;
                    const int py = t % p.geo.Hr;
                    const int pn = t / p.geo.Hr;
                    const int sy = py * p.geo.stride - p.geo.pad + b_dy[i];
                    const int sx = px * p.geo.stride - p.geo.pad + b_dx[i];
                    const bool ok = (pix < K) && (c < N) && sy >= 0 && sy < p.geo.Hs && sx >= 0 && sx < p.geo.Ws;
                    const long long off = (((long long)pn * p.geo.Hs + sy) * p.geo.Ws + sx) * p.geo.Cs + b_ci[i];
                    rb[i] = ldv4(B + (ok ? off : 0)); rb_n[i] = ok ? 4 : 0;
                }
            }
            return;
        }
        // ------------------------------ A ------------------------------
        if constexpr (AK == VBG_OP_DENSE_K) {
            // host normalises a_nseg >= 1 (segment 0 = {A, K, lda, 0} for the plain case)
            int seg = 0;
            while (seg + 1 < p.a_nseg && k0 >= p.a_seg_kend[seg]) ++seg;
            const float* base = p.a_seg_ptr[seg] + (A - p.A);          // + group offset
            const long long ld = p.a_seg_ld[seg];
            const int kbeg = (seg == 0) ? 0 : p.a_seg_kend[seg - 1];
            const int kend = (p.a_nseg == 1) ? K : p.a_seg_kend[seg];
            const int sh = p.a_seg_shift[seg];
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int f = tid + i * 256;
                const int k = k0 + (f % KF) * 4;
                long long row = m0 + f / KF;
                if (sh > 0) row = ((long long)a_n[i] * (p.a_H >> sh) + (a_y[i] >> sh)) * (p.a_W >> sh) + (a_x[i] >> sh);
                ra[i] = ld4(base + row * ld + (k - kbeg), p.a_vec, a_rv[i] ? (kend - k) : 0);
            }
        } else if constexpr (AK == VBG_OP_CONV_K) {
            const int Cs = p.geo.Cs;
            const int tap = k0 / Cs;
            const int c0 = k0 - tap * Cs;
            const int dy = tap / p.geo.kw, dx = tap - dy * p.geo.kw;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int f = tid + i * 256;
                int sy, sx;
                bool ok = a_rv[i];
                if (!p.geo.dgrad) {
                    sy = a_y[i] * p.geo.stride - p.geo.pad + dy;
                    sx = a_x[i] * p.geo.stride - p.geo.pad + dx;
                } else {
                    const int ty = a_y[i] + p.geo.pad - dy, tx = a_x[i] + p.geo.pad - dx;
                    ok = ok && ty >= 0 && tx >= 0;
                    if (p.geo.stride == 1) { sy = ty; sx = tx; }
                    else {
                        sy = ty / p.geo.stride; sx = tx / p.geo.stride;
                        ok = ok && (sy * p.geo.stride == ty) && (sx * p.geo.stride == tx);
                    }
                }
                ok = ok && sy >= 0 && sy < p.geo.Hs && sx >= 0 && sx < p.geo.Ws;
                const long long off = (((long long)a_n[i] * p.geo.Hs + sy) * p.geo.Ws + sx) * Cs + c0 + (f % KF) * 4;
                ra[i] = ld4(A + (ok ? off : 0), true, ok ? 4 : 0);
            }
        } else {  // VBG_OP_DENSE_R : elem(row, k) = A[k*lda + row]
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int f = tid + i * 256;
                const int r = m0 + (f % (BM / 4)) * 4;
                const int k = k0 + f / (BM / 4);
                ra[i] = ld4(A + (long long)k * p.lda + r, p.a_vec, (k < K) ? (M - r) : 0);
            }
        }
        if (p.a_prologue == 1) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                ra[i].x = fmaxf(ra[i].x, 0.f) * p.a_scale; ra[i].y = fmaxf(ra[i].y, 0.f) * p.a_scale;
                ra[i].z = fmaxf(ra[i].z, 0.f) * p.a_scale; ra[i].w = fmaxf(ra[i].w, 0.f) * p.a_scale;
            }
        }
        // ------------------------------ B ------------------------------
        if constexpr (BKD == VBG_OP_DENSE_K) {      // elem(col, k) = B[col*ldb + k]
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int f = tid + i * 256;
                const int k = k0 + (f % KF) * 4;
                const int col = n0 + f / KF;
                rb[i] = ld4(B + (long long)col * p.ldb + k, p.b_vec, (col < N) ? (K - k) : 0);
            }
        } else if constexpr (BKD == VBG_OP_DENSE_R) {   // elem(col, k) = B[k*ldb + col]
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int f = tid + i * 256;
                const int c = n0 + (f % (BN / 4)) * 4;
                const int k = k0 + f / (BN / 4);
                rb[i] = ld4(B + (long long)k * p.ldb + c, p.b_vec, (k < K) ? (N - c) : 0);
            }
        } else if constexpr (BKD == VBG_OP_WT_R) {     // dgrad weights: k = tap*Cout + co, col = ci
            const int Cout = p.geo.Cs;                  // gather source of A is dY: Cs == Cout
            const int taps = p.geo.kh * p.geo.kw;
            const int tap = k0 / Cout;
            const int co0 = k0 - tap * Cout;
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int f = tid + i * 256;
                const int c = n0 + (f % (BN / 4)) * 4;
                const int co = co0 + f / (BN / 4);
                rb[i] = ld4(B + ((long long)co * taps + tap) * N + c, p.b_vec, N - c);
            }
        } else {                                        // VBG_OP_CONV_R: k = pixel, col = (tap, ci)
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int f = tid + i * 256;
                const int c = n0 + (f % (BN / 4)) * 4;
                const int pix = k0 + f / (BN / 4);
                bool ok = (pix < K) && (c < N);
                const int px = pix % p.geo.Wr;
                const int t = pix / p.geo.Wr;
                const int py = t % p.geo.Hr;
                const int pn = t / p.geo.Hr;
                const int sy = py * p.geo.stride - p.geo.pad + b_dy[i];
                const int sx = px * p.geo.stride - p.geo.pad + b_dx[i];
                ok = ok && sy >= 0 && sy < p.geo.Hs && sx >= 0 && sx < p.geo.Ws;
                const long long off = (((long long)pn * p.geo.Hs + sy) * p.geo.Ws + sx) * p.geo.Cs + b_ci[i];
                rb[i] = ld4(B + (ok ? off : 0), true, ok ? 4 : 0);
            }
        }
    };

    auto store_tiles = [&](int buf) {
        float* as = As + buf * BK * SA;
        float* bs = Bs + buf * BK * SB;
        if constexpr (VEC) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                ra[i] = mask4(ra[i], ra_n[i]);
                if (p.a_prologue == 1) {
                    ra[i].x = fmaxf(ra[i].x, 0.f) * p.a_scale; ra[i].y = fmaxf(ra[i].y, 0.f) * p.a_scale;
                    ra[i].z = fmaxf(ra[i].z, 0.f) * p.a_scale; ra[i].w = fmaxf(ra[i].w, 0.f) * p.a_scale;
                }
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) rb[i] = mask4(rb[i], rb_n[i]);
        }
        if constexpr (A_KC) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int f = tid + i * 256;
                const int row = f / KF, k = (f % KF) * 4;
                as[(k + 0) * SA + row] = ra[i].x; as[(k + 1) * SA + row] = ra[i].y;
                as[(k + 2) * SA + row] = ra[i].z; as[(k + 3) * SA + row] = ra[i].w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int f = tid + i * 256;
                *reinterpret_cast<float4*>(&as[(f / (BM / 4)) * SA + (f % (BM / 4)) * 4]) = ra[i];
            }
        }
        if constexpr (B_KC) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int f = tid + i * 256;
                const int row = f / KF, k = (f % KF) * 4;
                bs[(k + 0) * SB + row] = rb[i].x; bs[(k + 1) * SB + row] = rb[i].y;
                bs[(k + 2) * SB + row] = rb[i].z; bs[(k + 3) * SB + row] = rb[i].w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int f = tid + i * 256;
                *reinterpret_cast<float4*>(&bs[(f / (BN / 4)) * SB + (f % (BN / 4)) * 4]) = rb[i];
            }
        }
    };

    // ---------------- main loop ---------------------------------------------------------
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lk = lane >> 5;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_tiles(kt0);
    store_tiles(0);
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = (kt - kt0) & 1;
        if (kt + 1 < kt1) load_tiles(kt + 1);
        const float* as = As + buf * BK * SA + wm * WM + lr + lk * SA;
        const float* bs = Bs + buf * BK * SB + wn * WN + lr + lk * SB;
        // fragments of k-step ks+1 are fetched while the MFMAs of k-step ks run (two register sets)
        float a0[TM], b0[TN], a1[TM], b1[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a0[i] = as[i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) b0[j] = bs[j * 32];
#pragma unroll
        for (int ks = 0; ks < BK / 2; ks += 2) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a1[i] = as[(ks + 1) * 2 * SA + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b1[j] = bs[(ks + 1) * 2 * SB + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b0[j], acc[i][j], 0, 0, 0);
            if (ks + 2 < BK / 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a0[i] = as[(ks + 2) * 2 * SA + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) b0[j] = bs[(ks + 2) * 2 * SB + j * 32];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b1[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < kt1) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // ---------------- epilogue ----------------------------------------------------------
    const bool add_bias = (bias != nullptr) && (split == 0);
    const bool atomic = p.accumulate && p.splitk > 1;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WN + j * 32 + lr;
        if (n >= N) continue;
        const float bv = add_bias ? bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (m >= M) continue;
                float v = acc[i][j][r] * p.alpha + bv;
                const long long o = (long long)m * p.ldc + n;
                if (atomic) {
                    unsafeAtomicAdd(C + o, v);
                } else if (p.accumulate) {
                    C[o] += v;                       // single owner per element: plain read-modify-write
                } else if (p.epi == VBG_EPI_RELU) {
                    C[o] = fmaxf(v, 0.f);
                } else if (p.epi == VBG_EPI_GELU_DUAL) {
                    C[o] = v;
                    C2[o] = gelu_erf(v);
                } else {
                    C[o] = v;
                }
            }
        }
    }
}

template <int BM, int BN, int BK, int AK, int BKD, bool VEC>
static void launch_one(const vbg_gemm_desc& d, int groups, int maxM, int maxN, hipStream_t s) {
    dim3 g(cdiv(maxM, BM), cdiv(maxN, BN), groups * d.splitk);
    VBG_LAUNCH((gemm_kernel<BM, BN, BK, AK, BKD, VEC>), g, dim3(256), 0, s, d);
}

// tile code: BM*1000+BN (128128, 128064, 64064); 0 = heuristic.  bk: 16 / 32; 0 = heuristic.
static void pick_tile(const vbg_gemm_desc& d, int groups, int maxM, int maxN, int& tile, int& bk) {
    tile = d.tile;
    bk = d.bk;
    if (tile == 64) tile = 64064;
    if (tile == 128) tile = 128128;
    if (tile == 0) {
        // measured on MI355X (tools/gemm_bench.py): 64x64 wins almost everywhere (8 waves/SIMD hide the LDS/barrier
        // phases); 128x128 only pays for very large M with N >= 256 (FPN merge / seg-head convs)
        const long t128 = (long)cdiv(maxM, 128) * cdiv(maxN, 128) * groups * d.splitk;
        tile = (maxN >= 256 && t128 >= 1536 && d.a_kind != VBG_OP_DENSE_R) ? 128128 : 64064;
    }
    if (bk == 0) bk = 32;
}

template <int AK, int BKD>
static int launch_pair(const vbg_gemm_desc& d, int groups, int maxM, int maxN, hipStream_t s) {
    int tile, bk;
    pick_tile(d, groups, maxM, maxN, tile, bk);
    if (!(d.a_vec && d.b_vec)) {                         // unaligned operands: general scalar-load path
        launch_one<64, 64, 16, AK, BKD, false>(d, groups, maxM, maxN, s);
    } else if (bk == 32) {
        if (tile == 128128) launch_one<128, 128, 32, AK, BKD, true>(d, groups, maxM, maxN, s);
        else if (tile == 128064) launch_one<128, 64, 32, AK, BKD, true>(d, groups, maxM, maxN, s);
        else launch_one<64, 64, 32, AK, BKD, true>(d, groups, maxM, maxN, s);
    } else {
        if (tile == 128128) launch_one<128, 128, 16, AK, BKD, true>(d, groups, maxM, maxN, s);
        else if (tile == 128064) launch_one<128, 64, 16, AK, BKD, true>(d, groups, maxM, maxN, s);
        else launch_one<64, 64, 16, AK, BKD, true>(d, groups, maxM, maxN, s);
    }
    VBG_LAUNCH_RET();
}

}  // namespace vbg

extern "C" int vbg_gemm(const vbg_gemm_desc* desc, void* stream) {
    using namespace vbg;
    VBG_CHECK_ARG(desc != nullptr);
    vbg_gemm_desc d = *desc;
    VBG_CHECK_ARG(d.A && d.B && d.C);
    VBG_CHECK_ARG(d.M >= 0 && d.N >= 0 && d.K >= 0);
    if (d.splitk < 1) d.splitk = 1;
    if (d.splitk > 1) VBG_CHECK_ARG(d.accumulate == 1);
    if (d.epi == VBG_EPI_GELU_DUAL) VBG_CHECK_ARG(d.C2 != nullptr);
    if (d.accumulate) VBG_CHECK_ARG(d.epi == VBG_EPI_NONE);
    VBG_CHECK_ARG(d.a_nseg >= 0 && d.a_nseg <= 4);
    if (d.a_nseg == 0) {
        d.a_nseg = 1; d.a_seg_ptr[0] = d.A; d.a_seg_kend[0] = d.K; d.a_seg_ld[0] = d.lda; d.a_seg_shift[0] = 0;
    }
    if (d.a_nseg > 1) {
        VBG_CHECK_ARG(d.a_kind == VBG_OP_DENSE_K && d.grp == nullptr && d.a_seg_kend[d.a_nseg - 1] == d.K);
        for (int i = 0; i < d.a_nseg; ++i) {
            VBG_CHECK_ARG(d.a_seg_kend[i] % 16 == 0 && d.a_seg_ptr[i] != nullptr);
            if (d.a_seg_kend[i] % 32 != 0) d.bk = 16;
            if (d.a_seg_shift[i] > 0) VBG_CHECK_ARG(d.a_H > 0 && d.a_W > 0);
        }
    }
    const int groups = d.grp ? d.ngroups : 1;
    VBG_CHECK_ARG(groups >= 1);
    const int maxM = d.grp ? d.grp_maxM : d.M, maxN = d.grp ? d.grp_maxN : d.N;
    if (maxM == 0 || maxN == 0 || groups == 0) return VBG_OK;
    if (d.K == 0 && !d.grp) return VBG_EARG;
    hipStream_t s = (hipStream_t)stream;
    const bool conv = d.a_kind == VBG_OP_CONV_K || d.b_kind == VBG_OP_CONV_R || d.b_kind == VBG_OP_WT_R;
    if (conv) {
        VBG_CHECK_ARG(d.geo.Cs % 16 == 0 && d.geo.kh > 0 && d.geo.kw > 0 && d.geo.stride > 0);
        if (d.geo.Cs % 32 != 0) d.bk = 16;      // a k-tile must stay inside one filter tap
        VBG_CHECK_ARG(d.grp == nullptr);
    }
    if (d.a_kind == VBG_OP_CONV_K) d.a_vec = ((uintptr_t)d.A % 16 == 0) && (d.geo.Cs % 4 == 0);
    if (d.b_kind == VBG_OP_CONV_R) d.b_vec = ((uintptr_t)d.B % 16 == 0) && (d.geo.Cs % 4 == 0);
    if (d.b_kind == VBG_OP_WT_R) d.b_vec = ((uintptr_t)d.B % 16 == 0) && (d.N % 4 == 0);
    if (d.a_kind == VBG_OP_DENSE_K && d.b_kind == VBG_OP_DENSE_K) return launch_pair<VBG_OP_DENSE_K, VBG_OP_DENSE_K>(d, groups, maxM, maxN, s);
    if (d.a_kind == VBG_OP_DENSE_K && d.b_kind == VBG_OP_DENSE_R) return launch_pair<VBG_OP_DENSE_K, VBG_OP_DENSE_R>(d, groups, maxM, maxN, s);
    if (d.a_kind == VBG_OP_DENSE_R && d.b_kind == VBG_OP_DENSE_R) return launch_pair<VBG_OP_DENSE_R, VBG_OP_DENSE_R>(d, groups, maxM, maxN, s);
    if (d.a_kind == VBG_OP_CONV_K && d.b_kind == VBG_OP_DENSE_K) return launch_pair<VBG_OP_CONV_K, VBG_OP_DENSE_K>(d, groups, maxM, maxN, s);
    if (d.a_kind == VBG_OP_CONV_K && d.b_kind == VBG_OP_WT_R) {
        VBG_CHECK_ARG(d.geo.dgrad == 1 && d.N % 4 == 0);
        return launch_pair<VBG_OP_CONV_K, VBG_OP_WT_R>(d, groups, maxM, maxN, s);
    }
    if (d.a_kind == VBG_OP_DENSE_R && d.b_kind == VBG_OP_CONV_R) {
        VBG_CHECK_ARG(d.N == d.geo.kh * d.geo.kw * d.geo.Cs);
        return launch_pair<VBG_OP_DENSE_R, VBG_OP_CONV_R>(d, groups, maxM, maxN, s);
    }
    return VBG_EARG;
}
