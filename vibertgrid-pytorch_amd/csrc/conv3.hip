// 3x3 / stride 1 / pad 1 convolution over NHWC activations with ROW REUSE across the three horizontal taps (gfx950).
//
// Replaces, for the wide trunk / FPN / segmentation-head convolutions (reference model/ResNetFPN_ViBERTgrid.py:478-508, 612-648 and
// model/semantic_segmentation_head.py conv stacks, torch.nn.Conv2d(k=3, s=1, p=1)), the generic implicit GEMM of gemm.hip, whose
// k-loop fetches and splits the 128 x 16 activation tile once per filter tap: nine times per (row of taps x channel chunk).  Every
// matrix kernel of this library levels off where a CU ingests ~12.5 B / clk from L2, so the lever is bytes per
// product.  Here an output tile is 128 consecutive pixels = whole image rows (W in {32, 64, 128}); for a filter row kh and a chunk
// of 16 input channels the source row y + kh - 1 is loaded and split ONCE into an LDS image that carries one zero pixel on either
// side of each image row, and the three taps kw = 0, 1, 2 read their MFMA fragments from it at row offsets 0, 1, 2 (a uniform
// shift keeps the bank pattern of the [row][16 + 8] bf16 layout).  Activation traffic and split work per tap drop to a third;
// the weight tile [128 x 16] is fetched per tap as before.  Arithmetic is the library's fp32-grade form: exact three-way bf16
// split, six v_mfma_f32_32x32x16_bf16 piece products smallest first, fp32 accumulation -- the same products in the same order per
// (tap, chunk) as gemm.hip, only the order of the taps inside the reduction differs ((kh, chunk, kw) instead of (kh, kw, chunk)).
//
// The data gradient of such a convolution is the same convolution of dy with the filter turned by 180 degrees and its channel
// roles swapped: vbg_conv3x3_wflip writes that filter (2.4 MB at 256 x 256) and the backward calls this kernel again.
#include "vbg_common.h"
#include <type_traits>
#include "../../include/vbg.h"

namespace vbg {

typedef unsigned c3_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 c3_bf16x8 __attribute__((ext_vector_type(8)));
constexpr unsigned C3_INVALID = 0x80000000u;       // outside every descriptor: the load returns 0 without touching memory

struct conv3_args {
    const float* X;        // [B, H, W, Cs]
    const float* Wt;       // [N, 3, 3, Cs]
    const float* bias;     // [N] or null
    float* Y;              // [B, H, W, N]
    double* stats;         // BatchNorm slot workspace [slots][2][N] or null
    int stats_slots;
    int H, W, wsh, Cs, N, M;
    int accumulate;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t c3_rsrc(const float* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)0x80000000u, 0x00020000);
}
__device__ __forceinline__ float4 c3_load(__amdgpu_buffer_rsrc_t r, unsigned vo) {
    const c3_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)vo, 0, 0);
    return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}
// exact three-way split of a pair of floats into packed bf16 pairs (a in the low half), by truncation
__device__ __forceinline__ void c3_split3(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
    const float ra = a - __uint_as_float(__float_as_uint(a) & 0xffff0000u), rb = b - __uint_as_float(__float_as_uint(b) & 0xffff0000u);
    const float sa = ra - __uint_as_float(__float_as_uint(ra) & 0xffff0000u), sb = rb - __uint_as_float(__float_as_uint(rb) & 0xffff0000u);
    hi = __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
    mid = __builtin_amdgcn_perm(__float_as_uint(rb), __float_as_uint(ra), 0x07060302u);
    lo = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
}
// after every MFMA its share of the NV VALU and ND LDS-write instructions of the region
template <int M, int NM, int NV, int ND>
struct c3_pipe {
    static __device__ __forceinline__ void run() {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        constexpr int v = ((M + 1) * NV) / NM - (M * NV) / NM;
        if constexpr (v > 0) __builtin_amdgcn_sched_group_barrier(0x002, v, 0);
        constexpr int w = ((M + 1) * ND) / NM - (M * ND) / NM;
        if constexpr (w > 0) __builtin_amdgcn_sched_group_barrier(0x200, w, 0);
        if constexpr (M + 1 < NM) c3_pipe<M + 1, NM, NV, ND>::run();
    }
};

__global__ __launch_bounds__(256, 2) void conv3x3_kernel(const conv3_args p) {
    constexpr int BM = 128, BN = 128, NT = 256, SKH = 24;
    constexpr int AROWS = BM + 8;                              // + a zero pixel on either side of each of up to 4 image rows
    constexpr int PA = AROWS * SKH / 2, PB = BN * SKH / 2;     // one bf16 plane (dwords)
    constexpr int ASZ = 3 * PA, BSZ = 3 * PB;
    constexpr int CTS = BN + 4;
    constexpr int SMEM = 2 * (ASZ + BSZ);                      // 76 KB: two workgroups per CU
    static_assert(BM * CTS <= SMEM, "staged output tile");
    __shared__ __attribute__((aligned(16))) unsigned smem[SMEM];
    unsigned* const As = smem;
    unsigned* const Bs = smem + 2 * ASZ;

    const int tid = threadIdx.x;
    // XCD-aware block -> tile map (same as gemm.hip: XCD k owns the k-th eighth of the tile sequence, bands of 8 row tiles)
    constexpr unsigned XCDS = 8, XCD_GROUP = 8;
    const unsigned gx = gridDim.x, gy = gridDim.y;
    const unsigned lin = blockIdx.x + gx * blockIdx.y;
    const unsigned total = gx * gy;
    const unsigned xcd = lin % XCDS, local = lin / XCDS;
    const unsigned per_xcd = (total + XCDS - 1) / XCDS, tall = (total % XCDS) ? (total % XCDS) : XCDS;
    const unsigned pid = xcd < tall ? xcd * per_xcd + local : tall * per_xcd + (xcd - tall) * (per_xcd - 1) + local;
    const unsigned band = XCD_GROUP * gy, bid = pid / band, first = bid * XCD_GROUP;
    const unsigned bm = min(gx - first, XCD_GROUP), inb = pid - bid * band;
    const unsigned tile_m = first + inb % bm, tile_n = inb / bm;
    const int H = p.H, W = p.W, wsh = p.wsh, Cs = p.Cs, N = p.N;
    const int K = 9 * Cs;
    const int m0 = (int)tile_m * BM, n0 = (int)tile_n * BN;

    // ---------------- loader state -------------------------------------------------------------
    const int HW = H * W;
    const int nb = m0 / HW, p0 = m0 - nb * HW;                 // the tile lies inside one image (H*W % 128 == 0)
    const float* const a_img = p.X + (long long)nb * HW * Cs;
    const int kc = (tid & 3) * 4;                              // this thread's 4 channels of a 16-channel chunk
    int a_y[2], a_x[2], a_lrow[2];
    unsigned avo[2], bvo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (tid + i * NT) >> 2;
        const int pix = p0 + r;
        a_x[i] = pix & (W - 1);
        a_y[i] = pix >> wsh;
        a_lrow[i] = r + 1 + 2 * (r >> wsh);
        bvo[i] = (n0 + r < N) ? (unsigned)((r * K + kc) * 4) : C3_INVALID;
    }
    auto set_a = [&](int kh) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int sy = a_y[i] + kh - 1;
            avo[i] = ((unsigned)sy < (unsigned)H) ? (unsigned)(((sy * W + a_x[i]) * Cs + kc) * 4) : C3_INVALID;
        }
    };
    const float* const wbase = p.Wt + (long long)n0 * K;
    float4 ra[2], rb[2];
    int a_kh = 0, a_c0 = 0;                    // next activation super-tile (filter row, channel chunk) to load
    int b_kh = 0, b_c0 = 0, b_kw = 0;          // next weight tile to load: order (kh, chunk, kw)
    auto load_a = [&]() {
        const __amdgpu_buffer_rsrc_t r = c3_rsrc(a_img + a_c0);
        ra[0] = c3_load(r, avo[0]);
        ra[1] = c3_load(r, avo[1]);
        a_c0 += 16;
        if (a_c0 >= Cs) { a_c0 = 0; ++a_kh; set_a(a_kh); }
    };
    auto load_b = [&]() {
        const __amdgpu_buffer_rsrc_t r = c3_rsrc(wbase + (b_kh * 3 + b_kw) * Cs + b_c0);
        rb[0] = c3_load(r, bvo[0]);
        rb[1] = c3_load(r, bvo[1]);
        if (++b_kw == 3) {
            b_kw = 0; b_c0 += 16;
            if (b_c0 >= Cs) { b_c0 = 0; ++b_kh; }
        }
    };
    auto store4 = [&](unsigned* dst, int PL, int row, const float4& v) {
        const int o = row * (SKH / 2) + kc / 2;
        uint2 h, m, l;
        c3_split3(v.x, v.y, h.x, m.x, l.x);
        c3_split3(v.z, v.w, h.y, m.y, l.y);
        *reinterpret_cast<uint2*>(&dst[o]) = h;
        *reinterpret_cast<uint2*>(&dst[o + PL]) = m;
        *reinterpret_cast<uint2*>(&dst[o + 2 * PL]) = l;
    };
    auto store_a = [&](int buf) {
        store4(As + buf * ASZ, PA, a_lrow[0], ra[0]);
        store4(As + buf * ASZ, PA, a_lrow[1], ra[1]);
    };
    auto store_b = [&](int buf) {
        store4(Bs + buf * BSZ, PB, tid >> 2, rb[0]);
        store4(Bs + buf * BSZ, PB, (tid + NT) >> 2, rb[1]);
    };

    // ---------------- main loop ---------------------------------------------------------------
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lk = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int arow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r0 = wm * 64 + i * 32;
        arow[i] = r0 + lr + 2 * (r0 >> wsh);                  // LDS row of pixel x - 1 (tap kw adds kw)
    }
    const int brow = wn * 64 + lr;

    using yes_t = std::integral_constant<bool, true>;
    using no_t = std::integral_constant<bool, false>;
    // one (tap, chunk) k-tile: fragments of this tile from LDS -> loads of the next tile -> first third of the piece products (covers
    // the load latency) -> the rest of the products with the split + LDS writes of the loaded tile in their gaps -> barrier
    auto k_tile = [&](auto loada_tag, auto more_tag, int abuf, int bbuf, int kw) {
        constexpr bool LOADA = decltype(loada_tag)::value, MORE = decltype(more_tag)::value;
        const c3_u32x4* as = reinterpret_cast<const c3_u32x4*>(As + abuf * ASZ);
        const c3_u32x4* bs = reinterpret_cast<const c3_u32x4*>(Bs + bbuf * BSZ);
        c3_u32x4 fa[3][2], fb[3][2];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[q][i] = as[q * (PA / 4) + (arow[i] + kw) * (SKH / 8) + lk];
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[q][j] = bs[q * (PB / 4) + (brow + j * 32) * (SKH / 8) + lk];
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MORE) load_b();
        if constexpr (LOADA) load_a();
        __builtin_amdgcn_sched_barrier(0);
        // piece products, smallest first: (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi)
        constexpr int qa[6] = {2, 0, 1, 1, 0, 0}, qb[6] = {0, 2, 1, 0, 1, 0};
        auto mma_range = [&](auto t0_tag, auto t1_tag) {
#pragma unroll
            for (int t = decltype(t0_tag)::value; t < decltype(t1_tag)::value; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(c3_bf16x8, fa[qa[t]][i]),
                                                                            __builtin_bit_cast(c3_bf16x8, fb[qb[t]][n]), acc[i][n], 0, 0, 0);
        };
        using i0 = std::integral_constant<int, 0>;
        using ih = std::integral_constant<int, 2>;
        using i1 = std::integral_constant<int, 6>;
        if constexpr (MORE) {
            mma_range(i0{}, ih{});
            __builtin_amdgcn_sched_barrier(0);
            mma_range(ih{}, i1{});
            store_b(bbuf ^ 1);
            if constexpr (LOADA) store_a(abuf ^ 1);
            constexpr int NL = LOADA ? 4 : 2;
            c3_pipe<0, 16, NL * 22, NL * 3>::run();
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        } else {
            mma_range(i0{}, i1{});
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // the zero pixels beside the image rows are written once; the loop only ever stores the pixel rows
    for (int e = tid; e < 2 * ASZ / 4; e += NT) reinterpret_cast<uint4*>(As)[e] = make_uint4(0u, 0u, 0u, 0u);
    set_a(0);
    load_a();
    load_b();
    __syncthreads();
    store_a(0);
    store_b(0);
    __syncthreads();
    const int nsup = 3 * Cs / 16;
    int t = 0;
    for (int s = 0; s + 1 < nsup; ++s) {
        k_tile(no_t{}, yes_t{}, s & 1, t & 1, 0); ++t;
        k_tile(no_t{}, yes_t{}, s & 1, t & 1, 1); ++t;
        k_tile(yes_t{}, yes_t{}, s & 1, t & 1, 2); ++t;
    }
    {
        const int s = nsup - 1;
        k_tile(no_t{}, yes_t{}, s & 1, t & 1, 0); ++t;
        k_tile(no_t{}, yes_t{}, s & 1, t & 1, 1); ++t;
        k_tile(no_t{}, no_t{}, s & 1, t & 1, 2);
    }

    // ---------------- epilogue: staged through LDS, float4 row pieces, optional bias / accumulate / BatchNorm statistics ---------
    const float* bias = p.bias;
    const int accumulate = p.accumulate;
    __syncthreads();
    float* const Ct = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Ct[(wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * CTS + wn * 64 + j * 32 + lr] = acc[i][j][r];
    __syncthreads();
    constexpr int QN = BN / 4;
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f), cq = cs;
#pragma unroll 4
    for (int q = 0; q < BM * QN / NT; ++q) {
        const int idx = tid + q * NT;
        const int row = idx / QN, c = (idx % QN) * 4;
        const int gm = m0 + row, gn = n0 + c;
        if (gn >= N) continue;                                 // (N % 4 == 0: a float4 is inside or outside)
        float4 v = *reinterpret_cast<const float4*>(&Ct[row * CTS + c]);
        if (bias) { v.x += bias[gn]; v.y += bias[gn + 1]; v.z += bias[gn + 2]; v.w += bias[gn + 3]; }
        float* cp = p.Y + (long long)gm * N + gn;
        if (p.stats) {
            cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
            cq.x += v.x * v.x; cq.y += v.y * v.y; cq.z += v.z * v.z; cq.w += v.w * v.w;
        }
        if (accumulate) {
            const float4 o = *reinterpret_cast<const float4*>(cp);
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        *reinterpret_cast<float4*>(cp) = v;
    }
    if (p.stats) {
        constexpr int RG = NT / QN;
        float* red = reinterpret_cast<float*>(smem);
        __syncthreads();
        const int rg = tid / QN, c = (tid % QN) * 4;
        *reinterpret_cast<float4*>(&red[(0 * RG + rg) * BN + c]) = cs;
        *reinterpret_cast<float4*>(&red[(1 * RG + rg) * BN + c]) = cq;
        __syncthreads();
        for (int e = tid; e < 2 * BN; e += NT) {
            const int st = e / BN, col = e - st * BN;
            if (n0 + col >= N) continue;
            double acc64 = 0.0;
            for (int g2 = 0; g2 < RG; ++g2) acc64 += (double)red[(st * RG + g2) * BN + col];
            unsafeAtomicAdd(p.stats + (size_t)(tile_m % (unsigned)p.stats_slots) * 2 * N + (size_t)st * N + n0 + col, acc64);
        }
    }
}

// out[ci][2 - kh][2 - kw][co] = w[co][kh][kw][ci]: the filter of the data gradient (32 x 32 LDS transposes per tap)
__global__ void conv3_wflip_kernel(const float* __restrict__ w, int Cout, int Cin, float* __restrict__ out) {
    __shared__ float t[32][33];
    const int tap = blockIdx.z, co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
    const int tx = threadIdx.x, ty = threadIdx.y;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int co = co0 + ty + 8 * j, ci = ci0 + tx;
        t[ty + 8 * j][tx] = (co < Cout && ci < Cin) ? w[((long long)co * 9 + tap) * Cin + ci] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ci = ci0 + ty + 8 * j, co = co0 + tx;
        if (ci < Cin && co < Cout) out[((long long)ci * 9 + (8 - tap)) * Cout + co] = t[tx][ty + 8 * j];
    }
}

}  // namespace vbg

extern "C" int vbg_conv3x3(const float* x, const float* w, const float* bias, float* y, double* stats, int stats_slots, int B, int H,
                           int W, int Cs, int N, int accumulate, void* stream) {
    VBG_CHECK_ARG(x && w && y && B > 0 && H > 0);
    VBG_CHECK_ARG(W == 32 || W == 64 || W == 128);
    VBG_CHECK_ARG(((long long)H * W) % 128 == 0 && (long long)H * W * Cs < (1ll << 29));
    VBG_CHECK_ARG(Cs >= 16 && Cs % 16 == 0 && N >= 4 && N % 4 == 0);
    VBG_CHECK_ARG((((uintptr_t)x) & 15) == 0 && (((uintptr_t)w) & 15) == 0 && (((uintptr_t)y) & 15) == 0);
    VBG_CHECK_ARG(!stats || (stats_slots >= 1 && !accumulate));
    vbg::conv3_args a;
    a.X = x; a.Wt = w; a.bias = bias; a.Y = y; a.stats = stats; a.stats_slots = stats_slots;
    a.H = H; a.W = W; a.wsh = W == 32 ? 5 : (W == 64 ? 6 : 7); a.Cs = Cs; a.N = N;
    const long long M = (long long)B * H * W;
    VBG_CHECK_ARG(M < (1ll << 31));
    a.M = (int)M; a.accumulate = accumulate;
    dim3 g((unsigned)(M / 128), (unsigned)vbg::cdiv(N, 128), 1);
    VBG_LAUNCH(vbg::conv3x3_kernel, g, dim3(256), 0, (hipStream_t)stream, a);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_conv3x3_wflip(const float* w, int Cout, int Cin, float* out, void* stream) {
    VBG_CHECK_ARG(w && out && Cout > 0 && Cin > 0);
    dim3 g((unsigned)vbg::cdiv(Cin, 32), (unsigned)vbg::cdiv(Cout, 32), 9);
    VBG_LAUNCH(vbg::conv3_wflip_kernel, g, dim3(32, 8), 0, (hipStream_t)stream, w, Cout, Cin, out);
    VBG_LAUNCH_RET();
}
