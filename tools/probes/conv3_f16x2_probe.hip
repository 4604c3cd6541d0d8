// PROBE (not part of libvbg): the row-reuse forward kernel of csrc/conv3.hip with TWO fp16 pieces per operand and three piece products
// (two accumulator sets), to measure what an arithmetic with half the matrix-core work would buy and cost.  Generated from conv3.hip by a
// script of the session that measured it; see DESIGN.md section 7.
#include "../../vibertgrid-pytorch_amd/csrc/vbg_common.h"
#include <type_traits>
#include <stdlib.h>

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
typedef _Float16 c3_f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 c3_f16x2 __attribute__((ext_vector_type(2)));
// two fp16 pieces of a pair of floats (a in the low half): hi = fp16(x) (toward zero), lo = fp16((x - hi) * 2^11): x = hi + lo * 2^-11 up to 2^-21 |x|
__device__ __forceinline__ void c3_split2(float a, float b, unsigned& hi, unsigned& lo) {
    const c3_f16x2 h = __builtin_amdgcn_cvt_pkrtz(a, b);
    const float ra = (a - (float)h[0]) * 2048.f, rb = (b - (float)h[1]) * 2048.f;
    const c3_f16x2 l = __builtin_amdgcn_cvt_pkrtz(ra, rb);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
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

// BM = 128 pixels per tile, or 64 for the late stages whose pixel count would leave half the chip without a 128-pixel tile
template <int BM>
__global__ __launch_bounds__(256, 2) void conv3x3_f16x2_kernel(const conv3_args p) {
    constexpr int BN = 128, NT = 256, SKH = 24;
    constexpr int NAI = BM * 4 / NT;                           // activation float4s per thread and super-tile
    constexpr int TM = BM / 64;                                // 32-row fragments per wave (waves 2 x 2: BM / 2 pixels x 64 filters each)
    constexpr int AROWS = BM + BM / 8;                         // + a zero pixel on either side of each image row (W >= 16)
    constexpr int PA = AROWS * SKH / 2, PB = BN * SKH / 2;     // one bf16 plane (dwords)
    constexpr int ASZ = 2 * PA, BSZ = 2 * PB;
    constexpr int CTS = BN + 4;
    constexpr int SMEM = 2 * (ASZ + BSZ) > BM * (BN + 4) ? 2 * (ASZ + BSZ) : BM * (BN + 4);                      // 76 KB at BM = 128: two workgroups per CU
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
    int a_y[NAI], a_x[NAI], a_lrow[NAI];
    unsigned avo[NAI], bvo[2];
#pragma unroll
    for (int i = 0; i < NAI; ++i) {
        const int r = (tid + i * NT) >> 2;
        const int pix = p0 + r;
        a_x[i] = pix & (W - 1);
        a_y[i] = pix >> wsh;
        a_lrow[i] = r + 1 + 2 * (r >> wsh);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (tid + i * NT) >> 2;
        bvo[i] = (n0 + r < N) ? (unsigned)((r * K + kc) * 4) : C3_INVALID;
    }
    // image rows wider than the tile (W = 256, 512, ...: the tile is a piece of ONE row): the pixels left and right of it are real
    // pixels, not padding -- eight lanes fetch them per super-tile into the two halo rows of the LDS image
    const bool wide = W > BM;
    const int h_side = (tid >> 2) & 1, h_x = (p0 & (W - 1)) + (h_side ? BM : -1), h_y = p0 >> wsh, h_row = h_side ? BM + 1 : 0;
    unsigned hvo = C3_INVALID;
    float4 rh = make_float4(0.f, 0.f, 0.f, 0.f);
    auto set_a = [&](int kh) {
#pragma unroll
        for (int i = 0; i < NAI; ++i) {
            const int sy = a_y[i] + kh - 1;
            avo[i] = ((unsigned)sy < (unsigned)H) ? (unsigned)(((sy * W + a_x[i]) * Cs + kc) * 4) : C3_INVALID;
        }
        const int sy = h_y + kh - 1;
        hvo = (wide && tid < 8 && (unsigned)sy < (unsigned)H && (unsigned)h_x < (unsigned)W) ? (unsigned)(((sy * W + h_x) * Cs + kc) * 4) : C3_INVALID;
    };
    const float* const wbase = p.Wt + (long long)n0 * K;
    float4 ra[NAI], rb[2];
    int a_kh = 0, a_c0 = 0;                    // next activation super-tile (filter row, channel chunk) to load
    int b_kh = 0, b_c0 = 0, b_kw = 0;          // next weight tile to load: order (kh, chunk, kw)
    auto load_a = [&]() {
        const __amdgpu_buffer_rsrc_t r = c3_rsrc(a_img + a_c0);
#pragma unroll
        for (int i = 0; i < NAI; ++i) ra[i] = c3_load(r, avo[i]);
        if (wide) rh = c3_load(r, hvo);
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
        uint2 h, l;
        c3_split2(v.x, v.y, h.x, l.x);
        c3_split2(v.z, v.w, h.y, l.y);
        *reinterpret_cast<uint2*>(&dst[o]) = h;
        *reinterpret_cast<uint2*>(&dst[o + PL]) = l;
    };
    auto store_a = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NAI; ++i) store4(As + buf * ASZ, PA, a_lrow[i], ra[i]);
        if (wide && tid < 8) store4(As + buf * ASZ, PA, h_row, rh);
    };
    auto store_b = [&](int buf) {
        store4(Bs + buf * BSZ, PB, tid >> 2, rb[0]);
        store4(Bs + buf * BSZ, PB, (tid + NT) >> 2, rb[1]);
    };

    // ---------------- main loop ---------------------------------------------------------------
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lk = lane >> 5;
    f32x16 acc[TM][2], acx[TM][2];       // main (hi hi) and cross (hi lo + lo hi, scaled by 2^11) sums
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = acx[i][j][r] = 0.f;
    int arow[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * (BM / 2) + i * 32 + lr;
        arow[i] = r + 2 * (r >> wsh);                         // LDS row of pixel x - 1 (tap kw adds kw)
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
        c3_u32x4 fa[2][TM], fb[2][2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[q][i] = as[q * (PA / 4) + (arow[i] + kw) * (SKH / 8) + lk];
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[q][j] = bs[q * (PB / 4) + (brow + j * 32) * (SKH / 8) + lk];
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MORE) load_b();
        if constexpr (LOADA) load_a();
        __builtin_amdgcn_sched_barrier(0);
        // piece products: (lo,hi) and (hi,lo) into the cross sums, (hi,hi) into the main sums
        auto mma_range = [&](auto t0_tag, auto t1_tag) {
#pragma unroll
            for (int t = decltype(t0_tag)::value; t < decltype(t1_tag)::value; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        if (t == 0) acx[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(c3_f16x8, fa[1][i]), __builtin_bit_cast(c3_f16x8, fb[0][n]), acx[i][n], 0, 0, 0);
                        if (t == 1) acx[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(c3_f16x8, fa[0][i]), __builtin_bit_cast(c3_f16x8, fb[1][n]), acx[i][n], 0, 0, 0);
                        if (t == 2) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(c3_f16x8, fa[0][i]), __builtin_bit_cast(c3_f16x8, fb[0][n]), acc[i][n], 0, 0, 0);
                    }
        };
        using i0 = std::integral_constant<int, 0>;
        using ih = std::integral_constant<int, 1>;
        using i1 = std::integral_constant<int, 3>;
        if constexpr (MORE) {
            mma_range(i0{}, ih{});
            __builtin_amdgcn_sched_barrier(0);
            mma_range(ih{}, i1{});
            store_b(bbuf ^ 1);
            if constexpr (LOADA) store_a(abuf ^ 1);
            constexpr int NL = LOADA ? 2 + NAI : 2;
            c3_pipe<0, 4 * TM, NL * 10, NL * 2>::run();
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
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Ct[(wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * CTS + wn * 64 + j * 32 + lr] = acc[i][j][r] + acx[i][j][r] * (1.f / 2048.f);
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


}  // namespace vbg

extern "C" int probe_conv3x3_f16x2(const float* x, const float* w, float* y, int B, int H, int W, int Cs, int N, void* stream) {
    vbg::conv3_args a;
    a.X = x; a.Wt = w; a.bias = nullptr; a.Y = y; a.stats = nullptr; a.stats_slots = 0;
    a.H = H; a.W = W; a.wsh = 31 - __builtin_clz((unsigned)W); a.Cs = Cs; a.N = N;
    const long long M = (long long)B * H * W;
    a.M = (int)M; a.accumulate = 0;
    hipLaunchKernelGGL(vbg::conv3x3_f16x2_kernel<128>, dim3((unsigned)(M / 128), (unsigned)((N + 127) / 128), 1), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}
