// MFMA GEMM / implicit-GEMM convolution family for gfx950 (MI355X).
//
// One kernel template computes  C[M,N] (+)= opA[M,K] * opB[K,N]  on fp32 tensors in one of three arithmetic forms (PREC):
//   0  v_mfma_f32_32x32x2_f32, exact fp32 products (bitwise an fmaf chain, 157 TF peak);
//   3  fp32-grade on the bf16 matrix cores: exact three-way bf16 split of every operand element on its way into LDS, six
//      v_mfma_f32_32x32x16_bf16 piece products per k-step, fp32 accumulation (2500 / 6 = 417 TF fp32-equivalent peak) -- default;
//   1  amp: operands rounded to bf16, one bf16 MFMA per k-step;
// for every contraction on the
// ViBERTgrid hot path: BERT linears and attention products (reference: transformers BertModel
// called at model/BERTgrid_generator.py:134), the ResNet-FPN convolutions as implicit GEMM over
// NHWC activations (model/ResNetFPN_ViBERTgrid.py:478-508, 612-648), the concat-free early /
// late / P_fuse fusions (K segments read from several tensors in place; :315-321, :502-506,
// model/field_type_classification_head.py:181-188) and all their dgrad / wgrad products.
//
// Structure (256 threads = 4 waves in 2x2, block tile BMxBN, k-tile BK, LDS double buffered, one barrier per k-tile):
//  * block -> tile: XCD-aware (workgroup b runs on XCD b % 8; each XCD walks its own band of tiles so its 4 MB L2 keeps the
//    row / column panels it is using);
//  * global -> registers: BUFFER loads with a scalar descriptor base advanced per k-tile and per-thread byte offsets that stay
//    constant from tile to tile; lanes that must read zero carry an out-of-range offset.  VALU instructions share the issue port
//    with the MFMAs (tools/mfma_lds.hip), so the steady-state loop has ~4 of them per 16 MFMAs; the loads of tile t+1 are issued
//    after the first MFMA group of tile t and written to LDS before its last group;
//  * registers -> LDS: one 16-byte ds_write per float4 for both layouts.  K-contiguous operands are stored row-major [row][BK+4]:
//    the 144-/80-byte row stride puts 16 consecutive rows on 16 different 16-byte slots, so the fragment read is ONE conflict-free
//    ds_read_b128 per 32 rows per 8 k (the MFMA k index is permuted identically for A and B: lanes 0-31 take k = 8g+j, lanes
//    32-63 k = 8g+4+j at step j).  Row-contiguous operands are stored [k][rows+4] and read with ds_read_b32;
//  * the reduction tail (K % BK != 0) and the last tile run in peeled copies of the k-tile body, the loop itself holds one copy;
//  * epilogue: plain and single-owner accumulating stores are staged through the idle operand tiles in LDS and leave as float4
//    row pieces; split-K partial sums use float atomics.
#include "vbg_common.h"
#include <type_traits>
#include <hip/hip_ext.h>
#include "../../include/vbg.h"

#ifndef VBG_PIPE2_X16
#define VBG_PIPE2_X16 0
#endif

namespace vbg {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// Every global read of the k-loop is a BUFFER load: address = descriptor base (SGPRs, advanced per k-tile with
// scalar adds) + a per-thread byte offset that does not change from tile to tile.  VALU instructions issue through
// the same port as the MFMAs (tools/mfma_lds.hip: +80 VALU per 16 MFMA costs 20 %), so the loop keeps per-thread
// address arithmetic, clamping and zero-masking out of it: a lane that must read zero (row outside the operand,
// convolution padding, reduction tail) carries VO_INVALID, which is outside every descriptor -> the load returns 0.
constexpr unsigned VO_INVALID = 0x80000000u;
constexpr long long NREC_MAX = 0x80000000ll;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, long long nbytes) {
    const long long n = nbytes < 0 ? 0 : (nbytes < NREC_MAX ? nbytes : NREC_MAX);      // (a tile past the end of the reduction: nothing valid)
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)(unsigned)n, 0x00020000);
}
// one float4 of an operand: a single 16-byte load (VEC) or four dword loads with their own validity (NE = 4)
template <int NE>
__device__ __forceinline__ float4 bload(__amdgpu_buffer_rsrc_t r, const unsigned (&vo)[NE], unsigned inv = 0) {
    if constexpr (NE == 1) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(vo[0] | inv), 0, 0);     // inv = VO_INVALID: the whole tile reads 0
        return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
    } else {
        float4 o;
        o.x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)vo[0], 0, 0));
        o.y = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)vo[1], 0, 0));
        o.z = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)vo[2], 0, 0));
        o.w = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)vo[3], 0, 0));
        return o;
    }
}
// keep the first `nvalid` of 4 elements (branch-free: v_cndmask); only used on the reduction-tail tile
__device__ __forceinline__ float4 mask4(float4 v, int nvalid) {
    v.x = nvalid > 0 ? v.x : 0.f; v.y = nvalid > 1 ? v.y : 0.f;
    v.z = nvalid > 2 ? v.z : 0.f; v.w = nvalid > 3 ? v.w : 0.f;
    return v;
}
// byte offsets of the NE pieces of a float4 that starts `elem` floats behind the descriptor base
template <int NE>
__device__ __forceinline__ void set_vo(unsigned (&vo)[NE], long long elem, bool ok, int nvalid = 4) {
#pragma unroll
    for (int e = 0; e < NE; ++e) vo[e] = (ok && (NE == 1 || e < nvalid)) ? (unsigned)((elem + e) * 4) : VO_INVALID;
}

// two floats -> two bf16 (round to nearest even) in one dword, `lo` in the low half
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// scheduling pipeline of a bf16 k-tile body: after every MFMA its share of the NV VALU and ND LDS-write instructions of the region
template <int M, int NM, int NV, int ND>
struct sched_pipe {
    static __device__ __forceinline__ void run() {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        constexpr int v = ((M + 1) * NV) / NM - (M * NV) / NM;
        if constexpr (v > 0) __builtin_amdgcn_sched_group_barrier(0x002, v, 0);
        constexpr int w = ((M + 1) * ND) / NM - (M * ND) / NM;
        if constexpr (w > 0) __builtin_amdgcn_sched_group_barrier(0x200, w, 0);
        if constexpr (M + 1 < NM) sched_pipe<M + 1, NM, NV, ND>::run();
    }
};

// NT = threads per block (256: 4 waves in 2x2, LDS double buffered, one barrier per k-tile).
// PREC 1 ("amp"): the operands stay fp32 in HBM, are rounded to bf16 on their way into LDS and multiplied with
// v_mfma_f32_32x32x16_bf16 (fp32 accumulate, 16x the fp32 matrix rate); loaders and epilogue are shared with the fp32 form.
// PREC 3: fp32-grade products on the bf16 matrix cores.  Every operand element is split EXACTLY into three bf16 pieces
// (x = hi + mid + lo: 8 + 8 + 8 significant bits, by truncation) held as three planes of the LDS tile, and a product is the sum
// of the six piece products of order <= 2^-16 (hh, hm, mh, hl, lh, mm; each exact in fp32, accumulated in fp32).  The dropped
// terms (ml, lm, ll) are <= 2^-23 of the product -- the size of the fp32 rounding of the product itself.
template <int BM, int BN, int BK, int NT, int AK, int BKD, bool VEC, int PREC = 0>
__global__ __launch_bounds__(NT) void gemm_kernel(const vbg_gemm_desc p) {
    constexpr bool HB = PREC != 0;
    constexpr bool PIPE2_X16 = VBG_PIPE2_X16;      // 16-deep bf16 k-tiles on the two-tiles-in-flight loop as well (measured, see below)
    // PREC 2 (round 6): fp32-grade products on the fp16 matrix cores for operands inside fp16's range (FORWARD products: activations and
    // weights) -- two pieces per operand, hi = fp16(x), lo' = fp16((x - hi) 2^11), both rounded to nearest, three piece products (hi hi into
    // the main accumulators, lo' hi + hi lo' into a second set that is scaled by 2^-11 once, behind the loop): the arithmetic of
    // csrc/gemm_planes.hip FORM 1 and of the row-reuse convolutions, with the split done here, in registers.  Half the matrix-core work of PREC 3.
    constexpr int NP = PREC == 3 ? 3 : (PREC == 2 ? 2 : 1);      // 16-bit planes per operand tile
    constexpr int WGM = 2, WGN = 2;
    constexpr int NBUF = 2;
    constexpr int NE = VEC ? 1 : 4;            // separately addressed pieces per float4
    constexpr int KF = BK / 4;                 // float4 chunks per row of a K-contiguous tile
    constexpr int NG = BK / 8;                 // k-groups (8 k = 4 MFMA steps) per tile
    constexpr bool A_KC = (AK == VBG_OP_DENSE_K || AK == VBG_OP_CONV_K);
    constexpr bool B_KC = (BKD == VBG_OP_DENSE_K);
    constexpr int SKR = BK + 4;                // row stride of a row-major (K-contiguous) LDS tile
    constexpr int SA = BM + 4, SB = BN + 4;    // k-row stride of a k-major (row-contiguous) LDS tile
    constexpr int SKH = BK + 8;                // HB: bf16 row stride; both operands are K-contiguous [row][BK+8] in LDS
    // Row-contiguous operands keep their tile rows in a permuted order in the bf16 forms: tile row r lives at LDS row
    // (r % 4) * PS + r / 4 with PS = rows / 4 + 4.  A lane's four rows (r .. r+3, one float4 along the rows) then go to four
    // row groups 16*k banks apart and the 16 lanes of a k row walk the banks in steps of 20 (2-way at worst, free for
    // ds_write_b32; the plain order is 8-way: every 4th row is 16 banks on), while the fragment read stays conflict free:
    // slot = 5 * (PS * (lr % 4) + lr / 4) is distinct over each of ds_read_b128's 16-lane groups exactly when PS = 4 mod 16.
    constexpr int PSA = BM / 4 + 4, PSB = BN / 4 + 4;
    constexpr int PA = (A_KC ? BM : 4 * PSA) * SKH / 2, PB = (B_KC ? BN : 4 * PSB) * SKH / 2;      // one bf16 plane (dwords)
    constexpr int ASZ = HB ? NP * PA : (A_KC ? BM * SKR : BK * SA);             // (floats)
    constexpr int BSZ = HB ? NP * PB : (B_KC ? BN * SKR : BK * SB);
    constexpr int CTS = BN + 4;                // row stride of the staged output tile (epilogue)
    constexpr int SMEM = (HB && BM * CTS > NBUF * (ASZ + BSZ)) ? BM * CTS : NBUF * (ASZ + BSZ);
    static_assert(!HB || (VEC && BK % 16 == 0 && (BM * KF / NT) % 2 == 0 && (BN * KF / NT) % 2 == 0), "bf16 form: vector loads, paired k rows");
    constexpr int NA = BM * KF / NT;
    constexpr int NB = BN * KF / NT;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    static_assert(NT == 256, "4-wave blocks only");
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    float* const As = smem;
    float* const Bs = smem + NBUF * ASZ;

    const vbg_conv_geo geo = p.geo;            // uniform descriptor fields live in SGPRs for the whole kernel
    const int tid = threadIdx.x;
    // ---- XCD-aware block -> tile map ------------------------------------------------------------------------------
    // The dispatcher places workgroup b on XCD b % 8 (observed, speed only) and each XCD has its own 4 MB L2.  With the
    // plain (x, y) mapping every L2 sees every row panel AND every column panel of the problem (measured with
    // rocprofv3 FETCH_SIZE: 661 MB of L2 misses for the 22 MB of operands of a 4128x3072x768 GEMM).  Remap: XCD k owns the
    // k-th contiguous eighth of the tile sequence, and the sequence walks the tile grid in bands of XCD_GROUP row tiles
    // (row tile fastest inside a band), so the ~128 blocks resident on one XCD share ~8 row panels and ~16 column panels.
    constexpr unsigned XCDS = 8, XCD_GROUP = 8;
    const unsigned gx = gridDim.x, gy = gridDim.y;
    const unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned total = gx * gy * gridDim.z;
    const unsigned xcd = lin % XCDS, local = lin / XCDS;
    const unsigned per_xcd = (total + XCDS - 1) / XCDS, tall = (total % XCDS) ? (total % XCDS) : XCDS;
    // (grouped launches -- the attention products, many tiny independent problems -- keep the plain order: measured slower
    // with the remap, 38 -> 71 us per call)
    const unsigned pid = p.grp ? lin : (xcd < tall ? xcd * per_xcd + local : tall * per_xcd + (xcd - tall) * (per_xcd - 1) + local);
    const unsigned slice = gx * gy;
    const int z = (int)(pid / slice);
    const unsigned rem = pid - (unsigned)z * slice;
    const unsigned band = XCD_GROUP * gy, bid = rem / band, first = bid * XCD_GROUP;
    const unsigned bm = min(gx - first, XCD_GROUP), inb = rem - bid * band;
    const unsigned tile_m = p.grp ? rem % gx : first + inb % bm, tile_n = p.grp ? rem / gx : inb / bm;
    const int grp = z / p.splitk, split = z - grp * p.splitk;
    int M = p.M, N = p.N, K = p.K;
    const float* A = p.A;
    const float* B = p.B;
    float* C = p.C + (long long)split * p.slab_stride;          // (slab_stride > 0: every split stores its own partial product)
    float* C2 = p.C2;
    const float* bias = p.bias;
    if (p.grp) {
        const long long* g = p.grp + 8 * (long long)grp;
        M = (int)g[0]; N = (int)g[1]; K = (int)g[2];
        A += g[3]; B += g[4]; C += g[5];
        if (C2) C2 += g[5];
        if (bias) bias += g[6];
    }
    const int m0 = (int)tile_m * BM, n0 = (int)tile_n * BN;
    if (m0 >= M || n0 >= N) return;
    const int nkt = (K + BK - 1) / BK;
    const int per = (nkt + p.splitk - 1) / p.splitk;
    const int kt0 = split * per;
    const int kt1 = min(nkt, kt0 + per);
    if (kt0 >= kt1) return;

    // ---------------- loader state -------------------------------------------------------
    // K-contiguous kinds : float4 #i (f = tid + NT i) covers tile row f / KF, k offset (f % KF) * 4
    // row-contiguous kinds: float4 #i covers tile rows (f % (BR/4)) * 4 .. +3, k index f / (BR/4)
    // Scalars (SGPR): k0 = first k of the NEXT tile to load, descriptor bases abase / bbase (pointing at that tile),
    // remaining bytes for the k-major kinds (their reduction tail falls off the descriptor and reads 0).
    int k0 = kt0 * BK;
    const float* abase = A;
    const float* bbase = B;
    long long a_rem = NREC_MAX, b_rem = NREC_MAX;
    unsigned avo[NA][NE], bvo[NB][NE];
    const int kcA = (tid % KF) * 4;            // k offset of this thread's float4s in a K-contiguous tile (NT % KF == 0)
    // row-contiguous kinds (BR4 float4 per k row of the tile): float4 #i covers tile rows r_row .. +3 at tile k index r_k(i).
    // fp32 form: pass i takes k rows [i*NT/BR4, (i+1)*NT/BR4); bf16 form: passes 2j, 2j+1 take ADJACENT k rows, so that a
    // thread can pack (k, k+1) pairs into dwords for the K-contiguous bf16 LDS tile.
    // (bf16 forms, 128-row tiles: the 32 lanes of a half wave take 16 row groups x 2 k pairs instead of 32 row groups of one
    // k pair -- their ds_write_b32 then hit every bank at most twice, see the LDS layout above)
    auto r_c = [&](int BR4) { return (HB && BR4 == 32) ? (tid & 15) + 16 * ((tid >> 5) & 1) : tid % BR4; };
    auto r_q = [&](int BR4) { return (HB && BR4 == 32) ? ((tid >> 4) & 1) + 2 * (tid >> 6) : tid / BR4; };
    auto r_row = [&](int BR4) { return r_c(BR4) * 4; };
    auto r_k = [&](int i, int BR4) { return HB ? 2 * r_q(BR4) + (i & 1) + 2 * (NT / BR4) * (i >> 1) : tid / BR4 + i * (NT / BR4); };

    // ---- A -------------------------------------------------------------------------------
    int a_n[NA], a_y[NA], a_x[NA];             // tile row -> (image, y, x) for conv / up-sampled segments
    bool a_rv[NA];
    if constexpr (A_KC) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int gm = m0 + (tid + i * NT) / KF;
            a_rv[i] = gm < M;
            a_n[i] = a_y[i] = a_x[i] = 0;
            if (AK == VBG_OP_CONV_K || p.a_H > 0) {
                const int Hr = (AK == VBG_OP_CONV_K) ? geo.Hr : p.a_H;
                const int Wr = (AK == VBG_OP_CONV_K) ? geo.Wr : p.a_W;
                const int g = a_rv[i] ? gm : M - 1;
                a_x[i] = g % Wr;
                const int t = g / Wr;
                a_y[i] = t % Hr;
                a_n[i] = t / Hr;
            }
        }
    }
    // DENSE_K A: K segments (early / late / P_fuse fusion read several tensors in place, some through a nearest
    // 2^shift up-sampling); the kernarg arrays are only read when the k-loop crosses into the next segment
    int seg = 0, seg_kend = K;
    auto enter_segment = [&](int sg) {
        seg = sg;
        const int kbeg = (sg == 0) ? 0 : p.a_seg_kend[sg - 1];
        seg_kend = (p.a_nseg == 1) ? K : p.a_seg_kend[sg];
        const long long ld = p.a_seg_ld[sg];
        const int sh = p.a_seg_shift[sg];
        const float* sp = p.a_seg_ptr[sg] + (A - p.A);                  // + group offset
        if (sh == 0) sp += (long long)m0 * ld;
        abase = sp + (k0 - kbeg);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            long long row = (tid + i * NT) / KF;
            if (sh > 0) row = ((long long)a_n[i] * (p.a_H >> sh) + (a_y[i] >> sh)) * (p.a_W >> sh) + (a_x[i] >> sh);
            set_vo<NE>(avo[i], row * ld + kcA, a_rv[i]);
        }
    };
    // CONV_K A: gathered NHWC source; offsets are relative to the image of the block's first output pixel and are
    // recomputed only when the k-loop moves to the next filter tap (every Cs / BK tiles)
    int a_tap = 0, a_c0 = 0, a_dy = 0, a_dx = 0;
    bool a_dirty = true;
    const float* a_img = A;
    auto conv_a_offsets = [&]() {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            int sy, sx;
            bool ok = a_rv[i];
            if (!geo.dgrad) {
                sy = a_y[i] * geo.stride - geo.pad + a_dy;
                sx = a_x[i] * geo.stride - geo.pad + a_dx;
            } else {
                const int ty = a_y[i] + geo.pad - a_dy, tx = a_x[i] + geo.pad - a_dx;
                sy = ty / geo.stride; sx = tx / geo.stride;
                ok = ok && ty >= 0 && tx >= 0 && (sy * geo.stride == ty) && (sx * geo.stride == tx);
            }
            ok = ok && sy >= 0 && sy < geo.Hs && sx >= 0 && sx < geo.Ws;
            set_vo<NE>(avo[i], (((long long)a_n[i] * geo.Hs + sy) * geo.Ws + sx) * geo.Cs + kcA, ok);
        }
    };
    if constexpr (AK == VBG_OP_DENSE_K) {
        enter_segment(0);
        while (seg + 1 < p.a_nseg && k0 >= seg_kend) enter_segment(seg + 1);
    } else if constexpr (AK == VBG_OP_CONV_K) {
        const int nb = m0 / (geo.Hr * geo.Wr);
        a_img = A + (long long)nb * geo.Hs * geo.Ws * geo.Cs;
#pragma unroll
        for (int i = 0; i < NA; ++i) a_n[i] -= nb;
        a_tap = k0 / geo.Cs;
        a_c0 = k0 - a_tap * geo.Cs;
        a_dy = a_tap / geo.kw;
        a_dx = a_tap - a_dy * geo.kw;
        abase = a_img + a_c0;
    } else {  // VBG_OP_DENSE_R : elem(row, k) = A[k*lda + row]
        abase = A + (long long)k0 * p.lda + m0;
        a_rem = ((long long)(K - k0) * p.lda) * 4;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int rl = r_row(BM / 4);
            set_vo<NE>(avo[i], (long long)r_k(i, BM / 4) * p.lda + rl, m0 + rl < M, M - m0 - rl);
        }
    }
    // ---- B -------------------------------------------------------------------------------
    int b_tap = 0, b_co0 = 0;                  // WT_R: k = tap * Cout + co
    int b_dy[NB], b_dx[NB], b_ci[NB];          // CONV_R: columns (tap, ci) are fixed per thread
    bool b_cv[NB];
    int b_pn[NB], b_py[NB], b_px[NB];          // CONV_R: the reduction index is the output pixel; each thread walks its pixel
    const int pix_dq = (BKD == VBG_OP_CONV_R) ? BK / max(geo.Wr, 1) : 0;       // forward by BK per k-tile
    const int pix_dr = (BKD == VBG_OP_CONV_R) ? BK - pix_dq * geo.Wr : 0;
    int b_img_n = 0, b_img_pix = 0;            // CONV_R: image / in-image pixel of the tile's first reduction pixel (scalar walk)
    const int b_hw = (BKD == VBG_OP_CONV_R) ? geo.Hr * geo.Wr : 1;
    // CONV_R fast path: when a k-tile is a piece of one pixel row (Wr % BK == 0) or a whole number of rows of one image
    // (BK % Wr == 0, Hr*Wr % BK == 0), a thread's pixel keeps its position RELATIVE to the tile's first pixel, whose
    // (y0, xs) walk in SGPRs: source row = y0*stride + b_cy, source column = xs*stride + b_cx with per-thread constants,
    // the address splits into a scalar part (descriptor base) + the constant b_co, and only the two range checks stay VALU.
    const bool b_fast = (BKD == VBG_OP_CONV_R) && ((geo.Wr % BK == 0) || (BK % geo.Wr == 0 && b_hw % BK == 0));
    int b_cy[NB], b_cx[NB];
    unsigned b_co[NB];
    int b_y0 = 0, b_xs = 0;
    const float* b_imgp = B;
    if constexpr (BKD == VBG_OP_DENSE_K) {      // elem(col, k) = B[col*ldb + k]
        bbase = B + (long long)n0 * p.ldb + k0;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int rl = (tid + i * NT) / KF;
            set_vo<NE>(bvo[i], (long long)rl * p.ldb + kcA, n0 + rl < N);
        }
    } else if constexpr (BKD == VBG_OP_DENSE_R) {   // elem(col, k) = B[k*ldb + col]
        bbase = B + (long long)k0 * p.ldb + n0;
        b_rem = ((long long)(K - k0) * p.ldb) * 4;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int cl = r_row(BN / 4);
            set_vo<NE>(bvo[i], (long long)r_k(i, BN / 4) * p.ldb + cl, n0 + cl < N, N - n0 - cl);
        }
    } else if constexpr (BKD == VBG_OP_WT_R) {      // dgrad weights [Cout][taps][Cin]: k = tap*Cout + co, col = ci
        const int Cout = geo.Cs, taps = geo.kh * geo.kw;
        b_tap = k0 / Cout;
        b_co0 = k0 - b_tap * Cout;
        bbase = B + ((long long)b_co0 * taps + b_tap) * N + n0;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int cl = r_row(BN / 4);
            set_vo<NE>(bvo[i], (long long)r_k(i, BN / 4) * taps * N + cl, n0 + cl < N);
        }
    } else {                                        // VBG_OP_CONV_R: k = pixel, col = (tap, ci)
        b_img_n = k0 / b_hw;
        b_img_pix = k0 - b_img_n * b_hw;
        bbase = B + (long long)b_img_n * geo.Hs * geo.Ws * geo.Cs;
        b_imgp = bbase;
        b_y0 = b_img_pix / geo.Wr;
        b_xs = b_img_pix - b_y0 * geo.Wr;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int c = n0 + r_row(BN / 4);
            b_cv[i] = c < N;
            const int cc = min(c, N - 4);
            const int tap = cc / geo.Cs;
            b_ci[i] = cc - tap * geo.Cs;
            b_dy[i] = tap / geo.kw;
            b_dx[i] = tap - b_dy[i] * geo.kw;
            const int pix = k0 + r_k(i, BN / 4);
            b_px[i] = pix % geo.Wr;
            const int t = pix / geo.Wr;
            b_py[i] = t % geo.Hr;
            b_pn[i] = t / geo.Hr - b_img_n;        // relative to the descriptor's image
            // fast path constants (offsets are biased by +pad rows / columns so they are never negative)
            const int kl = r_k(i, BN / 4);
            const int ry = kl / geo.Wr, rx = kl - ry * geo.Wr;
            b_cy[i] = ry * geo.stride - geo.pad + b_dy[i];
            b_cx[i] = rx * geo.stride - geo.pad + b_dx[i];
            b_co[i] = b_cv[i] ? (unsigned)((((long long)(b_cy[i] + geo.pad) * geo.Ws + (b_cx[i] + geo.pad)) * geo.Cs + b_ci[i]) * 4) : VO_INVALID;
        }
    }

    float4 ra[NA], rb[NB];
    int st_rem_a = BK, st_rem_b = BK;      // valid reduction length of the tile held in ra / rb (only < BK on a K-contiguous tail)
    // Only the K-contiguous dense kinds need per-thread work on the reduction tail (K % BK != 0); that code is compiled
    // into a separate copy of the k-tile body (TAIL = true) which runs for the single tile that needs it.
    constexpr bool HAS_TAIL = (AK == VBG_OP_DENSE_K || BKD == VBG_OP_DENSE_K);

    // loads the tile at k0 into ra / rb and advances every piece of scalar state to the next tile
    auto load_tiles_to = [&](auto tail_tag, auto& RA, auto& RB, int& REMA, int& REMB, unsigned inv) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        // ------------------------------ A ------------------------------
        if constexpr (AK == VBG_OP_DENSE_K) {
            if (seg + 1 < p.a_nseg && k0 >= seg_kend) enter_segment(seg + 1);     // rare, uniform
            const int rem = seg_kend - k0;
            REMA = rem;
            const __amdgpu_buffer_rsrc_t r = make_rsrc(abase, NREC_MAX);
            if constexpr (!TAIL) {
#pragma unroll
                for (int i = 0; i < NA; ++i) RA[i] = bload<NE>(r, avo[i], inv);
            } else {                      // reduction tail: chunks (elements) at or beyond K must not be touched
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    unsigned vo[NE];
#pragma unroll
                    for (int e = 0; e < NE; ++e) vo[e] = (kcA + e < rem) ? avo[i][e] : VO_INVALID;
                    RA[i] = bload<NE>(r, vo, inv);
                }
            }
            abase += BK;
        } else if constexpr (AK == VBG_OP_CONV_K) {
            if (a_dirty) { conv_a_offsets(); a_dirty = false; }
            const __amdgpu_buffer_rsrc_t r = make_rsrc(abase, NREC_MAX);
#pragma unroll
            for (int i = 0; i < NA; ++i) RA[i] = bload<NE>(r, avo[i], inv);
            a_c0 += BK;
            abase += BK;
            if (a_c0 >= geo.Cs) {
                a_c0 = 0; abase = a_img; a_dirty = true;
                if (++a_dx == geo.kw) { a_dx = 0; ++a_dy; }
            }
        } else {
            const __amdgpu_buffer_rsrc_t r = make_rsrc(abase, a_rem);
#pragma unroll
            for (int i = 0; i < NA; ++i) RA[i] = bload<NE>(r, avo[i], inv);
            abase += (long long)BK * p.lda;
            a_rem -= (long long)BK * p.lda * 4;
        }
        // ------------------------------ B ------------------------------
        if constexpr (BKD == VBG_OP_DENSE_K) {
            const int rem = K - k0;
            REMB = rem;
            const __amdgpu_buffer_rsrc_t r = make_rsrc(bbase, NREC_MAX);
            if constexpr (!TAIL) {
#pragma unroll
                for (int i = 0; i < NB; ++i) RB[i] = bload<NE>(r, bvo[i], inv);
            } else {
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    unsigned vo[NE];
#pragma unroll
                    for (int e = 0; e < NE; ++e) vo[e] = (kcA + e < rem) ? bvo[i][e] : VO_INVALID;
                    RB[i] = bload<NE>(r, vo, inv);
                }
            }
            bbase += BK;
        } else if constexpr (BKD == VBG_OP_DENSE_R) {
            const __amdgpu_buffer_rsrc_t r = make_rsrc(bbase, b_rem);
#pragma unroll
            for (int i = 0; i < NB; ++i) RB[i] = bload<NE>(r, bvo[i], inv);
            bbase += (long long)BK * p.ldb;
            b_rem -= (long long)BK * p.ldb * 4;
        } else if constexpr (BKD == VBG_OP_WT_R) {
            const __amdgpu_buffer_rsrc_t r = make_rsrc(bbase, NREC_MAX);
#pragma unroll
            for (int i = 0; i < NB; ++i) RB[i] = bload<NE>(r, bvo[i], inv);
            const int taps = geo.kh * geo.kw;
            b_co0 += BK;
            bbase += (long long)BK * taps * N;
            if (b_co0 >= geo.Cs) {
                b_co0 = 0; ++b_tap;
                bbase = B + (long long)b_tap * N + n0;
            }
        } else if (b_fast) {
            const int ys = b_y0 * geo.stride, xs = b_xs * geo.stride;
            const __amdgpu_buffer_rsrc_t r =
                make_rsrc(b_imgp + ((long long)(ys - geo.pad) * geo.Ws + (xs - geo.pad)) * geo.Cs, NREC_MAX);
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const bool ok = (unsigned)(ys + b_cy[i]) < (unsigned)geo.Hs && (unsigned)(xs + b_cx[i]) < (unsigned)geo.Ws;
                unsigned vo[1] = {ok ? b_co[i] : VO_INVALID};
                RB[i] = bload<1>(r, vo, inv);
            }
            b_xs += BK;
            while (b_xs >= geo.Wr) { b_xs -= geo.Wr; ++b_y0; }
            if (b_y0 >= geo.Hr) { b_y0 -= geo.Hr; b_imgp += (long long)geo.Hs * geo.Ws * geo.Cs; }
        } else {
            const __amdgpu_buffer_rsrc_t r = make_rsrc(bbase, NREC_MAX);
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int pix = k0 + r_k(i, BN / 4);
                const int sy = b_py[i] * geo.stride - geo.pad + b_dy[i];
                const int sx = b_px[i] * geo.stride - geo.pad + b_dx[i];
                const bool ok = (pix < K) && b_cv[i] && sy >= 0 && sy < geo.Hs && sx >= 0 && sx < geo.Ws;
                unsigned vo[NE];
                set_vo<NE>(vo, (((long long)b_pn[i] * geo.Hs + sy) * geo.Ws + sx) * geo.Cs + b_ci[i], ok);
                RB[i] = bload<NE>(r, vo, inv);
                // advance this thread's pixel by BK for the next k-tile
                b_px[i] += pix_dr;
                const int cx = b_px[i] >= geo.Wr;
                b_px[i] -= cx ? geo.Wr : 0;
                b_py[i] += pix_dq + cx;
                while (b_py[i] >= geo.Hr) { b_py[i] -= geo.Hr; ++b_pn[i]; }
            }
            // the descriptor follows the image of the next tile's first pixel (scalar walk)
            b_img_pix += BK;
            while (b_img_pix >= b_hw) {
                b_img_pix -= b_hw;
                bbase += (long long)geo.Hs * geo.Ws * geo.Cs;
#pragma unroll
                for (int i = 0; i < NB; ++i) --b_pn[i];
            }
        }
        k0 += BK;
    };
    auto load_tiles = [&](auto tail_tag) { load_tiles_to(tail_tag, ra, rb, st_rem_a, st_rem_b, 0u); };

    const int a_prologue = p.a_prologue;
    const float a_scale = p.a_scale;
    // bf16 form of one operand tile: K-contiguous float4s become 4 bf16 (one 8-byte ds_write); row-contiguous float4s of two
    // adjacent k rows are packed pairwise (k, k+1) into one dword per tile row
    // exact three-way split of a PAIR of floats into packed bf16 pairs (a in the low half): hi = top 16 bits, mid = top 16 bits of
    // the (exact) remainder, lo = the (exact, <= 8 significant bits) remainder of that
    auto split3 = [&](float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
        const float ra = a - __uint_as_float(__float_as_uint(a) & 0xffff0000u), rb = b - __uint_as_float(__float_as_uint(b) & 0xffff0000u);
        const float sa = ra - __uint_as_float(__float_as_uint(ra) & 0xffff0000u), sb = rb - __uint_as_float(__float_as_uint(rb) & 0xffff0000u);
        hi = __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
        mid = __builtin_amdgcn_perm(__float_as_uint(rb), __float_as_uint(ra), 0x07060302u);
        lo = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
    };
    auto split2 = [&](float a, float b, unsigned& hi, unsigned& lo) {
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
        const f32x2_t v = {a, b};
        const f16x2_t h = __builtin_convertvector(v, f16x2_t);                        // v_cvt_pk_f16_f32: round to nearest even
        const f16x2_t l = __builtin_convertvector((v - __builtin_convertvector(h, f32x2_t)) * 2048.f, f16x2_t);
        hi = __builtin_bit_cast(unsigned, h);
        lo = __builtin_bit_cast(unsigned, l);
    };
    auto store_half = [&](unsigned* dst, const auto& r, auto kc_tag, auto br4_tag, auto plane_tag) {
        constexpr int n = std::extent<std::remove_reference_t<decltype(r)>>::value;
        constexpr int BR4 = decltype(br4_tag)::value;
        constexpr int PS = BR4 + 4;
        constexpr int PL = decltype(plane_tag)::value;          // plane stride (dwords)
        if constexpr (decltype(kc_tag)::value) {
#pragma unroll
            for (int i = 0; i < n; ++i) {
                const int f = tid + i * NT;
                const int o = ((f / KF) * SKH + (f % KF) * 4) / 2;
                if constexpr (PREC == 3) {
                    uint2 h, m, l;
                    split3(r[i].x, r[i].y, h.x, m.x, l.x);
                    split3(r[i].z, r[i].w, h.y, m.y, l.y);
                    *reinterpret_cast<uint2*>(&dst[o]) = h;
                    *reinterpret_cast<uint2*>(&dst[o + PL]) = m;
                    *reinterpret_cast<uint2*>(&dst[o + 2 * PL]) = l;
                } else if constexpr (PREC == 2) {
                    uint2 h, l;
                    split2(r[i].x, r[i].y, h.x, l.x);
                    split2(r[i].z, r[i].w, h.y, l.y);
                    *reinterpret_cast<uint2*>(&dst[o]) = h;
                    *reinterpret_cast<uint2*>(&dst[o + PL]) = l;
                } else {
                    uint2 w;
                    w.x = cvt_pk_bf16(r[i].x, r[i].y); w.y = cvt_pk_bf16(r[i].z, r[i].w);
                    *reinterpret_cast<uint2*>(&dst[o]) = w;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < n; i += 2) {
                const int row4 = r_c(BR4), k = r_k(i, BR4);
                const float e0[4] = {r[i].x, r[i].y, r[i].z, r[i].w}, e1[4] = {r[i + 1].x, r[i + 1].y, r[i + 1].z, r[i + 1].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int o = ((e * PS + row4) * SKH + k) / 2;
                    if constexpr (PREC == 3) {
                        unsigned h, m, l;
                        split3(e0[e], e1[e], h, m, l);
                        dst[o] = h; dst[o + PL] = m; dst[o + 2 * PL] = l;
                    } else if constexpr (PREC == 2) {
                        unsigned h, l;
                        split2(e0[e], e1[e], h, l);
                        dst[o] = h; dst[o + PL] = l;
                    } else {
                        dst[o] = cvt_pk_bf16(e0[e], e1[e]);
                    }
                }
            }
        }
    };
    auto store_tiles = [&](auto tail_tag, int buf) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        float* as = As + buf * ASZ;
        float* bs = Bs + buf * BSZ;
        if constexpr (HB) {
            float4 va[NA], vb[NB];
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                float4 v = ra[i];
                if constexpr (AK == VBG_OP_DENSE_K && TAIL) v = mask4(v, st_rem_a - kcA);
                if (a_prologue == 1) {
                    v.x = fmaxf(v.x, 0.f) * a_scale; v.y = fmaxf(v.y, 0.f) * a_scale;
                    v.z = fmaxf(v.z, 0.f) * a_scale; v.w = fmaxf(v.w, 0.f) * a_scale;
                }
                va[i] = v;
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                float4 v = rb[i];
                if constexpr (BKD == VBG_OP_DENSE_K && TAIL) v = mask4(v, st_rem_b - kcA);
                vb[i] = v;
            }
            store_half(reinterpret_cast<unsigned*>(as), va, std::integral_constant<bool, A_KC>{}, std::integral_constant<int, BM / 4>{}, std::integral_constant<int, PA>{});
            store_half(reinterpret_cast<unsigned*>(bs), vb, std::integral_constant<bool, B_KC>{}, std::integral_constant<int, BN / 4>{}, std::integral_constant<int, PB>{});
            return;
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            float4 v = ra[i];
            if constexpr (AK == VBG_OP_DENSE_K && VEC && TAIL) v = mask4(v, st_rem_a - kcA);
            if (a_prologue == 1) {
                v.x = fmaxf(v.x, 0.f) * a_scale; v.y = fmaxf(v.y, 0.f) * a_scale;
                v.z = fmaxf(v.z, 0.f) * a_scale; v.w = fmaxf(v.w, 0.f) * a_scale;
            }
            const int f = tid + i * NT;
            if constexpr (A_KC) *reinterpret_cast<float4*>(&as[(f / KF) * SKR + (f % KF) * 4]) = v;
            else *reinterpret_cast<float4*>(&as[(f / (BM / 4)) * SA + (f % (BM / 4)) * 4]) = v;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            float4 v = rb[i];
            if constexpr (BKD == VBG_OP_DENSE_K && VEC && TAIL) v = mask4(v, st_rem_b - kcA);
            const int f = tid + i * NT;
            if constexpr (B_KC) *reinterpret_cast<float4*>(&bs[(f / KF) * SKR + (f % KF) * 4]) = v;
            else *reinterpret_cast<float4*>(&bs[(f / (BN / 4)) * SB + (f % (BN / 4)) * 4]) = v;
        }
    };

    // ---------------- main loop ---------------------------------------------------------
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lk = lane >> 5;
    f32x16 acc[TM][TN], acx[PREC == 2 ? TM : 1][PREC == 2 ? TN : 1];          // (PREC 2: the cross products, scaled by 2^11)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][j][r] = 0.f;
                if constexpr (PREC == 2) acx[i][j][r] = 0.f;
            }

    // MFMA step j of k-group g uses k = 8g + 4*lk + j (same permutation for A and B)
    const int a_off = A_KC ? (wm * WM + lr) * SKR + 4 * lk : 4 * lk * SA + wm * WM + lr;
    const int b_off = B_KC ? (wn * WN + lr) * SKR + 4 * lk : 4 * lk * SB + wn * WN + lr;
    auto read_frag = [&](const float* as, const float* bs, int g, float (&fa)[TM][4], float (&fb)[TN][4]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if constexpr (A_KC) {
                const float4 v = *reinterpret_cast<const float4*>(as + i * 32 * SKR + 8 * g);
                fa[i][0] = v.x; fa[i][1] = v.y; fa[i][2] = v.z; fa[i][3] = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) fa[i][j] = as[(8 * g + j) * SA + i * 32];
            }
        }
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            if constexpr (B_KC) {
                const float4 v = *reinterpret_cast<const float4*>(bs + i * 32 * SKR + 8 * g);
                fb[i][0] = v.x; fb[i][1] = v.y; fb[i][2] = v.z; fb[i][3] = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[i][j] = bs[(8 * g + j) * SB + i * 32];
            }
        }
    };
    auto mma_group = [&](const float (&fa)[TM][4], const float (&fb)[TN][4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int n = 0; n < TN; ++n)
                    acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][j], fb[n][j], acc[i][n], 0, 0, 0);
    };

    using yes_t = std::integral_constant<bool, true>;
    using no_t = std::integral_constant<bool, false>;
    const int ntiles = kt1 - kt0;
    const bool tail_in_range = HAS_TAIL && kt1 == nkt && (K % BK) != 0;      // the LAST tile of this block is a reduction tail
    if constexpr (HB && (BK == 32 || PIPE2_X16)) {
        // ---- bf16 forms: two tiles in flight ---------------------------------------------------------------------------
        // Their MFMA phase is 4-16x shorter than the fp32 form's and no longer covers the latency of loads issued in the same
        // iteration (measured: ~2700 cycles per 128x128x16 k-tile against 768 cycles of MFMAs).  Two register sets: tile j
        // lives in set j & 1.  Iteration `it`: fragments of tile it from LDS  ->  loads of tile it+2 issued into the set tile it
        // came from  ->  MFMAs of tile it, with the split / conversion of tile it+1 (loaded a whole iteration ago, other set) and
        // its LDS writes issued into the gaps of the last piece products  ->  barrier.
        // Every iteration is the same code (even / odd copy for the two set assignments): tiles beyond the block's range are
        // loaded with every lane's offset marked invalid (they read 0, no memory traffic) and stored as zeros, K-contiguous
        // operands always use the chunk-masked tail form of the load -- no run-time branch arms around the MFMA blocks (those
        // cost accumulator copies), no peeled iteration kinds.
        float4 ra2[NA], rb2[NB];
        int rem_a1 = BK, rem_b1 = BK, rem_a2 = BK, rem_b2 = BK;
        using load_t = std::integral_constant<bool, HAS_TAIL>;
        auto fix = [&](auto& RA, auto& RB, int rema, int remb) {      // in place, before the split: partial chunks of a tail, prologue
            if (HAS_TAIL && (rema < BK || remb < BK)) {              // (uniform)
                if constexpr (AK == VBG_OP_DENSE_K) {
#pragma unroll
                    for (int i = 0; i < NA; ++i) RA[i] = mask4(RA[i], rema - kcA);
                }
                if constexpr (BKD == VBG_OP_DENSE_K) {
#pragma unroll
                    for (int i = 0; i < NB; ++i) RB[i] = mask4(RB[i], remb - kcA);
                }
            }
            if (a_prologue == 1) {
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    RA[i].x = fmaxf(RA[i].x, 0.f) * a_scale; RA[i].y = fmaxf(RA[i].y, 0.f) * a_scale;
                    RA[i].z = fmaxf(RA[i].z, 0.f) * a_scale; RA[i].w = fmaxf(RA[i].w, 0.f) * a_scale;
                }
            }
        };
        auto store_set = [&](auto& RA, auto& RB, int buf) {
            store_half(reinterpret_cast<unsigned*>(As + buf * ASZ), RA, std::integral_constant<bool, A_KC>{}, std::integral_constant<int, BM / 4>{}, std::integral_constant<int, PA>{});
            store_half(reinterpret_cast<unsigned*>(Bs + buf * BSZ), RB, std::integral_constant<bool, B_KC>{}, std::integral_constant<int, BN / 4>{}, std::integral_constant<int, PB>{});
        };
        constexpr int KS = BK / 16;
        constexpr int NPP = PREC == 3 ? 6 : (PREC == 2 ? 3 : 1);
        const int arow = A_KC ? wm * WM + lr : (lr & 3) * PSA + (wm * WM + lr) / 4;
        const int brow = B_KC ? wn * WN + lr : (lr & 3) * PSB + (wn * WN + lr) / 4;
        constexpr int AI = A_KC ? 32 : 8, BI = B_KC ? 32 : 8;              // LDS rows between a wave's 32-row fragments
        u32x4 fa[KS][NP][TM], fb[KS][NP][TN];
        // lane (lr, lk) of MFMA step s reads the 8 bf16 k = 16 s + 8 lk .. +7 of its row: one ds_read_b128 per fragment and plane
        auto read_frags = [&](int buf) {
            const u32x4* as = reinterpret_cast<const u32x4*>(As + buf * ASZ) + (arow * SKH + 8 * lk) / 8;
            const u32x4* bs = reinterpret_cast<const u32x4*>(Bs + buf * BSZ) + (brow * SKH + 8 * lk) / 8;
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int q = 0; q < NP; ++q) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) fa[s][q][i] = as[(q * 2 * PA + i * AI * SKH + 16 * s) / 8];
#pragma unroll
                    for (int i = 0; i < TN; ++i) fb[s][q][i] = bs[(q * 2 * PB + i * BI * SKH + 16 * s) / 8];
                }
        };
        // piece products, smallest first: (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi)
        auto mma_range = [&](auto t0_tag, auto t1_tag) {
            constexpr int qa[6] = {PREC == 3 ? 2 : 0, 0, 1, 1, 0, 0}, qb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int t = decltype(t0_tag)::value; t < decltype(t1_tag)::value; ++t)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int n = 0; n < TN; ++n)
                            if constexpr (PREC == 2) {          // (lo', hi) (hi, lo') -> cross sums, (hi, hi) -> main
                                f32x16& d_ = t < 2 ? acx[i][n] : acc[i][n];
                                d_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[s][t == 0 ? 1 : 0][i]),
                                                                            __builtin_bit_cast(f16x8, fb[s][t == 1 ? 1 : 0][n]), d_, 0, 0, 0);
                            } else
                            acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[s][qa[t]][i]),
                                                                                __builtin_bit_cast(bf16x8, fb[s][qb[t]][n]), acc[i][n], 0, 0, 0);
        };
        using i0 = std::integral_constant<int, 0>;
        using ih = std::integral_constant<int, NPP / 3>;
        using i1 = std::integral_constant<int, NPP>;
        // one iteration; P = it & 1 (static): tile it and it+2 use set P (P = 0: ra / rb), tile it+1 the other set
        auto body = [&](auto par_tag, int it) {
            constexpr int P = decltype(par_tag)::value;
            auto& LA = P ? ra2 : ra;
            auto& LB = P ? rb2 : rb;
            auto& SA = P ? ra : ra2;
            auto& SB = P ? rb : rb2;
            int& lrem_a = P ? rem_a2 : rem_a1;
            int& lrem_b = P ? rem_b2 : rem_b1;
            const int srem_a = P ? rem_a1 : rem_a2, srem_b = P ? rem_b1 : rem_b2;
            read_frags(P);
            __builtin_amdgcn_sched_barrier(0);
            load_tiles_to(load_t{}, LA, LB, lrem_a, lrem_b, it + 2 < ntiles ? 0u : VO_INVALID);
            __builtin_amdgcn_sched_barrier(0);
            fix(SA, SB, srem_a, srem_b);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PREC == 3 || PREC == 2) {
                mma_range(i0{}, ih{});
                __builtin_amdgcn_sched_barrier(0);
                mma_range(ih{}, i1{});
                store_set(SA, SB, P ^ 1);
                constexpr int NM2 = KS * (NPP - NPP / 3) * TM * TN;
                constexpr int NV = (NA + NB) * 22;
                constexpr int ND = NP * ((A_KC ? NA : 2 * NA) + (B_KC ? NB : 2 * NB));
                sched_pipe<0, NM2, NV, ND>::run();
            } else {
                mma_range(i0{}, i1{});
                __builtin_amdgcn_sched_barrier(0);
                store_set(SA, SB, P ^ 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        };
        load_tiles_to(load_t{}, ra, rb, rem_a1, rem_b1, 0u);
        load_tiles_to(load_t{}, ra2, rb2, rem_a2, rem_b2, ntiles > 1 ? 0u : VO_INVALID);
        fix(ra, rb, rem_a1, rem_b1);
        store_set(ra, rb, 0);
        __syncthreads();
        int it = 0;
        for (; it + 1 < ntiles; it += 2) {
            body(std::integral_constant<int, 0>{}, it);
            body(std::integral_constant<int, 1>{}, it + 1);
        }
        if (it < ntiles) body(std::integral_constant<int, 0>{}, it);
    } else {
        if (tail_in_range && ntiles == 1) { load_tiles(yes_t{}); store_tiles(yes_t{}, 0); }
        else { load_tiles(no_t{}); store_tiles(no_t{}, 0); }
        __syncthreads();
        // One k-tile: compute tile `it` from LDS buffer it & 1 while (MORE) the next tile is fetched and stored into the other
        // buffer.  Order (pinned with sched_barrier; hipcc otherwise sinks the ds_reads below the dependent MFMA chains and
        // parks all non-MFMA work before/after the whole MFMA block):
        //   fragments g0, g1  ->  MFMA g0  ->  next tile's buffer loads  ->  fragments / MFMA g1..  ->
        //   ds_write of the prefetched tile BEFORE the last MFMA group  ->  last MFMA group  ->  barrier
        // The steady-state loop holds exactly one copy of the body (MORE, no tail); the tail-loading and the final tile are
        // peeled after it, so the loop carries no per-tile branches on them.
        auto k_tile = [&](auto tail_tag, auto more_tag, int buf) {
            constexpr bool MORE = decltype(more_tag)::value;
            if constexpr (HB) {
                // lane (lr, lk) of MFMA step s reads the 8 bf16 k = 16 s + 8 lk .. +7 of its row: one ds_read_b128 per fragment
                constexpr int KS = BK / 16;
                const int arow = A_KC ? wm * WM + lr : (lr & 3) * PSA + (wm * WM + lr) / 4;
                const int brow = B_KC ? wn * WN + lr : (lr & 3) * PSB + (wn * WN + lr) / 4;
                constexpr int AI = A_KC ? 32 : 8, BI = B_KC ? 32 : 8;          // LDS rows between a wave's 32-row fragments
                const u32x4* as = reinterpret_cast<const u32x4*>(As + buf * ASZ) + (arow * SKH + 8 * lk) / 8;
                const u32x4* bs = reinterpret_cast<const u32x4*>(Bs + buf * BSZ) + (brow * SKH + 8 * lk) / 8;
                u32x4 fa[KS][NP][TM], fb[KS][NP][TN];
#pragma unroll
                for (int s = 0; s < KS; ++s)
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
#pragma unroll
                        for (int i = 0; i < TM; ++i) fa[s][q][i] = as[(q * 2 * PA + i * AI * SKH + 16 * s) / 8];
#pragma unroll
                        for (int i = 0; i < TN; ++i) fb[s][q][i] = bs[(q * 2 * PB + i * BI * SKH + 16 * s) / 8];
                    }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (MORE) load_tiles(tail_tag);
                __builtin_amdgcn_sched_barrier(0);
                // piece products, smallest first: (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi)
                constexpr int NPP = PREC == 3 ? 6 : (PREC == 2 ? 3 : 1);
                constexpr int qa[6] = {PREC == 3 ? 2 : 0, 0, 1, 1, 0, 0}, qb[6] = {0, 2, 1, 0, 1, 0};
                auto mma_range = [&](auto t0_tag, auto t1_tag) {
#pragma unroll
                    for (int s = 0; s < KS; ++s)
#pragma unroll
                        for (int t = decltype(t0_tag)::value; t < decltype(t1_tag)::value; ++t)
#pragma unroll
                            for (int i = 0; i < TM; ++i)
#pragma unroll
                                for (int n = 0; n < TN; ++n)
                                    if constexpr (PREC == 2) {
                                        f32x16& d_ = t < 2 ? acx[i][n] : acc[i][n];
                                        d_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[s][t == 0 ? 1 : 0][i]),
                                                                                    __builtin_bit_cast(f16x8, fb[s][t == 1 ? 1 : 0][n]), d_, 0, 0, 0);
                                    } else
                                    acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[s][qa[t]][i]),
                                                                                        __builtin_bit_cast(bf16x8, fb[s][qb[t]][n]), acc[i][n], 0, 0, 0);
                };
                using i0 = std::integral_constant<int, 0>;
                using ih = std::integral_constant<int, NPP / 3>;
                using i1 = std::integral_constant<int, NPP>;
                if constexpr ((PREC == 3 || PREC == 2) && MORE) {
                    // first half of the piece products covers the latency of the loads just issued; the split of the loaded tile and its
                    // LDS writes are then issued into the gaps of the second half (a 32x32x16 bf16 MFMA holds the matrix pipe for 32
                    // cycles = ~8 issue slots)
                    mma_range(i0{}, ih{});
                    __builtin_amdgcn_sched_barrier(0);
                    mma_range(ih{}, i1{});
                    store_tiles(tail_tag, buf ^ 1);
                    constexpr int NM2 = KS * (NPP - NPP / 3) * TM * TN;
                    constexpr int NV = (NA + NB) * 22;
                    constexpr int ND = NP * ((A_KC ? NA : 2 * NA) + (B_KC ? NB : 2 * NB));
                    sched_pipe<0, NM2, NV, ND>::run();
                    __builtin_amdgcn_sched_barrier(0);
                    __syncthreads();
                } else {
                    mma_range(i0{}, i1{});
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (MORE) { store_tiles(tail_tag, buf ^ 1); __syncthreads(); }
                }
                return;
            }
            const float* as = As + buf * ASZ + a_off;
            const float* bs = Bs + buf * BSZ + b_off;
            float fa0[TM][4], fb0[TN][4], fa1[TM][4], fb1[TN][4];
            read_frag(as, bs, 0, fa0, fb0);
            read_frag(as, bs, 1, fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            mma_group(fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MORE) load_tiles(tail_tag);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NG == 2) {
                if constexpr (MORE) store_tiles(tail_tag, buf ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                mma_group(fa1, fb1);
            } else {
#pragma unroll
                for (int g = 2; g < NG; g += 2) {
                    read_frag(as, bs, g, fa0, fb0);
                    __builtin_amdgcn_sched_barrier(0);
                    mma_group(fa1, fb1);
                    __builtin_amdgcn_sched_barrier(0);
                    read_frag(as, bs, g + 1, fa1, fb1);
                    __builtin_amdgcn_sched_barrier(0);
                    mma_group(fa0, fb0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (MORE) store_tiles(tail_tag, buf ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                mma_group(fa1, fb1);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MORE) __syncthreads();
        };
        int it = 0;
        const int n_plain = ntiles - 1 - ((tail_in_range && ntiles >= 2) ? 1 : 0);
        for (; it < n_plain; ++it) k_tile(no_t{}, yes_t{}, it & 1);
        if constexpr (HAS_TAIL) {
            if (tail_in_range && ntiles >= 2) { k_tile(yes_t{}, yes_t{}, it & 1); ++it; }
        }
        k_tile(no_t{}, no_t{}, it & 1);
    }

    if constexpr (PREC == 2) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += acx[i][j][r] * (1.f / 2048.f);
    }

    // ---------------- epilogue ----------------------------------------------------------
    const bool add_bias = (bias != nullptr) && (split == 0);
    const int accumulate = p.accumulate, epi = p.epi;
    const bool atomic = accumulate && p.splitk > 1;
    const float alpha = p.alpha;
    const long long ldc = p.ldc;
    // Plain stores go through LDS: the accumulator layout gives each lane single floats of 16 different rows (16 dword stores per
    // 32x32 tile, 128 contiguous bytes per row); staged through the (now idle) operand tiles, every thread writes float4s of
    // complete 64-float row pieces instead.  Output-bound products (attention scores, K = 64) are limited by exactly this.
    constexpr bool STAGE_OK = BM * CTS <= SMEM;
    if (STAGE_OK && !atomic && (ldc & 3) == 0 && (((uintptr_t)C) & 15) == 0 && (epi != VBG_EPI_GELU_DUAL || (((uintptr_t)C2) & 15) == 0)) {
        __syncthreads();                                  // every wave is done with the operand tiles
        float* const Ct = smem;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Ct[(wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * CTS + wn * WN + j * 32 + lr] = acc[i][j][r] * alpha;
        __syncthreads();
        constexpr int QN = BN / 4;                        // float4 per tile row
        float4 cs = make_float4(0.f, 0.f, 0.f, 0.f), cq = cs;       // fused BatchNorm statistics: this thread's column partials
#pragma unroll
        for (int q = 0; q < BM * QN / NT; ++q) {
            const int idx = tid + q * NT;
            const int row = idx / QN, c = (idx % QN) * 4;
            const int gm = m0 + row, gn = n0 + c;
            if (gm >= M || gn >= N) continue;
            float4 v = *reinterpret_cast<const float4*>(&Ct[row * CTS + c]);
            if (add_bias) {
                v.x += bias[gn];
                if (gn + 1 < N) v.y += bias[gn + 1];
                if (gn + 2 < N) v.z += bias[gn + 2];
                if (gn + 3 < N) v.w += bias[gn + 3];
            }
            float* cp = C + (long long)gm * ldc + gn;
            if (p.stats) {                                 // (N % 4 == 0 is checked on the host when statistics are requested)
                cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
                cq.x += v.x * v.x; cq.y += v.y * v.y; cq.z += v.z * v.z; cq.w += v.w * v.w;
            }
            if (gn + 3 < N) {
                if (accumulate) {                          // single owner per element: plain read-modify-write
                    const float4 o = *reinterpret_cast<const float4*>(cp);
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                }
                if (epi == VBG_EPI_RELU) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                *reinterpret_cast<float4*>(cp) = v;
                if (epi == VBG_EPI_GELU_DUAL)
                    *reinterpret_cast<float4*>(C2 + (long long)gm * ldc + gn) = make_float4(gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w));
            } else {
                const float e[4] = {v.x, v.y, v.z, v.w};
                for (int t = 0; t < 4 && gn + t < N; ++t) {
                    const float x = (epi == VBG_EPI_RELU) ? fmaxf(e[t], 0.f) : e[t] + (accumulate ? cp[t] : 0.f);
                    cp[t] = x;
                    if (epi == VBG_EPI_GELU_DUAL) C2[(long long)gm * ldc + gn + t] = gelu_erf(x);
                }
            }
        }
        if (p.stats) {
            // threads with the same idx % QN own the same 4 columns: reduce their partials through LDS (behind the staged tile),
            // then one fp64 atomic per column and statistic into this row-tile's slot row
            constexpr int RG = NT / QN;                    // row groups
            float* red = smem;                             // [2][RG][BN], over the staged tile once every thread has read its pieces
            static_assert(2 * (NT / (BN / 4)) * BN <= SMEM, "no room for the statistics partials");
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
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WN + j * 32 + lr;
        if (n >= N) continue;
        const float bv = add_bias ? bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + wm * WM + i * 32 + 4 * lk;
            float* cp = C + (long long)mb * ldc + n;
            if (accumulate && !atomic) {
                // single owner per element: read-modify-write with ALL loads issued before the first store
                float old[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dm = (r & 3) + 8 * (r >> 2);
                    old[r] = (mb + dm < M) ? cp[(long long)dm * ldc] : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dm = (r & 3) + 8 * (r >> 2);
                    if (mb + dm < M) cp[(long long)dm * ldc] = old[r] + (acc[i][j][r] * alpha + bv);
                }
                continue;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dm = (r & 3) + 8 * (r >> 2);
                if (mb + dm >= M) continue;
                const float v = acc[i][j][r] * alpha + bv;
                const long long o = (long long)dm * ldc;
                if (atomic) {
                    unsafeAtomicAdd(cp + o, v);
                } else if (epi == VBG_EPI_RELU) {
                    cp[o] = fmaxf(v, 0.f);
                } else if (epi == VBG_EPI_GELU_DUAL) {
                    cp[o] = v;
                    C2[(long long)(mb + dm) * ldc + n] = gelu_erf(v);
                } else {
                    cp[o] = v;
                }
            }
        }
    }
}

// optional per-dispatch timing (vbg_gemm_timed): the start / stop events receive the dispatch packet's own begin / end
// timestamps -- what rocprofv3's kernel trace reads -- with no barrier packets around the kernel
struct launch_timer { hipEvent_t start = nullptr, stop = nullptr; };

template <int BM, int BN, int BK, int NT, int AK, int BKD, bool VEC, int PREC = 0>
static void launch_one(const vbg_gemm_desc& d, int groups, int maxM, int maxN, hipStream_t s, const launch_timer& t) {
    dim3 g(cdiv(maxM, BM), cdiv(maxN, BN), groups * d.splitk);
    (void)hipGetLastError();
    if (t.start && t.stop) hipExtLaunchKernelGGL((gemm_kernel<BM, BN, BK, NT, AK, BKD, VEC, PREC>), g, dim3(NT), 0, s, t.start, t.stop, 0, d);
    else hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, NT, AK, BKD, VEC, PREC>), g, dim3(NT), 0, s, d);
}

// tile code: BM*1000+BN (128128, 128064, 64064); 0 = heuristic.  (A barrier-free one-wave-per-tile variant (NT = 64) was
// measured 25-40 % slower than the 4-wave blocks on every shape and is not instantiated.)  bk: 16 / 32; 0 = heuristic.
static void pick_tile(const vbg_gemm_desc& d, int groups, int maxM, int maxN, int& tile, int& bk) {
    tile = d.tile;
    bk = d.bk;
    if (tile == 64) tile = 64064;
    if (tile == 128) tile = 128128;
    if (tile == 0) {
        // measured on MI355X (tools/gemm_bench.py): 64x64 wins almost everywhere (4 blocks = 16 waves per CU hide the LDS / barrier
        // phases); for very large M with N >= 256 (FPN merge / seg-head convs) 128x128 ties in isolation (130 TF/s both) and is
        // 0.7 % better for the training step as a whole (fewer LDS reads per MFMA, lower power), so it stays for those
        const long t128 = (long)cdiv(maxM, 128) * cdiv(maxN, 128) * groups * d.splitk;
        tile = (maxN >= 256 && t128 >= 1536 && d.a_kind != VBG_OP_DENSE_R) ? 128128 : 64064;
    }
    // short reductions are latency / output bound: the 20 KB BK=16 tiles double the resident blocks per CU (measured on the
    // attention score GEMMs, K = 64: 49 -> 63 TF/s)
    if (bk == 0) bk = (!d.grp && d.K <= 128) ? 16 : 32;
}

template <int AK, int BKD>
static int launch_pair(const vbg_gemm_desc& d, int groups, int maxM, int maxN, hipStream_t s, const launch_timer& t) {
    int tile, bk;
    pick_tile(d, groups, maxM, maxN, tile, bk);
    if (!(d.a_vec && d.b_vec)) {                         // unaligned operands: general scalar-load path
        launch_one<64, 64, 16, 256, AK, BKD, false>(d, groups, maxM, maxN, s, t);
    } else if (d.bf16 == 2 && d.bk != 16 && (AK == VBG_OP_DENSE_K || AK == VBG_OP_CONV_K) && BKD == VBG_OP_DENSE_K &&
               (d.tile == 0 ? !((long)cdiv(maxM, 128) * cdiv(maxN, 128) * groups * d.splitk >= 192 && maxN >= 128) : tile == 64064)) {
        // the fp16-pair form of the forward kinds (PREC 2), where the six-product form would run 64 x 64 tiles: 1024 x 1024 x 12544 163 -> 120 us,
        // the strided 3x3 convolutions 36 -> 29 / 41 -> 30 us.  On the 128 x 128 tiles it measured SLOWER than six products (131072 x 256 x 1024:
        // 391 -> 423 us; a second accumulator set on 304 registers, a VALU-heavier split): those keep PREC 3 (tools/gemm_f16_bench.py)
        if constexpr ((AK == VBG_OP_DENSE_K || AK == VBG_OP_CONV_K) && BKD == VBG_OP_DENSE_K)
            launch_one<64, 64, 32, 256, AK, BKD, true, 2>(d, groups, maxM, maxN, s, t);
    } else if ((d.bf16 == 3 || d.bf16 == 2) && d.bk != 16 &&
               (AK != VBG_OP_DENSE_R || d.tile != 0 || BKD == VBG_OP_DENSE_R ||
                (BKD == VBG_OP_CONV_R && d.K / d.splitk >= 2048 && maxM >= 128 && maxN >= 128))) {
        // fp32-grade split form (tools/gemm_bench.py; --fp32 for the other form).  Forward and dgrad kinds: 128x128x16 tiles (73 KB
        // of LDS, two blocks per CU) from ~200 tiles on -- 4128x3072x768 116 vs 163 us for the fp32 form, the 128x128-map 3x3
        // convs 760 vs 1160 us; at most one wave of tiles runs them 32 deep (123 KB, one block per CU; two tiles in flight) -- and
        // 64x64x32 (61 KB) below.  Weight gradients (row-contiguous A): the dense ones as 64x64x32 (157 vs 168 us on 3072x768x4128),
        // the convolution ones only where a split keeps >= 64 k-tiles (the 128x128-map and RoI convs: 128x128x32, 1078 vs 1297
        // and 465 vs 576 us) -- the short-reduction conv wgrads are 5-10 % slower than the fp32 form (their gathered B operand
        // already spends the VALU slots the split needs) and stay on it, as do products forced to 16-deep k-tiles.
        if (d.tile == 0) {
            const long t128 = (long)cdiv(maxM, 128) * cdiv(maxN, 128) * groups * d.splitk;
            if (AK == VBG_OP_DENSE_R) tile = (BKD == VBG_OP_CONV_R) ? 128132 : 64064;
            else tile = (t128 >= 192 && maxN >= 128) ? (t128 <= 256 ? 128132 : 128128) : 64064;
        }
        if (tile == 128132 || (tile == 128128 && d.bk == 32)) launch_one<128, 128, 32, 256, AK, BKD, true, 3>(d, groups, maxM, maxN, s, t);
        else if (tile == 128128) launch_one<128, 128, 16, 256, AK, BKD, true, 3>(d, groups, maxM, maxN, s, t);
        else launch_one<64, 64, 32, 256, AK, BKD, true, 3>(d, groups, maxM, maxN, s, t);
    } else if (d.bf16 == 1 && (d.bk == 0 || d.bk == 32)) {
        // amp: bf16 matrix cores (fp32 operands rounded on the way into LDS).  The loop is bound by operand traffic, not by the
        // MFMAs, so the larger tile wins as soon as it fills the chip.  (Products forced to 16-deep k-tiles -- channel counts
        // that are not a multiple of 32 -- and unaligned operands stay on the fp32 form.)
        // (tools/gemm_bench.py --amp, profiles/r01_gemm_shapes_amp.txt: 128x128 wins for the forward and dgrad kinds from ~512
        // tiles on -- 4128x3072x768 49 vs 59 us, the 128x128-map convs 262 vs 377 us -- and for the weight gradients with many
        // output tiles or a long reduction per split; everything smaller is faster with 64x64 blocks)
        if (d.tile == 0) {
            const long t128 = (long)cdiv(maxM, 128) * cdiv(maxN, 128) * groups;
            bool big;
            if (AK == VBG_OP_DENSE_R) big = maxM >= 128 && maxN >= 128 && (t128 >= 512 || d.K / d.splitk >= 2048);
            else big = t128 * d.splitk >= 512 && maxN >= 256;
            tile = big ? 128128 : 64064;
        }
        if (tile == 128128) launch_one<128, 128, 32, 256, AK, BKD, true, 1>(d, groups, maxM, maxN, s, t);
        else launch_one<64, 64, 32, 256, AK, BKD, true, 1>(d, groups, maxM, maxN, s, t);
    } else if (bk == 32) {
        if (tile == 128128) launch_one<128, 128, 32, 256, AK, BKD, true>(d, groups, maxM, maxN, s, t);
        else if (tile == 128064) launch_one<128, 64, 32, 256, AK, BKD, true>(d, groups, maxM, maxN, s, t);
        else launch_one<64, 64, 32, 256, AK, BKD, true>(d, groups, maxM, maxN, s, t);
    } else {
        if (tile == 128128) launch_one<128, 128, 16, 256, AK, BKD, true>(d, groups, maxM, maxN, s, t);
        else if (tile == 128064) launch_one<128, 64, 16, 256, AK, BKD, true>(d, groups, maxM, maxN, s, t);
        else launch_one<64, 64, 16, 256, AK, BKD, true>(d, groups, maxM, maxN, s, t);
    }
    VBG_LAUNCH_RET();
}

}  // namespace vbg

static int gemm_dispatch(const vbg_gemm_desc* desc, void* stream, const vbg::launch_timer& t) {
    using namespace vbg;
    VBG_CHECK_ARG(desc != nullptr);
    vbg_gemm_desc d = *desc;
    VBG_CHECK_ARG(d.A && d.B && d.C);
    VBG_CHECK_ARG(d.M >= 0 && d.N >= 0 && d.K >= 0);
    VBG_CHECK_ARG(d.splitk >= 0);
    VBG_CHECK_ARG(d.slab_stride >= 0);
    if (d.slab_stride > 0)          // deterministic split: plain partial products, combined by vbg_slab_reduce
        VBG_CHECK_ARG(d.splitk > 1 && !d.accumulate && d.epi == VBG_EPI_NONE && !d.bias && !d.C2 && !d.grp && !d.stats && d.slab_stride >= (long long)d.M * d.ldc &&
                      d.K >= 64ll * d.splitk * d.splitk);          // (every split owns k-tiles: a split without any would leave its slab unwritten)
    else if (d.splitk > 1) VBG_CHECK_ARG(d.accumulate == 1);
    if (d.epi == VBG_EPI_GELU_DUAL) VBG_CHECK_ARG(d.C2 != nullptr);
    if (d.accumulate) VBG_CHECK_ARG(d.epi == VBG_EPI_NONE);
    VBG_CHECK_ARG(d.a_nseg >= 0 && d.a_nseg <= 4);
    VBG_CHECK_ARG(d.bk == 0 || d.bk == 16 || d.bk == 32);
    if (d.stats) {          // fused output statistics ride on the LDS-staged store path of an unsplit, non-accumulating launch
        VBG_CHECK_ARG(d.stats_slots >= 1 && d.splitk == 1 && !d.accumulate && d.grp == nullptr && d.N % 4 == 0 && d.ldc % 4 == 0 &&
                      (uintptr_t)d.C % 16 == 0 && d.epi == VBG_EPI_NONE);
    }
    if (d.a_nseg == 0) {
        d.a_nseg = 1; d.a_seg_ptr[0] = d.A; d.a_seg_kend[0] = d.K; d.a_seg_ld[0] = d.lda; d.a_seg_shift[0] = 0;
    }
    if (d.a_nseg > 1) {
        VBG_CHECK_ARG(d.a_kind == VBG_OP_DENSE_K && d.grp == nullptr && d.a_seg_kend[d.a_nseg - 1] == d.K);
        for (int i = 0; i < d.a_nseg; ++i) {
            VBG_CHECK_ARG(d.a_seg_kend[i] % 16 == 0 && d.a_seg_ptr[i] != nullptr);
            if (d.a_seg_kend[i] % 32 != 0) d.bk = 16;          // a k-tile must stay inside one segment
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
        VBG_CHECK_ARG(d.grp == nullptr);
        if (d.geo.Cs % 32 != 0) d.bk = 16;                      // a k-tile must stay inside one filter tap
    }
    // Automatic split-K for forward / dgrad products whose output has too few tiles to fill the chip but a long reduction
    // (layer4 convolutions: 2048 x 512 outputs = 256 tiles, K = 4608: 73 -> 110 TF/s): the (small) output is zeroed here and the
    // splits accumulate with atomics.  Only where the epilogue is linear (no ReLU / GELU; the bias is added by split 0).
    // Opt-in (splitk == 0): the atomic accumulation order is not reproducible run to run, so inference paths keep splitk = 1.
    const bool auto_split = d.splitk == 0;
    if (d.splitk < 1) d.splitk = 1;
    if (auto_split && !d.accumulate && d.epi == VBG_EPI_NONE && !d.grp && d.a_nseg == 1) {
        const long tiles = (long)cdiv(d.M, 64) * cdiv(d.N, 64);
        const int nkt = cdiv(d.K, 32);
        // (<= 800 tiles also gains 6 % on the isolated 4128x768x3072 products, but not in the training step: the extra memset +
        // atomic traffic on the 12.7 MB outputs costs as much; reductions of 24 k-tiles lose outright)
        if (tiles <= 384 && nkt >= 48) {
            int sk = (int)((1024 + tiles - 1) / tiles);
            if (sk > nkt / 12) sk = nkt / 12;
            if (sk > 8) sk = 8;
            if (sk >= 2) {
                const hipError_t e = hipMemset2DAsync(d.C, (size_t)d.ldc * 4, 0, (size_t)d.N * 4, (size_t)d.M, s);
                if (e != hipSuccess) return (int)e;
                d.splitk = sk;
                d.accumulate = 1;
            }
        }
    }
    // the gathered side of a conv operand is float4-legal by construction; the DENSE side keeps the caller's flag
    if (d.a_kind == VBG_OP_CONV_K) { VBG_CHECK_ARG((uintptr_t)d.A % 16 == 0); d.a_vec = 1; }
    if (d.b_kind == VBG_OP_CONV_R) { VBG_CHECK_ARG((uintptr_t)d.B % 16 == 0 && d.N % 4 == 0); d.b_vec = 1; }
    if (d.b_kind == VBG_OP_WT_R) { VBG_CHECK_ARG((uintptr_t)d.B % 16 == 0 && d.N % 4 == 0 && d.geo.dgrad == 1); d.b_vec = 1; }
    if (d.a_kind == VBG_OP_DENSE_K && d.b_kind == VBG_OP_DENSE_K) return launch_pair<VBG_OP_DENSE_K, VBG_OP_DENSE_K>(d, groups, maxM, maxN, s, t);
    if (d.a_kind == VBG_OP_DENSE_K && d.b_kind == VBG_OP_DENSE_R) return launch_pair<VBG_OP_DENSE_K, VBG_OP_DENSE_R>(d, groups, maxM, maxN, s, t);
    if (d.a_kind == VBG_OP_DENSE_R && d.b_kind == VBG_OP_DENSE_R) return launch_pair<VBG_OP_DENSE_R, VBG_OP_DENSE_R>(d, groups, maxM, maxN, s, t);
    if (d.a_kind == VBG_OP_CONV_K && d.b_kind == VBG_OP_DENSE_K) return launch_pair<VBG_OP_CONV_K, VBG_OP_DENSE_K>(d, groups, maxM, maxN, s, t);
    if (d.a_kind == VBG_OP_CONV_K && d.b_kind == VBG_OP_WT_R) return launch_pair<VBG_OP_CONV_K, VBG_OP_WT_R>(d, groups, maxM, maxN, s, t);
    if (d.a_kind == VBG_OP_DENSE_R && d.b_kind == VBG_OP_CONV_R) {
        VBG_CHECK_ARG(d.N == d.geo.kh * d.geo.kw * d.geo.Cs);
        return launch_pair<VBG_OP_DENSE_R, VBG_OP_CONV_R>(d, groups, maxM, maxN, s, t);
    }
    return VBG_EARG;
}

extern "C" int vbg_gemm(const vbg_gemm_desc* desc, void* stream) { return gemm_dispatch(desc, stream, vbg::launch_timer{}); }

extern "C" int vbg_gemm_timed(const vbg_gemm_desc* desc, void* stream, void* start_event, void* stop_event) {
    VBG_CHECK_ARG(start_event && stop_event);
    vbg::launch_timer t;
    t.start = (hipEvent_t)start_event;
    t.stop = (hipEvent_t)stop_event;
    return gemm_dispatch(desc, stream, t);
}

extern "C" int vbg_timer_create(void** event) {
    VBG_CHECK_ARG(event);
    hipEvent_t e;
    const hipError_t err = hipEventCreate(&e);
    if (err != hipSuccess) return (int)err;
    *event = (void*)e;
    return VBG_OK;
}

extern "C" int vbg_timer_destroy(void* event) {
    VBG_CHECK_ARG(event);
    const hipError_t err = hipEventDestroy((hipEvent_t)event);
    return err == hipSuccess ? VBG_OK : (int)err;
}

extern "C" int vbg_timer_elapsed_ms(void* start_event, void* stop_event, float* ms) {
    VBG_CHECK_ARG(start_event && stop_event && ms);
    const hipError_t err = hipEventElapsedTime(ms, (hipEvent_t)start_event, (hipEvent_t)stop_event);
    return err == hipSuccess ? VBG_OK : (int)err;
}

// ---- deterministic split-K, second pass: the slabs added in split order, bias, optional ReLU (include/vbg.h vbg_gemm_desc.slab_stride) ----
namespace vbg {
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, int nslabs, long long slab_stride, int M, int N4, long long lds,
                                                          const float* __restrict__ bias, int relu, float* __restrict__ out, long long ldc) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)M * N4) return;
    const int m = (int)(i / N4), c = (int)(i - (long long)m * N4) * 4;
    float4 v = *reinterpret_cast<const float4*>(slabs + (long long)m * lds + c);
    for (int s = 1; s < nslabs; ++s) {
        const float4 w = *reinterpret_cast<const float4*>(slabs + s * slab_stride + (long long)m * lds + c);
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    if (bias) { const float4 b = *reinterpret_cast<const float4*>(bias + c); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    *reinterpret_cast<float4*>(out + (long long)m * ldc + c) = v;
}
}  // namespace vbg

extern "C" int vbg_slab_reduce(const float* slabs, int nslabs, long long slab_stride, int M, int N, long long lds, const float* bias, int relu,
                               float* out, long long ldc, void* stream) {
    VBG_CHECK_ARG(slabs && out && nslabs >= 1 && M >= 0 && N >= 0 && N % 4 == 0 && lds % 4 == 0 && ldc % 4 == 0 && slab_stride % 4 == 0 && lds >= N && ldc >= N);
    VBG_CHECK_ARG(((uintptr_t)slabs | (uintptr_t)out | (uintptr_t)bias) % 16 == 0);
    if (M == 0 || N == 0) return VBG_OK;
    const long long n = (long long)M * (N / 4);
    VBG_LAUNCH(vbg::slab_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, slabs, nslabs, slab_stride, M, N / 4, lds,
               bias, relu, out, ldc);
    VBG_LAUNCH_RET();
}
