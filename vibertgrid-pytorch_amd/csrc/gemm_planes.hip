// Plane GEMM: fp32-grade NT products from operands that are ALREADY split into bf16 planes in HBM.
//
// The split form of gemm.hip (PREC 3) splits every fp32 operand element into three bf16 pieces on its way into LDS -- per block,
// per tile it re-reads, ~5.5 VALU per element sharing the issue port with the MFMAs (matrix pipe busy 26-37 % in the BERT GEMMs,
// profiles/r01_gemm_mfma_util.txt).  Here the split happens ONCE per tensor (vbg_split_planes / vbg_split_planes_t, or the
// producing kernel's epilogue): an operand is three bf16 planes [3][rows][ld] (hi, mid, lo: x = hi + mid + lo exactly, 8 + 8 + 8
// significant bits by truncation), K-contiguous, ld a multiple of 32 (zero padded).  The k-loop then has no VALU work at all:
//   * global -> LDS by `buffer_load_dwordx4 ... lds` (LDS-DMA, 1 KiB = 16 rows x 64 B of one plane per wave instruction; rows past
//     the operand carry an out-of-range offset and land as zeros), the next k-tile in flight behind the current tile's MFMAs;
//   * LDS image [plane][row][32 k] bf16 with the 16-byte chunk index XOR-swizzled by (row >> 2) & 3 -- applied on the SOURCE
//     address of the DMA, the image itself is lane-linear -- so every ds_read_b128 fragment read is conflict free;
//   * per 16-deep k-step and 32x32 accumulator the six piece products of order <= 2^-16, smallest first:
//     (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi) as v_mfma_f32_32x32x16_bf16, fp32 accumulation.
// Arithmetic is identical to gemm.hip PREC 3 (same pieces, same products, same order), so results agree bit for bit up to the
// k-tile summation order.  Reference products replaced: every nn.Linear of transformers BertModel (model/BERTgrid_generator.py:134)
// forward / dgrad / wgrad, the 1x1 convolutions of model/ResNetFPN_ViBERTgrid.py, the heads' MLPs.
#include "vbg_common.h"
#include <type_traits>
#include <cstdlib>
#include <hip/hip_ext.h>
#include "../../include/vbg.h"

namespace vbg {

typedef unsigned pg_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 pg_bf16x8 __attribute__((ext_vector_type(8)));
constexpr unsigned PG_INVALID = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t pg_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)0x80000000u, 0x00020000);
}

typedef _Float16 pg_f16x8 __attribute__((ext_vector_type(8)));
// two fp16 pieces of a pair of floats (a in the low half), round to nearest: hi = fp16(x), lo' = fp16((x - hi) * 2^11)
__device__ __forceinline__ void pg_split2(float a, float b, unsigned& hi, unsigned& lo) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {a, b};
    const f16x2_t h = __builtin_convertvector(v, f16x2_t);
    const f16x2_t l = __builtin_convertvector((v - __builtin_convertvector(h, f32x2_t)) * 2048.f, f16x2_t);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
typedef short pg_v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) pg_v4s* pg_lds_v4s;

// TRANS = false: C[M,N] = A[M,K] B[N,K]^T, both operands K-contiguous ("NT").
// TRANS = true : C[M,N] = sum_k A[k,M] B[k,N] ("TN": the weight gradient dW = dY^T X straight from the UNtransposed planes of dY
//   [tokens][M] and X [tokens][N]; the reduction index is the operands' ROW index).  The LDS image of a k-tile is then [32 k][128
//   cols] per plane (256-byte rows, 64-byte chunks XOR-swizzled by k & 3) and a fragment -- 8 consecutive k of one column -- is
//   two ds_read_b64_tr_b16: each 16-lane group reads a [4 k][16 col] block, lane i supplying the address of (k = i / 4, cols
//   4 (i % 4)..+3) and receiving the 4 k of column i (lane semantics verified on the hardware, tools/probes/tr_probe.hip).
//   Rows past the reduction length fall off the per-plane buffer descriptors and read 0.
// FORM = 0: three bf16 planes per operand, six piece products.  FORM = 1 (NT only): two fp16 planes per operand -- hi = fp16(x), lo' =
// fp16((x - hi) 2^11), round to nearest: x = hi + lo' 2^-11 to 2^-23 |x| for 2^-14 <= |x| < 65520, absolute error <= 2^-36 below that,
// inf from 65520 on -- and three piece products: hi hi into the main accumulators, lo' hi + hi lo' into a second set that the epilogue
// scales by 2^-11.  Half the matrix-core work and 4 instead of 6 operand bytes per element, for operands inside fp16's range: the
// forward products, whose operands are LayerNorm / GELU outputs and weights.
// FORM = 2 (`amp`): the same fp16-pair operands, ONE product -- only the hi planes are loaded (x rounded to fp16, the reference's autocast
// operand; gradients scaled into range by their amax slots exactly as in FORM 1), fp32 accumulation.
template <int BM, int BN, int WGM, int WGN, int NST, bool TRANS, bool PP = true, int FORM = 0>
__global__ __launch_bounds__(WGM * WGN * 64) void plane_gemm_kernel(const vbg_plane_gemm_desc p) {
    static_assert(!TRANS || ((BM == 128 || BM == 256) && BN == 128), "TN: 128- or 256-column A tiles, 128-column B tiles");
    constexpr int NW = WGM * WGN, NT = NW * 64;
    constexpr int BK = 32;
    constexpr int NPL = FORM == 2 ? 1 : (FORM ? 2 : 3);        // planes per operand that are loaded
    // (round 6: ... and for the 4-wave 64 x 64 NT tile -- single documents: a 512-token product is 96 such tiles, and the form halves what each computes)
    static_assert(FORM == 0 || (NW == 8 && (TRANS || PP)) || (NW == 4 && !TRANS && BM == 64 && BN == 64),
                  "the fp16-pair form exists for the 8-wave tiles (NT: ping-pong loop) and the 64 x 64 NT tile");
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    constexpr int PA = BM * 64, PB = BN * 64;                  // bytes of one plane of a stage (64 B per row)
    constexpr int STAGE = NPL * (PA + PB);
    constexpr int NIA = NPL * BM / 16 / NW, NIB = NPL * BN / 16 / NW;      // DMA instructions per wave and stage
    static_assert((NPL * BM / 16) % NW == 0 && (NPL * BN / 16) % NW == 0, "DMA units must divide over the waves");
    // (round 6: the 4-wave 64 x 64 NT tile takes more stages -- a single document's products are a few hundred tiles that wait for COLD weights)
    static_assert(NST == 2 || NST == 3 || (NST == 4 && !TRANS && WGM * WGN == 8 && PP) || (NST >= 4 && NST <= 8 && !TRANS && WGM * WGN == 4 && BM == 64 && BN == 64),
                  "two or three LDS stages (four: ping-pong loop; four to eight: the 64 x 64 NT tile)");
    constexpr int CTS = BN + 4;
    constexpr int SMEM = (NST * STAGE > BM * CTS * 4) ? NST * STAGE : BM * CTS * 4;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];          // (the ONE LDS object of the kernel)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // problem of this block: the descriptor's own, or (grouped launch: several independent products of one reduction length in
    // one grid, so that their partial rounds of tiles share the chip) entry g of p.grp
    const unsigned short* pA = p.A;
    const unsigned short* pB = p.B;
    float* pC = p.C;
    int M = p.M, N = p.N;
    long long a_plane = p.a_plane, b_plane = p.b_plane, lda = p.lda, ldb = p.ldb, ldc_ = p.ldc;
    const unsigned* a_amax = p.a_amax;               // (FORM 1) the A operand was written scaled by the power of two of this slot
    // ---- XCD-aware block -> tile map (same scheme as gemm.hip): workgroup b runs on XCD b % 8; XCD k owns the k-th contiguous
    //      eighth of the tile sequence, and the sequence walks each problem's tile grid in bands of 8 row tiles ------------------
    constexpr unsigned XCDS = 8, XCD_GROUP = 8;
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned total = gridDim.x * gridDim.y * gridDim.z;
    const unsigned xcd = lin % XCDS, local = lin / XCDS;
    const unsigned per_xcd = (total + XCDS - 1) / XCDS, tall = (total % XCDS) ? (total % XCDS) : XCDS;
    const unsigned pid = xcd < tall ? xcd * per_xcd + local : tall * per_xcd + (xcd - tall) * (per_xcd - 1) + local;
    unsigned gx = gridDim.x, gy = gridDim.y;
    const unsigned slice = gridDim.x * gridDim.y;
    const int split = (int)(pid / slice);
    unsigned rem = pid - (unsigned)split * slice;
    if (p.ngroups > 0) {
        int g = 0;
        while (g + 1 < p.ngroups && rem >= (unsigned)(p.grp[g].tiles_m * p.grp[g].tiles_n)) { rem -= p.grp[g].tiles_m * p.grp[g].tiles_n; ++g; }
        pA = p.grp[g].A; pB = p.grp[g].B; pC = p.grp[g].C;
        M = p.grp[g].M; N = p.grp[g].N;
        a_plane = p.grp[g].a_plane; b_plane = p.grp[g].b_plane; lda = p.grp[g].lda; ldb = p.grp[g].ldb; ldc_ = p.grp[g].ldc;
        a_amax = p.grp[g].a_amax;
        gx = p.grp[g].tiles_m; gy = p.grp[g].tiles_n;
    }
    const unsigned band = XCD_GROUP * gy, bid = rem / band, first = bid * XCD_GROUP;
    const unsigned bm = min(gx - first, XCD_GROUP), inb = rem - bid * band;
    const unsigned tile_m = first + inb % bm, tile_n = inb / bm;
    const int m0 = (int)tile_m * BM, n0 = (int)tile_n * BN;
    if (m0 >= M || n0 >= N) return;
    const int nkt = (p.K + BK - 1) / BK;
    const int per = (nkt + p.splitk - 1) / p.splitk;
    const int kt0 = split * per, kt1 = min(nkt, kt0 + per);
    if (kt0 >= kt1) return;
    const int ntiles = kt1 - kt0;

    // ---- DMA addressing: lane l of a unit (plane q, 16-row block rb) fills LDS bytes [16 l, 16 l + 16) of the unit's 1 KiB =
    //      row rb*16 + l/4, physical chunk l%4, from logical chunk (l%4) ^ ((row >> 2) & 3) of that row ------------------------
    unsigned avo[NIA], bvo[NIB];
    int alds[NIA], blds[NIB];
    const unsigned short* abase;
    const unsigned short* bbase;
    long long a_rem = 0, b_rem = 0;                 // TRANS: bytes left in a plane behind the descriptor base (rows >= K read 0)
    if constexpr (!TRANS) {
        const int lrow = lane >> 2;
        const int lchunk = (lane & 3) ^ ((lane >> 4) & 3);
#pragma unroll
        for (int i = 0; i < NIA; ++i) {
            const int u = wave + NW * i, q = u / (BM / 16), rb = u % (BM / 16);
            const int r = rb * 16 + lrow;
            avo[i] = (m0 + r < M) ? (unsigned)(((long long)q * a_plane + (long long)r * lda) * 2 + lchunk * 16) : PG_INVALID;
            alds[i] = __builtin_amdgcn_readfirstlane(q * PA + rb * 1024);
        }
#pragma unroll
        for (int i = 0; i < NIB; ++i) {
            const int u = wave + NW * i, q = u / (BN / 16), rb = u % (BN / 16);
            const int r = rb * 16 + lrow;
            bvo[i] = (n0 + r < N) ? (unsigned)(((long long)q * b_plane + (long long)r * ldb) * 2 + lchunk * 16) : PG_INVALID;
            blds[i] = __builtin_amdgcn_readfirstlane(NPL * PA + q * PB + rb * 1024);
        }
        abase = pA + (long long)m0 * lda + (long long)kt0 * BK;
        bbase = pB + (long long)n0 * ldb + (long long)kt0 * BK;
    } else {
        // a k-row of an operand tile is RS = 2 * columns bytes (256 or 512); a 1 KiB DMA unit = 1024 / RS rows: lane l fills LDS bytes
        // [16 l, +16) of the unit = k-row rb * RPU + l / PPR, physical 16-byte piece l % PPR, from logical 64-byte chunk
        // (piece / 4) ^ (row & 3) (XOR on the two low bits of the chunk index)
        constexpr int PPRA = BM / 8, PPRB = BN / 8, RPUA = 64 / PPRA, RPUB = 64 / PPRB, UPA = 32 / RPUA, UPB = 32 / RPUB;
        const int mpad = (M + 31) / 32 * 32, npad = (N + 31) / 32 * 32;                   // (plane rows are zero padded to 32)
#pragma unroll
        for (int i = 0; i < NIA; ++i) {
            const int u = wave + NW * i, rb = u % UPA;
            const int lrow = lane / PPRA, piece = lane % PPRA, row = rb * RPUA + lrow;
            const int lcol = (((((piece >> 2) ^ (row & 3))) << 2) + (piece & 3)) * 8;     // source column (elements)
            avo[i] = (m0 + lcol < mpad) ? (unsigned)(((long long)row * lda + lcol) * 2) : PG_INVALID;
            alds[i] = __builtin_amdgcn_readfirstlane((u / UPA) * PA + rb * 1024);
        }
#pragma unroll
        for (int i = 0; i < NIB; ++i) {
            const int u = wave + NW * i, rb = u % UPB;
            const int lrow = lane / PPRB, piece = lane % PPRB, row = rb * RPUB + lrow;
            const int lcol = (((((piece >> 2) ^ (row & 3))) << 2) + (piece & 3)) * 8;
            bvo[i] = (n0 + lcol < npad) ? (unsigned)(((long long)row * ldb + lcol) * 2) : PG_INVALID;
            blds[i] = __builtin_amdgcn_readfirstlane(NPL * PA + (u / UPB) * PB + rb * 1024);
        }
        abase = pA + (long long)kt0 * BK * lda + m0;
        bbase = pB + (long long)kt0 * BK * ldb + n0;
        a_rem = ((long long)(p.K - kt0 * BK) * lda - m0) * 2;
        b_rem = ((long long)(p.K - kt0 * BK) * ldb - n0) * 2;
    }
    typedef __attribute__((address_space(3))) void* lds_ptr;
    // `inv` = PG_INVALID for a tile past the block's range: every lane's offset falls outside the descriptor, the DMA writes zeros
    // into a stage nobody reads and moves no memory -- the loop body stays branch free (one scheduling region per half iteration)
    // (NT form, ping-pong schedule: the A and the B share of a tile are issued in different intervals)
    auto issue_a = [&](int stage, unsigned inv) {
        unsigned char* sb = smem + stage * STAGE;
        const __amdgpu_buffer_rsrc_t ra = pg_rsrc(abase);
#pragma unroll
        for (int i = 0; i < NIA; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(sb + alds[i]), 16, (int)(avo[i] | inv), 0, 0, 0);
        abase += BK;
    };
    auto issue_b = [&](int stage, unsigned inv) {
        unsigned char* sb = smem + stage * STAGE;
        const __amdgpu_buffer_rsrc_t rb = pg_rsrc(bbase);
#pragma unroll
        for (int i = 0; i < NIB; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)(sb + blds[i]), 16, (int)(bvo[i] | inv), 0, 0, 0);
        bbase += BK;
    };
    auto issue = [&](int stage, unsigned inv) {
        unsigned char* sb = smem + stage * STAGE;
        if constexpr (!TRANS) {
            issue_a(stage, inv);
            issue_b(stage, inv);
        } else {
            // one descriptor per plane, ending behind the last row of the reduction (the plane of unit i of a wave is a compile-time
            // constant: units are dealt to the waves plane by plane)
            constexpr int UPA = 32 / (64 / (BM / 8)), UPB = 32 / (64 / (BN / 8));
            static_assert(UPA % NW == 0 && UPB % NW == 0, "TN: a plane's DMA units divide over the waves");
            const int na = (int)(a_rem < 0 ? 0 : (a_rem > 0x7fffffffll ? 0x7fffffffll : a_rem));
            const int nb = (int)(b_rem < 0 ? 0 : (b_rem > 0x7fffffffll ? 0x7fffffffll : b_rem));
#pragma unroll
            for (int i = 0; i < NIA; ++i) {
                const int q = (NW * i) / UPA;
                const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(abase + q * a_plane), 0, na, 0x00020000);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(sb + alds[i]), 16, (int)(avo[i] | inv), 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NIB; ++i) {
                const int q = (NW * i) / UPB;
                const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(bbase + q * b_plane), 0, nb, 0x00020000);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)(sb + blds[i]), 16, (int)(bvo[i] | inv), 0, 0, 0);
            }
            abase += (long long)BK * lda;
            bbase += (long long)BK * ldb;
            a_rem -= (long long)BK * lda * 2;
            b_rem -= (long long)BK * ldb * 2;
        }
    };

    // ---- fragments ---------------------------------------------------------------------------------------------------
    const int wm = wave / WGN, wn = wave % WGN;
    const int lr = lane & 31, lk = lane >> 5;
    const int sw = (lr >> 2) & 3;
    // byte offset of lane's 16-byte fragment of k-step s inside a 32-row fragment block: row lr, logical chunk 2 s + lk
    // (TRANS: k-step 1 starts 16 k-rows further: 16 * RS bytes, applied per operand in read_frags)
    const int fo0 = TRANS ? 0 : lr * 64 + (((0 + lk) ^ sw) << 4), fo1 = TRANS ? 16 : lr * 64 + (((2 + lk) ^ sw) << 4);
    // TRANS: byte offset of this lane's tr-read address inside a plane tile for fragment block i (32 columns), k-step 0, half 0
    const int i16 = lane & 15, tg = (lane >> 4) & 1;
    int ta[TM], tb[TN];
    constexpr int RSA = BM * 2, RSB = BN * 2;                      // bytes of a k-row of the A / B tile image
#pragma unroll
    for (int i = 0; i < TM; ++i) ta[i] = (8 * lk + (i16 >> 2)) * RSA + (((wm * TM + i) ^ (i16 >> 2)) << 6) + (16 * tg + 4 * (i16 & 3)) * 2;
#pragma unroll
    for (int j = 0; j < TN; ++j) tb[j] = (8 * lk + (i16 >> 2)) * RSB + (((wn * TN + j) ^ (i16 >> 2)) << 6) + (16 * tg + 4 * (i16 & 3)) * 2;
    f32x16 acc[TM][TN], acx[FORM == 1 ? TM : 1][FORM == 1 ? TN : 1];         // (FORM 1: the cross products, scaled by 2^11)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][j][r] = 0.f;
                if constexpr (FORM == 1) acx[i][j][r] = 0.f;
            }

    pg_u32x4 fa0[NPL][TM], fb0[NPL][TN], fa1[NPL][TM], fb1[NPL][TN];       // fragment sets of k-step 0 / 1 of a tile
    auto read_frags = [&](int stage, int fo, pg_u32x4 (&fa)[NPL][TM], pg_u32x4 (&fb)[NPL][TN]) {
        if constexpr (TRANS) {
            typedef __attribute__((address_space(3))) unsigned char* lds_bytes;
            lds_bytes as = (lds_bytes)(smem + stage * STAGE + fo * RSA), bs = (lds_bytes)(smem + stage * STAGE + NPL * PA + fo * RSB);
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const pg_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pg_lds_v4s)(as + q * PA + ta[i]));
                    const pg_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pg_lds_v4s)(as + q * PA + ta[i] + 4 * RSA));
                    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                    fa[q][i] = pg_u32x4{l2.x, l2.y, h2.x, h2.y};
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const pg_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pg_lds_v4s)(bs + q * PB + tb[j]));
                    const pg_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pg_lds_v4s)(bs + q * PB + tb[j] + 4 * RSB));
                    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                    fb[q][j] = pg_u32x4{l2.x, l2.y, h2.x, h2.y};
                }
            }
            return;
        }
        const unsigned char* as = smem + stage * STAGE + (wm * WM) * 64 + fo;
        const unsigned char* bs = smem + stage * STAGE + NPL * PA + (wn * WN) * 64 + fo;
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[q][i] = *reinterpret_cast<const pg_u32x4*>(as + q * PA + i * 32 * 64);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[q][j] = *reinterpret_cast<const pg_u32x4*>(bs + q * PB + j * 32 * 64);
        }
    };
    // piece products, smallest first: (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi); FORM 1: (lo',hi) (hi,lo') -> cross sums, (hi,hi)
    auto mma = [&](const pg_u32x4 (&fa)[NPL][TM], const pg_u32x4 (&fb)[NPL][TN]) {
        if constexpr (FORM == 2) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pg_f16x8, fa[0][i]), __builtin_bit_cast(pg_f16x8, fb[0][j]), acc[i][j], 0, 0, 0);
        } else if constexpr (FORM) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        f32x16& d = t < 2 ? acx[i][j] : acc[i][j];
                        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pg_f16x8, fa[t == 0 ? 1 : 0][i]),
                                                                   __builtin_bit_cast(pg_f16x8, fb[t == 1 ? 1 : 0][j]), d, 0, 0, 0);
                    }
        } else {
            constexpr int qa[6] = {2, 0, 1, 1, 0, 0}, qb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pg_bf16x8, fa[qa[t]][i]),
                                                                            __builtin_bit_cast(pg_bf16x8, fb[qb[t]][j]), acc[i][j], 0, 0, 0);
        }
    };

    // ---- k loop ---------------------------------------------------------------------------------------------------------
    // Fragments are read one k-step ahead of the MFMAs that use them (set 1 of tile t behind the products of set 0, set 0 of tile
    // t+1 behind the products of set 1), so ds_read latency never sits between two MFMA groups.
    // NST = 3: tile t+2 travels while tile t is multiplied; tile t+1 landed an iteration ago and is published by the ONE barrier
    // in the middle of iteration t (counted vmcnt: the DMA of tile t+2 stays in flight across it), after which set 0 of tile t+1
    // is read -- no bubble after the barrier.  NST = 2 (the 8-wave tile, 2 waves per SIMD cover each other): tile t+1 travels
    // during tile t, barrier at the end of the iteration.
    constexpr int NIW = NIA + NIB;
    constexpr int NMMA = (FORM == 2 ? 1 : (FORM ? 3 : 6)) * TM * TN, NRD = (TRANS ? 2 * NPL : NPL) * (TM + TN);
    // ---- ping-pong schedule of the 8-wave NT tiles --------------------------------------------------------------------------
    // Waves w and w + 4 of a workgroup share a SIMD (measured: tools/probes/pingpong_gemm_probe.hip prints HW_ID).  In lockstep both
    // issue their DMA, read their fragments and then want the matrix pipe at the same moments: matrix-pipe busy 0.40-0.47 of a full
    // round of tiles.  Here the waves form two groups (0-3 / 4-7: the two waves of every SIMD in different groups) that run ONE
    // barrier interval apart: while a group issues the 12 MFMAs of a 16-deep k-step, its partners issue DMA and read the fragments of
    // their next k-step.  Intervals of a tile t per group: L0 (DMA + fragments of k-step 0) | M0 | L1 (DMA + fragments of k-step 1)
    // | M1, a barrier between any two.  Hazards (group 1 runs one interval behind group 0):
    //   NST = 3: tile t + 2 is issued during tile t (A share in L0, B share in L1) into the stage of tile t - 1, whose last readers
    //            (group 1's L1 of tile t - 1) finished one interval before group 0's L0 of tile t; every wave waits for its share of
    //            tile t + 1 at the end of L1 (counted vmcnt: tile t + 2 stays in flight), two barriers before anybody reads it;
    //   NST = 2: all of tile t + 1 is issued in L0 of tile t (same argument) and waited for at the end of L1.
    // Every accumulator sees the same MFMAs in the same order as in the lockstep loop: results are bit-identical.
    if constexpr (!TRANS && NW == 8 && PP) {
        const int grp = wave >> 2;
        //   NST = 4: the same with tile t + 3 (two tiles stay in flight across the wait): the L2 misses of a stage -- one row piece in
        //            seven comes from the Infinity Cache at 3000+ clocks under load -- then have three k-tiles of products to hide behind
        issue(0, 0u);
        if constexpr (NST >= 3) {
#pragma unroll
            for (int s_ = 1; s_ < NST - 1; ++s_) issue(s_, ntiles > s_ ? 0u : PG_INVALID);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * NIW) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                       // tile 0 has landed (every wave's share)
        if (grp) __builtin_amdgcn_s_barrier();              // group 1 starts one interval late
        int cur = 0;
        for (int t = 0; t < ntiles; ++t) {
            const int nxt = (cur + 1 == NST) ? 0 : cur + 1;
            const int nn = (cur == 0) ? NST - 1 : cur - 1;  // the stage of tile t - 1: where tile t + NST - 1 goes
            const unsigned inv = (t + NST - 1 < ntiles) ? 0u : PG_INVALID;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NST >= 3) issue_a(nn, inv); else issue(nxt, inv);
            read_frags(cur, fo0, fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            mma(fa0, fb0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NST >= 3) issue_b(nn, inv);
            read_frags(cur, fo1, fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NST >= 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NST - 2) * NIW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            mma(fa0, fb0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            cur = nxt;
        }
        if (!grp) __builtin_amdgcn_s_barrier();             // group 0 waits for group 1's last interval
    } else {
    issue(0, 0u);
    if constexpr (NST >= 3) {
#pragma unroll
        for (int s_ = 1; s_ < NST - 1; ++s_) issue(s_, ntiles > s_ ? 0u : PG_INVALID);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * NIW) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    auto settle0 = [&]() {
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
#pragma unroll
            for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(fa0[q][i]));
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(fb0[q][j]));
        }
    };
    read_frags(0, fo0, fa0, fb0);
    settle0();
    int cur = 0;                                   // stage of tile t
    for (int t = 0; t < ntiles; ++t) {
        const int nxt = (cur + 1 == NST) ? 0 : cur + 1;
        const int far = (cur == 0) ? NST - 1 : cur - 1;          // stage of tile t + NST - 1 = the one tile t - 1 left (NST = 3: cur + 2)
        // ---- first half: products of set 0; set 1 of this tile is read and the DMA of tile t+NST-1 is issued in their gaps ----
        __builtin_amdgcn_sched_barrier(0);
        issue(NST >= 3 ? far : nxt, t + NST - 1 < ntiles ? 0u : PG_INVALID);
        read_frags(cur, fo1, fa1, fb1);
        mma(fa0, fb0);
        // (a 32x32x16 bf16 MFMA holds the matrix pipe for 32 cycles = ~8 issue slots: one LDS read and one DMA per gap ride free)
#pragma unroll
        for (int g = 0; g < NMMA; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (g < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (g < NIW) { __builtin_amdgcn_sched_group_barrier(0x004, 2, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NST >= 3) {
            // tile t+1 must have landed (this wave's share; the DMAs of tiles t+2 .. t+NST-1 stay in flight), and this wave's reads of the
            // stage the NEXT iteration's DMA overwrites must be done, before anybody passes the barrier
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NST - 2) * NIW) : "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            read_frags(nxt, fo0, fa0, fb0);          // (unconditional: past the last tile it reads stale LDS that nobody uses)
            mma(fa1, fb1);
#pragma unroll
            for (int g = 0; g < NMMA; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (g < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        } else {
            mma(fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            read_frags(nxt, fo0, fa0, fb0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // (compiler bookkeeping: a use of the freshly read set-0 registers at the END of the iteration, behind the MFMAs, makes
        // hipcc place its lgkmcnt wait for them here -- where the data arrived long ago -- instead of a conservative lgkmcnt(0)
        // in front of the next iteration's first MFMA, which would also wait for the set-1 reads issued just before it)
        settle0();
        cur = nxt;
    }
    }
    __syncthreads();                               // every wave is done with the operand stages: the output tile is staged there

    // ---- epilogue: staged through LDS, float4 row pieces ------------------------------------------------------------------
    const float* bias = p.bias;
    const bool add_bias = (bias != nullptr) && (split == 0);
    const int accumulate = p.accumulate, epi = p.epi;
    const bool atomic = accumulate && p.splitk > 1;
    // a gradient operand of the fp16-pair form was multiplied by a power of two when its planes were written (the power that brings
    // the tensor's largest magnitude to [2^13, 2^14), vbg_split_planes_pair): the product is scaled back here -- exact
    float alpha = p.alpha;
    if constexpr (FORM) { if (a_amax) alpha *= vbg_pow2_scale(vbg_amax_read(a_amax)).y; }
    const long long ldc = ldc_;
    float* const C = pC;
    float* const C2 = p.C2;
    float* const Ct = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Ct[(wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * CTS + wn * WN + j * 32 + lr] =
                    (FORM == 1 ? acc[i][j][r] + acx[FORM == 1 ? i : 0][FORM == 1 ? j : 0][r] * (1.f / 2048.f) : acc[i][j][r]) * alpha;
    __syncthreads();
    constexpr int QN = BN / 4;
    // optional bf16 planes of the stored value (the A operand of the next product): [3][M][ldp], K-contiguous = along n here
    unsigned short* const Cp = p.Cp;
    unsigned short* const Cq = p.Cq;                 // optional fp16-pair planes of the stored value [2][M][ldq]
    float vmax = 0.f;                                // max |stored value| of this thread (p.c_amax)
    // Cq of a GRADIENT (cq_ref_in): scaled by the power of two of a rigorous bound of the stored values, max |A| * max row-L1 of B *
    // cq_mul -- every workgroup derives the same bound from the same two device words; one thread publishes its bit pattern for the
    // consumers (their a_amax)
    float qsc = 1.f;
    if (Cq && p.cq_ref_in) {                         // (uniform)
        const float amx = __uint_as_float(vbg_amax_read(p.cq_ref_in));
        const float l1 = p.cq_l1_in ? __uint_as_float(*p.cq_l1_in) : 1.f;
        const float bound = amx * l1 * p.cq_mul;
        qsc = vbg_pow2_scale(__float_as_uint(bound)).x;
        if (p.cq_ref_out && pid == 0 && tid == 0) p.cq_ref_out[0] = __float_as_uint(bound);
    }
#pragma unroll
    for (int q = 0; q < BM * QN / NT; ++q) {
        const int idx = tid + q * NT;
        const int row = idx / QN, c = (idx % QN) * 4;
        const int gm = m0 + row, gn = n0 + c;
        if (gm >= M || gn >= N) continue;
        float4 v = *reinterpret_cast<const float4*>(&Ct[row * CTS + c]);
        if (add_bias) {
            const float4 b = *reinterpret_cast<const float4*>(bias + gn);
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        float* cp = C + (long long)gm * ldc + gn;
        if (atomic) {
            unsafeAtomicAdd(cp + 0, v.x); unsafeAtomicAdd(cp + 1, v.y); unsafeAtomicAdd(cp + 2, v.z); unsafeAtomicAdd(cp + 3, v.w);
            continue;
        }
        if (accumulate) {
            const float4 o = *reinterpret_cast<const float4*>(cp);
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        if (epi == VBG_EPI_RELU) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        if (epi == VBG_EPI_MUL_GELU_GRAD) {          // C2 = h (input): the product is dL/d gelu(h), the stored value dL/dh
            const float4 hv = *reinterpret_cast<const float4*>(C2 + (long long)gm * ldc + gn);
            v.x *= gelu_erf_grad(hv.x); v.y *= gelu_erf_grad(hv.y); v.z *= gelu_erf_grad(hv.z); v.w *= gelu_erf_grad(hv.w);
        }
        if (C) *reinterpret_cast<float4*>(cp) = v;
        vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        if (p.colsum) *reinterpret_cast<float4*>(&Ct[row * CTS + c]) = v;      // (the stored value goes back for the column sums below)
        if (epi == VBG_EPI_GELU_DUAL) {
            v = make_float4(gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w));
            if (C2) *reinterpret_cast<float4*>(C2 + (long long)gm * ldc + gn) = v;
        }
        if (Cp) {
            const float e[4] = {v.x, v.y, v.z, v.w};
            unsigned short h[4], m[4], l[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const unsigned u = __float_as_uint(e[t]);
                const float r1 = e[t] - __uint_as_float(u & 0xffff0000u);
                const unsigned u1 = __float_as_uint(r1);
                const float r2 = r1 - __uint_as_float(u1 & 0xffff0000u);
                h[t] = (unsigned short)(u >> 16); m[t] = (unsigned short)(u1 >> 16); l[t] = (unsigned short)(__float_as_uint(r2) >> 16);
            }
            unsigned short* o = Cp + (long long)gm * p.ldp + gn;
            *reinterpret_cast<uint2*>(o) = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
            *reinterpret_cast<uint2*>(o + p.c_plane) = make_uint2(m[0] | ((unsigned)m[1] << 16), m[2] | ((unsigned)m[3] << 16));
            *reinterpret_cast<uint2*>(o + 2 * p.c_plane) = make_uint2(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16));
        }
        if (Cq) {
            unsigned short* o = Cq + (long long)gm * p.ldq + gn;
            uint2 h, l;
            pg_split2(v.x * qsc, v.y * qsc, h.x, l.x);
            pg_split2(v.z * qsc, v.w * qsc, h.y, l.y);
            *reinterpret_cast<uint2*>(o) = h;
            *reinterpret_cast<uint2*>(o + p.q_plane) = l;
        }
    }
    // optional column sums of the stored values (the bias gradient of the layer whose dL/d(output) this product produces): one thread
    // per column of the staged tile, one atomic per column and row tile (rows past M hold zeros: their operand rows were read as zeros)
    if (p.colsum) {                                   // (uniform)
        __syncthreads();
        if (tid < BN && n0 + tid < N) {
            float cs = 0.f;
            const int rows = min(BM, M - m0);
            for (int r = 0; r < rows; ++r) cs += Ct[r * CTS + tid];
            unsafeAtomicAdd(p.colsum + n0 + tid, cs);
        }
    }
    // optional: the largest magnitude this launch stored (the scale of the fp16-pair planes a split pass makes of C next).  LAST: the
    // publish borrows the first words of the staged tile as its per-wave scratch, and the column sums above read that tile
    if (p.c_amax) {                                   // (uniform)
        __syncthreads();
        vbg_amax_publish(vmax, p.c_amax, Ct);
    }
}

__device__ __forceinline__ void pg_split3(float x, unsigned short& h, unsigned short& m, unsigned short& l);

// ---- 256 x 256 tile form of the NT product -----------------------------------------------------------------------------------------
// Every matrix kernel of this library levels off where its CUs ingest ~12-13 bytes per clock from L2 / fabric (the 128 x 128 plane
// tile needs 31 B / clk / CU at full matrix rate -> 0.38-0.40 MFMA utilisation measured, the 256 x 128 tile 23 B -> 0.49, the fp32
// operands of gemm.hip's 128 x 128 x 16 tile 21 B -> 0.56-0.60): the lever is products per byte moved.  A 256 x 256 tile halves the
// bytes per product again (16 B / clk / CU at full rate).  Its stages only fit as 16-deep k-tiles (3 planes x 512 rows x 32 B =
// 48 KB, three of them in flight); 8 waves as 2 x 4, a wave owns 128 x 64 (eight 32 x 32 accumulators), one k-step = 48 MFMAs per
// barrier.  LDS image [plane][row][32 B], the two 16-byte halves of a row swapped where (row >> 3) & 1 (on the DMA's source
// address), which spreads the ds_read_b128 fragment reads of 32-byte rows over all banks.  The output tile is staged through LDS in
// four passes of 64 rows.  Used where the tile count still fills the chip's CUs reasonably (N >= 2304 at M = 4128).
__global__ __launch_bounds__(512) void plane_gemm_256_kernel(const vbg_plane_gemm_desc p) {
    constexpr int BM = 256, BN = 256, WGN = 4, NW = 8, NT = 512, NST = 3, BK = 16;
    constexpr int WM = 128, WN = 64, TM = 4, TN = 2;
    constexpr int PA = BM * 32, PB = BN * 32;                  // bytes of one plane of a stage (32 B per row)
    constexpr int STAGE = 3 * (PA + PB);
    constexpr int NIA = 3, NIB = 3;                            // DMA instructions per wave and stage (1 KiB = 32 rows x 32 B each)
    constexpr int CTS = BN + 4;
    constexpr int SMEM = NST * STAGE;
    static_assert(64 * CTS * 4 <= SMEM, "epilogue pass fits");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];          // (the ONE LDS object of the kernel)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int M = p.M, N = p.N;
    const long long a_plane = p.a_plane, b_plane = p.b_plane, lda = p.lda, ldb = p.ldb;
    // XCD-aware block -> tile map (as above)
    constexpr unsigned XCDS = 8, XCD_GROUP = 8;
    const unsigned gx = gridDim.x, gy = gridDim.y;
    const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
    const unsigned total = gx * gy;
    const unsigned xcd = lin % XCDS, local = lin / XCDS;
    const unsigned per_xcd = (total + XCDS - 1) / XCDS, tall = (total % XCDS) ? (total % XCDS) : XCDS;
    const unsigned rem = xcd < tall ? xcd * per_xcd + local : tall * per_xcd + (xcd - tall) * (per_xcd - 1) + local;
    const unsigned band = XCD_GROUP * gy, bid = rem / band, first = bid * XCD_GROUP;
    const unsigned bm = min(gx - first, XCD_GROUP), inb = rem - bid * band;
    const unsigned tile_m = first + inb % bm, tile_n = inb / bm;
    const int m0 = (int)tile_m * BM, n0 = (int)tile_n * BN;
    if (m0 >= M || n0 >= N) return;
    const int ntiles = p.K / BK;

    // DMA: unit (plane q, 32-row block rb) of an operand; lane l -> row rb*32 + l/2, physical half l%2 <- logical half (l%2) ^ (row>>3 & 1)
    unsigned avo[NIA], bvo[NIB];
    int alds[NIA], blds[NIB];
    {
        const int lrow = lane >> 1;
#pragma unroll
        for (int i = 0; i < NIA; ++i) {
            const int u = wave + NW * i, q = u / 8, rb = u % 8;
            const int r = rb * 32 + lrow;
            const int half = (lane & 1) ^ ((r >> 3) & 1);
            avo[i] = (m0 + r < M) ? (unsigned)(((long long)q * a_plane + (long long)r * lda) * 2 + half * 16) : PG_INVALID;
            alds[i] = __builtin_amdgcn_readfirstlane(q * PA + rb * 1024);
        }
#pragma unroll
        for (int i = 0; i < NIB; ++i) {
            const int u = wave + NW * i, q = u / 8, rb = u % 8;
            const int r = rb * 32 + lrow;
            const int half = (lane & 1) ^ ((r >> 3) & 1);
            bvo[i] = (n0 + r < N) ? (unsigned)(((long long)q * b_plane + (long long)r * ldb) * 2 + half * 16) : PG_INVALID;
            blds[i] = __builtin_amdgcn_readfirstlane(3 * PA + q * PB + rb * 1024);
        }
    }
    const unsigned short* abase = p.A + (long long)m0 * lda;
    const unsigned short* bbase = p.B + (long long)n0 * ldb;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto issue = [&](int stage, unsigned inv) {
        unsigned char* sb = smem + stage * STAGE;
        const __amdgpu_buffer_rsrc_t ra = pg_rsrc(abase), rb = pg_rsrc(bbase);
#pragma unroll
        for (int i = 0; i < NIA; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(sb + alds[i]), 16, (int)(avo[i] | inv), 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NIB; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)(sb + blds[i]), 16, (int)(bvo[i] | inv), 0, 0, 0);
        abase += BK;
        bbase += BK;
    };

    const int wm = wave / WGN, wn = wave % WGN;
    const int lr = lane & 31, lk = lane >> 5;
    // fragment: row lr of a 32-row block (block bases are multiples of 32 rows, so (row >> 3) & 1 == (lr >> 3) & 1), logical half lk
    const int fo = lr * 32 + ((lk ^ ((lr >> 3) & 1)) << 4);
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    pg_u32x4 fa[3][TM], fb[3][TN];
    auto read_frags = [&](int stage) {
        const unsigned char* as = smem + stage * STAGE + (wm * WM) * 32 + fo;
        const unsigned char* bs = smem + stage * STAGE + 3 * PA + (wn * WN) * 32 + fo;
        // (in the order the piece products need them: lo / hi first)
        constexpr int order[3] = {2, 0, 1};
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const int q = order[o];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[q][i] = *reinterpret_cast<const pg_u32x4*>(as + q * PA + i * 32 * 32);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[q][j] = *reinterpret_cast<const pg_u32x4*>(bs + q * PB + j * 32 * 32);
        }
    };
    auto mma = [&]() {
        constexpr int qa[6] = {2, 0, 1, 1, 0, 0}, qb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pg_bf16x8, fa[qa[t]][i]),
                                                                        __builtin_bit_cast(pg_bf16x8, fb[qb[t]][j]), acc[i][j], 0, 0, 0);
    };
    constexpr int NIW = NIA + NIB;
    issue(0, 0u);
    issue(1, ntiles > 1 ? 0u : PG_INVALID);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIW) : "memory");
    __builtin_amdgcn_s_barrier();
    int cur = 0;
    for (int t = 0; t < ntiles; ++t) {
        const int nxt = (cur + 1 == NST) ? 0 : cur + 1;
        const int nn = (nxt + 1 == NST) ? 0 : nxt + 1;
        issue(nn, t + 2 < ntiles ? 0u : PG_INVALID);         // (stage of tile t - 1: every wave read it before the last barrier)
        read_frags(cur);
        mma();
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NIW) : "memory");      // tile t + 1 landed; tile t + 2 stays in flight
        __builtin_amdgcn_s_barrier();
        cur = nxt;
    }
    __syncthreads();                               // (drains the DMA queue: the output tile is staged over the operand stages)

    // ---- epilogue: four passes of 64 rows through LDS, float4 row pieces ------------------------------------------------------
    const float* bias = p.bias;
    const int epi = p.epi;
    const float alpha = p.alpha;
    const long long ldc = p.ldc;
    float* const C = p.C;
    float* const C2 = p.C2;
    unsigned short* const Cp = p.Cp;
    float* const Ct = reinterpret_cast<float*>(smem);
    constexpr int QN = BN / 4;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        if (wm == (ps >> 1)) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        Ct[(ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * CTS + wn * WN + j * 32 + lr] = acc[2 * (ps & 1) + ii][j][r] * alpha;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 64 * QN / NT; ++q) {
            const int idx = tid + q * NT;
            const int row = idx / QN, c = (idx % QN) * 4;
            const int gm = m0 + ps * 64 + row, gn = n0 + c;
            if (gm >= M || gn >= N) continue;
            float4 v = *reinterpret_cast<const float4*>(&Ct[row * CTS + c]);
            if (bias) {
                const float4 b = *reinterpret_cast<const float4*>(bias + gn);
                v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
            }
            float* cp = C + (long long)gm * ldc + gn;
            if (epi == VBG_EPI_RELU) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
            if (epi == VBG_EPI_MUL_GELU_GRAD) {
                const float4 hv = *reinterpret_cast<const float4*>(C2 + (long long)gm * ldc + gn);
                v.x *= gelu_erf_grad(hv.x); v.y *= gelu_erf_grad(hv.y); v.z *= gelu_erf_grad(hv.z); v.w *= gelu_erf_grad(hv.w);
            }
            if (C) *reinterpret_cast<float4*>(cp) = v;
            if (epi == VBG_EPI_GELU_DUAL) {
                v = make_float4(gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w));
                if (C2) *reinterpret_cast<float4*>(C2 + (long long)gm * ldc + gn) = v;
            }
            if (Cp) {
                const float e[4] = {v.x, v.y, v.z, v.w};
                unsigned short h[4], m[4], l[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) pg_split3(e[t], h[t], m[t], l[t]);
                unsigned short* o = Cp + (long long)gm * p.ldp + gn;
                *reinterpret_cast<uint2*>(o) = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
                *reinterpret_cast<uint2*>(o + p.c_plane) = make_uint2(m[0] | ((unsigned)m[1] << 16), m[2] | ((unsigned)m[3] << 16));
                *reinterpret_cast<uint2*>(o + 2 * p.c_plane) = make_uint2(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16));
            }
        }
        __syncthreads();
    }
}

// ---- stream-K form of the NT product ------------------------------------------------------------------------------------------
// One block per CU holds a 128 x 128 tile's stages, so a product runs in ROUNDS of 256 tiles and the BERT shapes end in a
// round that is a quarter full (4128 x 768: 198 tiles, 4128 x 2304: 594 = 2.32 rounds, 4128 x 3072: 792 = 3.09 rounds).  Here the
// first `sk_full` tiles (whole rounds) run one block per tile as above; the k-tile sequence of the remaining tiles is cut into
// `sk_blocks` equal contiguous ranges, one per block.  A range touches one or two tiles; for each it writes its partial 128 x 128
// sums to a slab (write-through stores), draws a ticket on the tile's counter, and the block that draws the last ticket adds the
// slabs of all contributors IN BLOCK ORDER (its own included: the result does not depend on who arrives last) and runs the
// epilogue.  No block ever waits for another.  MEASURED (MI355X, cfg2 BERT shapes): slower than the plain rounds -- the two 64 KB slab
// publications per block and the slab reads of the reducers cost 20-50 us per launch against the <= 25 us an evenly filled chip
// would save (4128x2304x768: 112 vs 100 us, 4128x768x3072: 139 vs 110 us) -- so the library uses it only on request (sk_ws != NULL).  Ordering: sc1 (write-through) slab stores, every wave drains its stores, barrier,
// relaxed agent-scope ticket; the last arriver issues ONE agent-scope acquire before it reads the slabs.
template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(WGM * WGN * 64) void plane_gemm_sk_kernel(const vbg_plane_gemm_desc p) {
    constexpr int NW = WGM * WGN, NT = NW * 64, NST = 3;
    constexpr int BK = 32;
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    constexpr int PA = BM * 64, PB = BN * 64;
    constexpr int STAGE = 3 * (PA + PB);
    constexpr int NIA = 3 * BM / 16 / NW, NIB = 3 * BN / 16 / NW;
    constexpr int CTS = BN + 4;
    constexpr int CBYTES = BM * CTS * 4;
    constexpr int SMEM = (NST * STAGE > CBYTES + 16) ? NST * STAGE : CBYTES + 16;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];          // (the ONE LDS object of the kernel)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int M = p.M, N = p.N;
    const long long a_plane = p.a_plane, b_plane = p.b_plane, lda = p.lda, ldb = p.ldb;
    const int nkt = (p.K + BK - 1) / BK;
    const unsigned gx = p.sk_tiles_m, gy = p.sk_tiles_n;
    const unsigned full = p.sk_full, nsk = p.sk_blocks, ntiles_all = gx * gy, rtiles = ntiles_all - full;
    const unsigned lin = blockIdx.x;
    // units of this block: ONE whole tile (lin < full, XCD-aware order over the whole-round tiles) or a k-range of the tail tiles
    unsigned seg_tile[2];
    int seg_k0[2], seg_k1[2], nseg;
    const bool sk = lin >= full;
    const unsigned skb = lin - full;
    const long long units = (long long)rtiles * nkt;
    auto ub = [&](unsigned b) { return (long long)b * units / nsk; };             // first unit of tail block b
    if (!sk) {
        constexpr unsigned XCDS = 8;
        const unsigned xcd = lin % XCDS, local = lin / XCDS;
        const unsigned per_xcd = (full + XCDS - 1) / XCDS, tall = (full % XCDS) ? (full % XCDS) : XCDS;
        seg_tile[0] = xcd < tall ? xcd * per_xcd + local : tall * per_xcd + (xcd - tall) * (per_xcd - 1) + local;
        seg_k0[0] = 0; seg_k1[0] = nkt; nseg = 1;
    } else {
        const long long u0 = ub(skb), u1 = ub(skb + 1);
        nseg = 0;
        long long u = u0;
        while (u < u1 && nseg < 2) {
            const unsigned t = (unsigned)(u / nkt);
            const int k0 = (int)(u - (long long)t * nkt);
            const int k1 = (int)min((long long)nkt, k0 + (u1 - u));
            seg_tile[nseg] = full + t; seg_k0[nseg] = k0; seg_k1[nseg] = k1; ++nseg;
            u += k1 - k0;
        }
    }

    const int wm = wave / WGN, wn = wave % WGN;
    const int lr = lane & 31, lk = lane >> 5;
    const int sw = (lr >> 2) & 3;
    const int fo0 = lr * 64 + (((0 + lk) ^ sw) << 4), fo1 = lr * 64 + (((2 + lk) ^ sw) << 4);
    typedef __attribute__((address_space(3))) void* lds_ptr;
    constexpr unsigned XCD_GROUP = 8;

    for (int sg = 0; sg < nseg; ++sg) {
        const unsigned rem = seg_tile[sg];
        const unsigned band = XCD_GROUP * gy, bid = rem / band, first = bid * XCD_GROUP;
        const unsigned bm = min(gx - first, XCD_GROUP), inb = rem - bid * band;
        const unsigned tile_m = first + inb % bm, tile_n = inb / bm;
        const int m0 = (int)tile_m * BM, n0 = (int)tile_n * BN;
        const int kt0 = seg_k0[sg], kt1 = seg_k1[sg];
        const int ntiles = kt1 - kt0;
        const bool partial = ntiles < nkt;

        unsigned avo[NIA], bvo[NIB];
        int alds[NIA], blds[NIB];
        {
            const int lrow = lane >> 2;
            const int lchunk = (lane & 3) ^ ((lane >> 4) & 3);
#pragma unroll
            for (int i = 0; i < NIA; ++i) {
                const int u = wave + NW * i, q = u / (BM / 16), rb = u % (BM / 16);
                const int r = rb * 16 + lrow;
                avo[i] = (m0 + r < M) ? (unsigned)(((long long)q * a_plane + (long long)r * lda) * 2 + lchunk * 16) : PG_INVALID;
                alds[i] = __builtin_amdgcn_readfirstlane(q * PA + rb * 1024);
            }
#pragma unroll
            for (int i = 0; i < NIB; ++i) {
                const int u = wave + NW * i, q = u / (BN / 16), rb = u % (BN / 16);
                const int r = rb * 16 + lrow;
                bvo[i] = (n0 + r < N) ? (unsigned)(((long long)q * b_plane + (long long)r * ldb) * 2 + lchunk * 16) : PG_INVALID;
                blds[i] = __builtin_amdgcn_readfirstlane(3 * PA + q * PB + rb * 1024);
            }
        }
        const unsigned short* abase = p.A + (long long)m0 * lda + (long long)kt0 * BK;
        const unsigned short* bbase = p.B + (long long)n0 * ldb + (long long)kt0 * BK;
        auto issue = [&](int stage, unsigned inv) {
            unsigned char* sb = smem + stage * STAGE;
            const __amdgpu_buffer_rsrc_t ra = pg_rsrc(abase), rb = pg_rsrc(bbase);
#pragma unroll
            for (int i = 0; i < NIA; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(sb + alds[i]), 16, (int)(avo[i] | inv), 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NIB; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)(sb + blds[i]), 16, (int)(bvo[i] | inv), 0, 0, 0);
            abase += BK;
            bbase += BK;
        };
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        pg_u32x4 fa0[3][TM], fb0[3][TN], fa1[3][TM], fb1[3][TN];
        auto read_frags = [&](int stage, int fo, pg_u32x4 (&fa)[3][TM], pg_u32x4 (&fb)[3][TN]) {
            const unsigned char* as = smem + stage * STAGE + (wm * WM) * 64 + fo;
            const unsigned char* bs = smem + stage * STAGE + 3 * PA + (wn * WN) * 64 + fo;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[q][i] = *reinterpret_cast<const pg_u32x4*>(as + q * PA + i * 32 * 64);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[q][j] = *reinterpret_cast<const pg_u32x4*>(bs + q * PB + j * 32 * 64);
            }
        };
        auto mma = [&](const pg_u32x4 (&fa)[3][TM], const pg_u32x4 (&fb)[3][TN]) {
            constexpr int qa[6] = {2, 0, 1, 1, 0, 0}, qb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pg_bf16x8, fa[qa[t]][i]),
                                                                            __builtin_bit_cast(pg_bf16x8, fb[qb[t]][j]), acc[i][j], 0, 0, 0);
        };
        constexpr int NIW = NIA + NIB;
        constexpr int NMMA = 6 * TM * TN, NRD = 3 * (TM + TN);
        issue(0, 0u);
        issue(1, ntiles > 1 ? 0u : PG_INVALID);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIW) : "memory");
        __builtin_amdgcn_s_barrier();
        auto settle0 = [&]() {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
#pragma unroll
                for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(fa0[q][i]));
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(fb0[q][j]));
            }
        };
        read_frags(0, fo0, fa0, fb0);
        settle0();
        int cur = 0;
        for (int t = 0; t < ntiles; ++t) {
            const int nxt = (cur + 1 == NST) ? 0 : cur + 1;
            const int nn = (nxt + 1 == NST) ? 0 : nxt + 1;
            __builtin_amdgcn_sched_barrier(0);
            issue(nn, t + NST - 1 < ntiles ? 0u : PG_INVALID);
            read_frags(cur, fo1, fa1, fb1);
            mma(fa0, fb0);
#pragma unroll
            for (int g = 0; g < NMMA; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (g < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (g < NIW) { __builtin_amdgcn_sched_group_barrier(0x004, 2, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NIW) : "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            read_frags(nxt, fo0, fa0, fb0);
            mma(fa1, fb1);
#pragma unroll
            for (int g = 0; g < NMMA; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (g < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            settle0();
            cur = nxt;
        }
        __syncthreads();                               // (drains the DMA queue: the output tile is staged over the operand stages)

        const float alpha = p.alpha;
        float* const Ct = reinterpret_cast<float*>(smem);
        int* const flag = reinterpret_cast<int*>(smem + CBYTES);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Ct[(wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * CTS + wn * WN + j * 32 + lr] = acc[i][j][r] * alpha;
        __syncthreads();
        constexpr int QN = BN / 4;
        // contributors of this tail tile: the tail blocks whose unit range meets the tile's (consecutive block indices)
        unsigned c_first = 0, c_last = 0, first_slot = 0;
        if (partial) {
            const unsigned tt = rem - full;
            const long long t0 = (long long)tt * nkt, t1 = t0 + nkt;
            // the block that holds unit u is ceil((u + 1) nsk / units) - 1
            c_first = (unsigned)(((t0 + 1) * nsk + units - 1) / units - 1);
            c_last = (unsigned)((t1 * nsk + units - 1) / units - 1);
            first_slot = (ub(c_first) == t0) ? 0u : 1u;          // only the first contributor can have started in the previous tile
            // slab of block b for this tile: slot 0 if the tile holds b's first unit, else slot 1
            float* slab = p.sk_ws + ((size_t)skb * 2 + (sg ? 1 : 0)) * (size_t)(BM * BN);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slab, 0, BM * BN * 4, 0x00020000);
#pragma unroll
            for (int q = 0; q < BM * QN / NT; ++q) {
                const int idx = tid + q * NT;
                const int row = idx / QN, c = (idx % QN) * 4;
                const pg_u32x4 v = *reinterpret_cast<const pg_u32x4*>(&Ct[row * CTS + c]);
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, (row * BN + c) * 4, 0, 16 /* sc1: write-through */);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) *flag = __hip_atomic_fetch_add(p.sk_cnt + tt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const bool last = *flag == (int)(c_last - c_first);
            if (!last) { __syncthreads(); continue; }                 // (uniform) somebody else finishes this tile
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __hip_atomic_store(p.sk_cnt + tt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
            }
            __syncthreads();
        }
        const float* bias = p.bias;
        const int epi = p.epi, accumulate = p.accumulate;
        const long long ldc = p.ldc;
        float* const C = p.C;
        float* const C2 = p.C2;
        unsigned short* const Cp = p.Cp;
#pragma unroll
        for (int q = 0; q < BM * QN / NT; ++q) {
            const int idx = tid + q * NT;
            const int row = idx / QN, c = (idx % QN) * 4;
            const int gm = m0 + row, gn = n0 + c;
            if (gm >= M || gn >= N) continue;
            float4 v;
            if (partial) {
                v = make_float4(0.f, 0.f, 0.f, 0.f);
                for (unsigned b = c_first; b <= c_last; ++b) {
                    const unsigned slot = (b == c_first) ? first_slot : 0u;
                    const float4 o = *reinterpret_cast<const float4*>(p.sk_ws + ((size_t)b * 2 + slot) * (size_t)(BM * BN) + row * BN + c);
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                }
            } else {
                v = *reinterpret_cast<const float4*>(&Ct[row * CTS + c]);
            }
            if (bias) {
                const float4 b = *reinterpret_cast<const float4*>(bias + gn);
                v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
            }
            float* cp = C + (long long)gm * ldc + gn;
            if (accumulate) {
                const float4 o = *reinterpret_cast<const float4*>(cp);
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            if (epi == VBG_EPI_RELU) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
            if (C) *reinterpret_cast<float4*>(cp) = v;
            if (epi == VBG_EPI_GELU_DUAL) {
                v = make_float4(gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w));
                if (C2) *reinterpret_cast<float4*>(C2 + (long long)gm * ldc + gn) = v;
            }
            if (Cp) {
                const float e[4] = {v.x, v.y, v.z, v.w};
                unsigned short h[4], m[4], l[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const unsigned u = __float_as_uint(e[t]);
                    const float r1 = e[t] - __uint_as_float(u & 0xffff0000u);
                    const unsigned u1 = __float_as_uint(r1);
                    const float r2 = r1 - __uint_as_float(u1 & 0xffff0000u);
                    h[t] = (unsigned short)(u >> 16); m[t] = (unsigned short)(u1 >> 16); l[t] = (unsigned short)(__float_as_uint(r2) >> 16);
                }
                unsigned short* o = Cp + (long long)gm * p.ldp + gn;
                *reinterpret_cast<uint2*>(o) = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
                *reinterpret_cast<uint2*>(o + p.c_plane) = make_uint2(m[0] | ((unsigned)m[1] << 16), m[2] | ((unsigned)m[3] << 16));
                *reinterpret_cast<uint2*>(o + 2 * p.c_plane) = make_uint2(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16));
            }
        }
        __syncthreads();                               // the next segment's DMA overwrites the staged tile
    }
}

// exact three-way split of one float: top 16 bits, top 16 bits of the (exact) remainder, the (exact, <= 8 bit) rest
__device__ __forceinline__ void pg_split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
    const unsigned u = __float_as_uint(x);
    const float r1 = x - __uint_as_float(u & 0xffff0000u);
    const unsigned u1 = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(u1 & 0xffff0000u);
    h = (unsigned short)(u >> 16); m = (unsigned short)(u1 >> 16); l = (unsigned short)(__float_as_uint(r2) >> 16);
}

// x [rows][cols] fp32 (row stride ldx) -> planes [3][rows][ldp] bf16, columns cols..ldp-1 zero.  One thread = 8 columns.
// colsum != nullptr: colsum[c] += sum_r x[r][c] rides along (the bias gradient of a linear layer is the column sum of the same dy
// whose planes feed its data- and weight-gradient products).  The host then launches a thread count that is a multiple of the
// 8-column chunks per row, so a thread keeps ONE chunk for all its rows: 8 register partials, one LDS reduction per block, one
// global atomic per column and block (<= 288 blocks: same-address atomics serialise at ~0.1 us each).
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, long long ldx, int rows, int cols,
                                                            unsigned short* __restrict__ out, int ldp, long long plane, int relu,
                                                            float* colsum) {
    extern __shared__ float sh_cols[];
    const int cpr = ldp / 8;
    const long long n = (long long)rows * cpr;
    const long long stride = (long long)gridDim.x * blockDim.x;
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long long i = i0; i < n; i += stride) {
        const int r = (int)(i / cpr), c = (int)(i % cpr) * 8;
        float e[8];
        if (c + 8 <= cols && ((ldx & 3) == 0) && ((((uintptr_t)x) & 15) == 0)) {
            const float4 a = *reinterpret_cast<const float4*>(x + (long long)r * ldx + c);
            const float4 b = *reinterpret_cast<const float4*>(x + (long long)r * ldx + c + 4);
            e[0] = a.x; e[1] = a.y; e[2] = a.z; e[3] = a.w; e[4] = b.x; e[5] = b.y; e[6] = b.z; e[7] = b.w;
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) e[t] = (c + t < cols) ? x[(long long)r * ldx + c + t] : 0.f;
        }
        unsigned short h[8], m[8], l[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            cs[t] += e[t];
            pg_split3(relu ? fmaxf(e[t], 0.f) : e[t], h[t], m[t], l[t]);
        }
        unsigned short* o = out + (long long)r * ldp + c;
        auto pack = [](const unsigned short* s) {
            return make_uint4(s[0] | ((unsigned)s[1] << 16), s[2] | ((unsigned)s[3] << 16), s[4] | ((unsigned)s[5] << 16), s[6] | ((unsigned)s[7] << 16));
        };
        *reinterpret_cast<uint4*>(o) = pack(h);
        *reinterpret_cast<uint4*>(o + plane) = pack(m);
        *reinterpret_cast<uint4*>(o + 2 * plane) = pack(l);
    }
    if (colsum) {                                        // (uniform)
        for (int c = threadIdx.x; c < ldp; c += blockDim.x) sh_cols[c] = 0.f;
        __syncthreads();
        if (i0 < n) {
            const int c = (int)(i0 % cpr) * 8;           // the chunk this thread kept (stride % cpr == 0)
#pragma unroll
            for (int t = 0; t < 8; ++t) atomicAdd(&sh_cols[c + t], cs[t]);
        }
        __syncthreads();
        for (int c = threadIdx.x; c < cols; c += blockDim.x) {
            const float v = sh_cols[c];
            if (v != 0.f) unsafeAtomicAdd(colsum + c, v);
        }
    }
}

// x [rows][cols] fp32 (row stride ldx) -> fp16-pair planes [2][rows][ldp], columns cols..ldp-1 zero.  One thread = 8 columns.
// amax != nullptr: x is multiplied by the power of two that brings the slot's value (the tensor's largest magnitude) to [2^13, 2^14) --
// a gradient operand; the consumer scales its product back (vbg_plane_gemm_desc.a_amax).  colsum as in split_planes_kernel (of the
// UNSCALED values).
__global__ __launch_bounds__(256) void split_planes_pair_kernel(const float* __restrict__ x, long long ldx, int rows, int cols,
                                                                 unsigned short* __restrict__ out, int ldp, long long plane,
                                                                 const unsigned* amax, float* colsum) {
    extern __shared__ float sh_cols[];
    const float sc = amax ? vbg_pow2_scale(vbg_amax_read(amax)).x : 1.f;
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cpr = ldp / 8;
    const long long n = (long long)rows * cpr;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const bool vec = ((ldx & 3) == 0) && ((((uintptr_t)x) & 15) == 0);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int r = (int)(i / cpr), c = (int)(i % cpr) * 8;
        float e[8];
        if (c + 8 <= cols && vec) {
            const float4 a = *reinterpret_cast<const float4*>(x + (long long)r * ldx + c);
            const float4 b = *reinterpret_cast<const float4*>(x + (long long)r * ldx + c + 4);
            e[0] = a.x; e[1] = a.y; e[2] = a.z; e[3] = a.w; e[4] = b.x; e[5] = b.y; e[6] = b.z; e[7] = b.w;
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) e[t] = (c + t < cols) ? x[(long long)r * ldx + c + t] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) { cs[t] += e[t]; e[t] *= sc; }
        uint4 h, l;
        pg_split2(e[0], e[1], h.x, l.x); pg_split2(e[2], e[3], h.y, l.y); pg_split2(e[4], e[5], h.z, l.z); pg_split2(e[6], e[7], h.w, l.w);
        unsigned short* o = out + (long long)r * ldp + c;
        *reinterpret_cast<uint4*>(o) = h;
        *reinterpret_cast<uint4*>(o + plane) = l;
    }
    if (colsum) {                                        // (uniform; the host launched a thread count that keeps a thread on ONE chunk)
        for (int c = threadIdx.x; c < ldp; c += blockDim.x) sh_cols[c] = 0.f;
        __syncthreads();
        if (i0 < n) {
            const int c = (int)(i0 % cpr) * 8;
#pragma unroll
            for (int t = 0; t < 8; ++t) atomicAdd(&sh_cols[c + t], cs[t]);
        }
        __syncthreads();
        for (int c = threadIdx.x; c < cols; c += blockDim.x) {
            const float v = sh_cols[c];
            if (v != 0.f) unsafeAtomicAdd(colsum + c, v);
        }
    }
}

// x [rows][cols] fp32 -> TRANSPOSED planes [3][cols][ldp] bf16 with ldp >= rows (multiple of 32), entries rows..ldp-1 zero:
// out[q][c][r] = piece_q(x[r][c]).  64 x 64 tiles through LDS.
// out[matrix] = max over columns of sum over rows |w[r][c]| (bit pattern, atomicMax: floats >= 0 order like their bits).  A block =
// 64 columns (16 lanes x float4) x 16 row groups; the row groups meet in LDS.  (First version: one thread per column walking all rows,
// 144 blocks -> 278 us per launch; this one reads the 113 MB of the twelve bert-base matrices at HBM speed.)
__global__ __launch_bounds__(256) void col_l1_max_kernel(const vbg_l1_entry* __restrict__ tab, unsigned* __restrict__ out) {
    const vbg_l1_entry e = tab[blockIdx.y];
    const int c0 = blockIdx.x * 64;
    if (c0 >= e.cols) return;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c = c0 + tx * 4;
    const bool vec = (e.ld & 3) == 0 && ((((uintptr_t)e.w) & 15) == 0) && c + 4 <= e.cols;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int r = ty; r < e.rows; r += 16) {
        const float* p = e.w + (long long)r * e.ld + c;
        if (vec) {
            const float4 v = *reinterpret_cast<const float4*>(p);
            s.x += fabsf(v.x); s.y += fabsf(v.y); s.z += fabsf(v.z); s.w += fabsf(v.w);
        } else {
            if (c < e.cols) s.x += fabsf(p[0]);
            if (c + 1 < e.cols) s.y += fabsf(p[1]);
            if (c + 2 < e.cols) s.z += fabsf(p[2]);
            if (c + 3 < e.cols) s.w += fabsf(p[3]);
        }
    }
    __shared__ float4 red[16][16];
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0) {
        float4 t = red[0][tx];
#pragma unroll
        for (int g = 1; g < 16; ++g) { const float4 v = red[g][tx]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        float m = fmaxf(fmaxf(t.x, t.y), fmaxf(t.z, t.w));
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (tx == 0) atomicMax(out + blockIdx.y, __float_as_uint(m));
    }
}

__global__ __launch_bounds__(256) void split_planes_t_kernel(const float* __restrict__ x, long long ldx, int rows, int cols,
                                                              unsigned short* __restrict__ out, int ldp, long long plane) {
    __shared__ float t[64][65];
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tid = threadIdx.x;
    const bool vec = ((ldx & 3) == 0) && ((((uintptr_t)x) & 15) == 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + i * 256, r = f / 16, c = (f % 16) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r0 + r < rows) {
            const float* src = x + (long long)(r0 + r) * ldx + c0 + c;
            if (vec && c0 + c + 4 <= cols) v = *reinterpret_cast<const float4*>(src);
            else {
                if (c0 + c + 0 < cols) v.x = src[0];
                if (c0 + c + 1 < cols) v.y = src[1];
                if (c0 + c + 2 < cols) v.z = src[2];
                if (c0 + c + 3 < cols) v.w = src[3];
            }
        }
        t[r][c] = v.x; t[r][c + 1] = v.y; t[r][c + 2] = v.z; t[r][c + 3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f = tid + i * 256, c = f / 8, rg = (f % 8) * 8;
        if (c0 + c >= cols || r0 + rg >= ldp) continue;
        unsigned short h[8], m[8], l[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) pg_split3(t[rg + e][c], h[e], m[e], l[e]);
        unsigned short* o = out + (long long)(c0 + c) * ldp + r0 + rg;
        auto pack = [](const unsigned short* s) {
            return make_uint4(s[0] | ((unsigned)s[1] << 16), s[2] | ((unsigned)s[3] << 16), s[4] | ((unsigned)s[5] << 16), s[6] | ((unsigned)s[7] << 16));
        };
        *reinterpret_cast<uint4*>(o) = pack(h);
        *reinterpret_cast<uint4*>(o + plane) = pack(m);
        *reinterpret_cast<uint4*>(o + 2 * plane) = pack(l);
    }
}

// batched transposing split: job j = matrix [rows_j][cols_j] fp32 at src + soff_j -> transposed planes [3][cols_j][ldp_j] at
// dst + doff_j (plane stride `plane` elements for every job: one flat plane buffer), tiles of 64 x 64; tbl = int64 [njobs][6]:
// soff, rows, cols, doff, ldp, first tile.  The weights of a whole flat parameter buffer in ONE launch (their W^T planes are the
// B operands of the data-gradient products), refreshed once per optimizer step.
template <bool PAIR>
__global__ __launch_bounds__(256) void split_planes_t_batched_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst,
                                                                      const long long* __restrict__ tbl, int njobs, long long plane) {
    __shared__ float t[64][65];
    int j = 0;
    while (j + 1 < njobs && (long long)blockIdx.x >= tbl[(j + 1) * 6 + 5]) ++j;
    const long long* e = tbl + j * 6;
    const float* x = src + e[0];
    const int rows = (int)e[1], cols = (int)e[2], ldp = (int)e[4];
    unsigned short* out = dst + e[3];
    const int local = (int)(blockIdx.x - e[5]);
    const int tr = (ldp + 63) / 64;
    const int r0 = (local % tr) * 64, c0 = (local / tr) * 64;
    const int tid = threadIdx.x;
    const long long ldx = cols;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + i * 256, r = f / 16, c = (f % 16) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (r0 + r < rows) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (c0 + c + q < cols) v[q] = x[(long long)(r0 + r) * ldx + c0 + c + q];
        }
        t[r][c] = v[0]; t[r][c + 1] = v[1]; t[r][c + 2] = v[2]; t[r][c + 3] = v[3];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f = tid + i * 256, c = f / 8, rg = (f % 8) * 8;
        if (c0 + c >= cols || r0 + rg >= ldp) continue;
        unsigned short* o = out + (long long)(c0 + c) * ldp + r0 + rg;
        if constexpr (PAIR) {                          // fp16-pair planes [2][cols][ldp] (the W^T operands of the form-1 data gradients)
            uint4 h, l;
            pg_split2(t[rg + 0][c], t[rg + 1][c], h.x, l.x); pg_split2(t[rg + 2][c], t[rg + 3][c], h.y, l.y);
            pg_split2(t[rg + 4][c], t[rg + 5][c], h.z, l.z); pg_split2(t[rg + 6][c], t[rg + 7][c], h.w, l.w);
            *reinterpret_cast<uint4*>(o) = h;
            *reinterpret_cast<uint4*>(o + plane) = l;
            continue;
        }
        unsigned short h[8], m[8], l[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) pg_split3(t[rg + q][c], h[q], m[q], l[q]);
        auto pack = [](const unsigned short* s) {
            return make_uint4(s[0] | ((unsigned)s[1] << 16), s[2] | ((unsigned)s[3] << 16), s[4] | ((unsigned)s[5] << 16), s[6] | ((unsigned)s[7] << 16));
        };
        *reinterpret_cast<uint4*>(o) = pack(h);
        *reinterpret_cast<uint4*>(o + plane) = pack(m);
        *reinterpret_cast<uint4*>(o + 2 * plane) = pack(l);
    }
}

// VBG_PINGPONG=0: the lockstep k-loop of the 8-wave NT tiles (A/B measurements; results are bit-identical)
static bool pg_pingpong() {
    static const bool on = [] { const char* e = getenv("VBG_PINGPONG"); return !(e && e[0] == '0'); }();
    return on;
}

template <int BM, int BN, int WGM, int WGN, int NST, bool TRANS = false, int FORM = 1>
static void pg_launch_pair(const vbg_plane_gemm_desc& d, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    dim3 g(cdiv(d.M, BM), cdiv(d.N, BN), 1);
    if (d.ngroups > 0) {
        int total = 0;
        for (int i = 0; i < d.ngroups; ++i) total += d.grp[i].tiles_m * d.grp[i].tiles_n;
        g = dim3(total, 1, 1);
    }
    (void)hipGetLastError();
    if (e0 && e1) hipExtLaunchKernelGGL((plane_gemm_kernel<BM, BN, WGM, WGN, NST, TRANS, true, FORM>), g, dim3(WGM * WGN * 64), 0, s, e0, e1, 0, d);
    else hipLaunchKernelGGL((plane_gemm_kernel<BM, BN, WGM, WGN, NST, TRANS, true, FORM>), g, dim3(WGM * WGN * 64), 0, s, d);
}

template <int BM, int BN, int WGM, int WGN, int NST, bool TRANS = false>
static void pg_launch(const vbg_plane_gemm_desc& d, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    dim3 g(cdiv(d.M, BM), cdiv(d.N, BN), d.splitk);
    if (d.ngroups > 0) {
        int total = 0;
        for (int i = 0; i < d.ngroups; ++i) total += d.grp[i].tiles_m * d.grp[i].tiles_n;
        g = dim3(total, 1, d.splitk);
    }
    (void)hipGetLastError();
    if constexpr (!TRANS && WGM * WGN == 8) {
        if (!pg_pingpong()) {
            if (e0 && e1) hipExtLaunchKernelGGL((plane_gemm_kernel<BM, BN, WGM, WGN, NST, TRANS, false>), g, dim3(WGM * WGN * 64), 0, s, e0, e1, 0, d);
            else hipLaunchKernelGGL((plane_gemm_kernel<BM, BN, WGM, WGN, NST, TRANS, false>), g, dim3(WGM * WGN * 64), 0, s, d);
            return;
        }
    }
    if (e0 && e1) hipExtLaunchKernelGGL((plane_gemm_kernel<BM, BN, WGM, WGN, NST, TRANS>), g, dim3(WGM * WGN * 64), 0, s, e0, e1, 0, d);
    else hipLaunchKernelGGL((plane_gemm_kernel<BM, BN, WGM, WGN, NST, TRANS>), g, dim3(WGM * WGN * 64), 0, s, d);
}

}  // namespace vbg

using namespace vbg;

static int plane_gemm_dispatch(const vbg_plane_gemm_desc* desc, void* stream, void* e0, void* e1) {
    VBG_CHECK_ARG(desc != nullptr);
    vbg_plane_gemm_desc d = *desc;
    if (d.ngroups > 0) {          // grouped: the first entry also stands in the descriptor's own fields for the shared checks below
        VBG_CHECK_ARG(d.ngroups <= VBG_PLANE_MAX_GROUPS && d.Cp == nullptr && d.bias == nullptr && d.C2 == nullptr);
        for (int i = 0; i < d.ngroups; ++i) {
            vbg_plane_group& g = d.grp[i];
            VBG_CHECK_ARG(g.A && g.B && g.C && g.M > 0 && g.N > 0 && g.N % 4 == 0 && g.ldc % 4 == 0 && ((uintptr_t)g.C & 15) == 0);
            VBG_CHECK_ARG(((uintptr_t)g.A & 15) == 0 && ((uintptr_t)g.B & 15) == 0 && g.a_plane % 8 == 0 && g.b_plane % 8 == 0);
            if (d.trans) VBG_CHECK_ARG(g.lda % 32 == 0 && g.ldb % 32 == 0 && g.lda >= g.M && g.ldb >= g.N);
            else VBG_CHECK_ARG(g.lda % 8 == 0 && g.ldb % 8 == 0 && g.lda >= d.K && g.ldb >= d.K);
            VBG_CHECK_ARG(2 * g.a_plane * 2 + 256 * g.lda * 2 < 0x7fffffffll && 2 * g.b_plane * 2 + 256 * g.ldb * 2 < 0x7fffffffll);
            const int bn = (d.trans || d.tile != 64064) ? 128 : 64, bm = (d.trans && d.tile == 256128) ? 256 : bn;
            g.tiles_m = (g.M + bm - 1) / bm;
            g.tiles_n = (g.N + bn - 1) / bn;
        }
        d.A = d.grp[0].A; d.B = d.grp[0].B; d.C = d.grp[0].C; d.M = d.grp[0].M; d.N = d.grp[0].N;
        d.a_plane = d.grp[0].a_plane; d.b_plane = d.grp[0].b_plane; d.lda = d.grp[0].lda; d.ldb = d.grp[0].ldb; d.ldc = d.grp[0].ldc;
        if (!d.trans && d.tile != 64064) d.tile = 128129;
    }
    VBG_CHECK_ARG(d.A && d.B && (d.C || d.Cp || d.Cq));
    // bound-scaled Cq: the epilogue of the 8-wave NT tiles only (as for c_amax)
    if (d.cq_ref_in) VBG_CHECK_ARG(d.Cq && d.cq_mul > 0.f && d.tile != 256256 && !(d.sk_ws && d.sk_cnt) && !d.trans && d.ngroups == 0 && d.splitk == 1);
    VBG_CHECK_ARG(d.M >= 0 && d.N >= 0 && d.K > 0);
    if (d.trans) VBG_CHECK_ARG(d.lda % 32 == 0 && d.ldb % 32 == 0 && d.lda >= d.M && d.ldb >= d.N);
    else VBG_CHECK_ARG(d.K % 32 == 0 && d.lda % 8 == 0 && d.ldb % 8 == 0 && d.lda >= d.K && d.ldb >= d.K);
    VBG_CHECK_ARG(((uintptr_t)d.A & 15) == 0 && ((uintptr_t)d.B & 15) == 0 && (d.a_plane % 8) == 0 && (d.b_plane % 8) == 0);
    VBG_CHECK_ARG(d.N % 4 == 0 && d.ldc % 4 == 0 && ((uintptr_t)d.C & 15) == 0 && ((uintptr_t)d.C2 & 15) == 0 && ((uintptr_t)d.bias & 15) == 0);
    // 32-bit DMA offsets: the three planes of the rows a block touches must sit within 2 GiB of the block's first row
    VBG_CHECK_ARG(2 * d.a_plane * 2 + (long long)256 * d.lda * 2 < 0x7fffffffll && 2 * d.b_plane * 2 + (long long)256 * d.ldb * 2 < 0x7fffffffll);
    if (d.splitk < 1) d.splitk = 1;
    if (d.splitk > 1) VBG_CHECK_ARG(d.accumulate == 1 && d.C);
    if (d.accumulate) VBG_CHECK_ARG(d.epi == VBG_EPI_NONE && d.Cp == nullptr);
    if (d.colsum) VBG_CHECK_ARG(d.splitk == 1 && d.ngroups == 0 && !d.trans && !d.accumulate && d.epi != VBG_EPI_GELU_DUAL && d.tile != 256256 && !(d.sk_ws && d.sk_cnt));
    // the amax by-product exists in the epilogue of the 8-wave NT tiles only: elsewhere the slot would stay 0 and the consumer's scale
    // would be meaningless (gradient pieces underflowing in fp16) -- refuse instead of ignoring it
    if (d.c_amax) VBG_CHECK_ARG(d.tile != 256256 && !(d.sk_ws && d.sk_cnt) && !d.trans && d.ngroups == 0 && d.splitk == 1);
    if (d.epi == VBG_EPI_GELU_DUAL) VBG_CHECK_ARG(d.C2 != nullptr || d.Cp != nullptr || d.Cq != nullptr);
    if (d.epi == VBG_EPI_MUL_GELU_GRAD) VBG_CHECK_ARG(d.C2 != nullptr && d.ngroups == 0 && !d.trans && !(d.sk_ws && d.sk_cnt));
    if (d.Cp) VBG_CHECK_ARG(d.ldp % 8 == 0 && d.ldp >= d.N && ((uintptr_t)d.Cp & 7) == 0 && d.c_plane % 4 == 0);
    if (d.Cq) VBG_CHECK_ARG(d.ldq % 8 == 0 && d.ldq >= d.N && ((uintptr_t)d.Cq & 7) == 0 && d.q_plane % 4 == 0 && !d.accumulate && d.ngroups == 0 &&
                            !d.trans && d.tile != 256256 && !(d.sk_ws && d.sk_cnt));
    if (d.M == 0 || d.N == 0) return VBG_OK;
    hipStream_t s = (hipStream_t)stream;
    int tile = d.tile;
    if (d.form == 1 || d.form == 2) {
        // two fp16 planes per operand (csrc/gemm_planes.hip FORM 1; FORM 2 = `amp`: their hi planes only, one product): the 8-wave tiles only
        VBG_CHECK_ARG(d.splitk == 1 && !(d.sk_ws && d.sk_cnt) && (d.ngroups == 0 || d.trans));
        // two planes per operand leave room for one more LDS stage than the three-plane form has (128 x 128: 4 x 32 KB, 256 x 128:
        // 3 x 48 KB): a deeper ring hides the Infinity-Cache round trips of the row pieces an XCD touches first.  VBG_PAIR_DEEP=0: the
        // stage counts of the three-plane kernels (A/B switch)
        static const bool deep = !(getenv("VBG_PAIR_DEEP") && atoi(getenv("VBG_PAIR_DEEP")) == 0);
        hipEvent_t ev0 = (hipEvent_t)e0, ev1 = (hipEvent_t)e1;
        if (d.form == 2) {
            if (d.trans) {
                if (tile == 256128) pg_launch_pair<256, 128, 4, 2, 3, true, 2>(d, s, ev0, ev1);
                else pg_launch_pair<128, 128, 4, 2, 3, true, 2>(d, s, ev0, ev1);
            } else if (tile == 256128) pg_launch_pair<256, 128, 4, 2, 3, false, 2>(d, s, ev0, ev1);
            else if (tile == 128129 || tile == 0) pg_launch_pair<128, 128, 4, 2, 4, false, 2>(d, s, ev0, ev1);
            else if ((tile == 64064 || tile == 64004) && d.ngroups == 0 && !d.c_amax && !d.cq_ref_in && !d.colsum) {
                if (tile == 64004) pg_launch_pair<64, 64, 2, 2, 4, false, 2>(d, s, ev0, ev1);
                else pg_launch_pair<64, 64, 2, 2, 3, false, 2>(d, s, ev0, ev1);
            }
            else return VBG_EARG;
            VBG_LAUNCH_RET();
        }
        if (d.trans) {
            if (tile == 256128) {
                if (deep) pg_launch_pair<256, 128, 4, 2, 3, true>(d, s, ev0, ev1);
                else pg_launch_pair<256, 128, 4, 2, 2, true>(d, s, ev0, ev1);
            } else pg_launch_pair<128, 128, 4, 2, 3, true>(d, s, ev0, ev1);
            VBG_LAUNCH_RET();
        }
        if (tile == 256128) {
            if (deep) pg_launch_pair<256, 128, 4, 2, 3>(d, s, ev0, ev1);
            else pg_launch_pair<256, 128, 4, 2, 2>(d, s, ev0, ev1);
        } else if (tile == 128130) pg_launch_pair<128, 128, 2, 4, 3>(d, s, ev0, ev1);
        else if (tile == 128129 || tile == 0) {
            if (deep) pg_launch_pair<128, 128, 4, 2, 4>(d, s, ev0, ev1);
            else pg_launch_pair<128, 128, 4, 2, 3>(d, s, ev0, ev1);
        } else if ((tile == 64064 || tile == 64004) && d.ngroups == 0 && !d.c_amax && !d.cq_ref_in && !d.colsum) {
            // small forward products (single-document inference): no amax / bound / column-sum by-products, those live in the 8-wave epilogue.
            // 64004: four LDS stages (64 KB, two workgroups per CU): three k-tiles in flight for weights that come from HBM -- FFN2 of one document
            // 42 -> 32 us (tools/small_gemm_cold.py; eight stages and a split-K over atomics were measured too: no better / slower)
            if (tile == 64004) pg_launch_pair<64, 64, 2, 2, 4>(d, s, ev0, ev1);
            else pg_launch_pair<64, 64, 2, 2, 3>(d, s, ev0, ev1);
        } else return VBG_EARG;
        VBG_LAUNCH_RET();
    }
    VBG_CHECK_ARG(d.form == 0);
    if (d.trans) {
        if (tile == 256128) pg_launch<256, 128, 4, 2, 2, true>(d, s, (hipEvent_t)e0, (hipEvent_t)e1);
        else if (tile == 128130) pg_launch<128, 128, 2, 4, 3, true>(d, s, (hipEvent_t)e0, (hipEvent_t)e1);
        else pg_launch<128, 128, 4, 2, 3, true>(d, s, (hipEvent_t)e0, (hipEvent_t)e1);
        VBG_LAUNCH_RET();
    }
    if (tile == 0) {
        const long t128 = (long)cdiv(d.M, 128) * cdiv(d.N, 128) * d.splitk;
        tile = (t128 >= 200 && d.N >= 128) ? 128128 : 64064;
    }
    if ((tile == 128129 || tile == 128130) && d.sk_ws && d.sk_cnt && d.sk_blocks >= 8 && d.splitk == 1) {
        // stream-K tail when the last round of 128 x 128 tiles would leave a good part of the chip idle
        const int tm = cdiv(d.M, 128), tn = cdiv(d.N, 128), nkt = d.K / 32;
        const long T = (long)tm * tn, ncu = d.sk_blocks;
        const long full = T / ncu * ncu, r = T - full;
        if (r > 0 && r * 100 <= ncu * 85 && r * nkt >= 4 * ncu) {
            VBG_CHECK_ARG(((uintptr_t)d.sk_ws & 15) == 0);
            d.sk_full = (int)full; d.sk_tiles_m = tm; d.sk_tiles_n = tn;
            const dim3 g((unsigned)(full + ncu));
            (void)hipGetLastError();
            hipEvent_t ev0 = (hipEvent_t)e0, ev1 = (hipEvent_t)e1;
            if (tile == 128129) {
                if (ev0 && ev1) hipExtLaunchKernelGGL((plane_gemm_sk_kernel<128, 128, 4, 2>), g, dim3(512), 0, s, ev0, ev1, 0, d);
                else hipLaunchKernelGGL((plane_gemm_sk_kernel<128, 128, 4, 2>), g, dim3(512), 0, s, d);
            } else {
                if (ev0 && ev1) hipExtLaunchKernelGGL((plane_gemm_sk_kernel<128, 128, 2, 4>), g, dim3(512), 0, s, ev0, ev1, 0, d);
                else hipLaunchKernelGGL((plane_gemm_sk_kernel<128, 128, 2, 4>), g, dim3(512), 0, s, d);
            }
            VBG_LAUNCH_RET();
        }
    }
    if (tile == 256256) {
        VBG_CHECK_ARG(d.splitk == 1 && !d.accumulate && d.K % 16 == 0);
        const dim3 g(cdiv(d.M, 256), cdiv(d.N, 256));
        (void)hipGetLastError();
        hipEvent_t ev0 = (hipEvent_t)e0, ev1 = (hipEvent_t)e1;
        if (ev0 && ev1) hipExtLaunchKernelGGL(plane_gemm_256_kernel, g, dim3(512), 0, s, ev0, ev1, 0, d);
        else hipLaunchKernelGGL(plane_gemm_256_kernel, g, dim3(512), 0, s, d);
        VBG_LAUNCH_RET();
    }
    if (tile == 256128) pg_launch<256, 128, 4, 2, 2>(d, s, (hipEvent_t)e0, (hipEvent_t)e1);
    else if (tile == 128128) pg_launch<128, 128, 2, 2, 3>(d, s, (hipEvent_t)e0, (hipEvent_t)e1);
    else if (tile == 128129) pg_launch<128, 128, 4, 2, 3>(d, s, (hipEvent_t)e0, (hipEvent_t)e1);      // 8 waves of 32 x 64
    else if (tile == 128130) pg_launch<128, 128, 2, 4, 3>(d, s, (hipEvent_t)e0, (hipEvent_t)e1);      // 8 waves of 64 x 32
    else if (tile == 128064) pg_launch<128, 64, 2, 2, 3>(d, s, (hipEvent_t)e0, (hipEvent_t)e1);
    else if (tile == 64064) pg_launch<64, 64, 2, 2, 3>(d, s, (hipEvent_t)e0, (hipEvent_t)e1);
    else return VBG_EARG;
    VBG_LAUNCH_RET();
}

extern "C" int vbg_plane_gemm(const vbg_plane_gemm_desc* desc, void* stream) { return plane_gemm_dispatch(desc, stream, nullptr, nullptr); }

extern "C" int vbg_plane_gemm_timed(const vbg_plane_gemm_desc* desc, void* stream, void* start_event, void* stop_event) {
    VBG_CHECK_ARG(start_event && stop_event);
    return plane_gemm_dispatch(desc, stream, start_event, stop_event);
}

extern "C" int vbg_split_planes(const float* x, long long ldx, int rows, int cols, unsigned short* out, int ldp, long long plane,
                                int relu, float* colsum_accum, void* stream) {
    VBG_CHECK_ARG(rows >= 0 && cols >= 0 && ldp % 32 == 0 && ldp >= cols && plane >= (long long)rows * ldp && plane % 8 == 0);
    if (rows == 0 || cols == 0) return VBG_OK;
    VBG_CHECK_ARG(x && out && ((uintptr_t)out & 15) == 0);
    const long long n = (long long)rows * (ldp / 8);
    long long g = (n + 255) / 256;
    size_t lds = 0;
    if (colsum_accum) {
        VBG_CHECK_ARG(relu == 0 && ldp * 4 <= 64 * 1024);
        // thread count = a multiple of the chunks per row (every thread keeps one 8-column chunk), ~288 blocks at most
        const long long cpr = ldp / 8;
        long long a = cpr, b = 256;
        while (b) { const long long t = a % b; a = b; b = t; }
        const long long unit = cpr / a;                  // blocks per multiple of cpr threads
        long long k = 288 / unit;
        if (k < 1) k = 1;
        const long long need = (g + unit - 1) / unit;
        if (k > need) k = need;
        g = k * unit;
        lds = (size_t)ldp * 4;
    } else if (g > 256 * 16) {
        g = 256 * 16;
    }
    VBG_LAUNCH(split_planes_kernel, dim3((unsigned)g), dim3(256), lds, (hipStream_t)stream, x, ldx, rows, cols, out, ldp, plane, relu,
               colsum_accum);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_split_planes_pair(const float* x, long long ldx, int rows, int cols, unsigned short* out, int ldp, long long plane,
                                     const unsigned* amax, float* colsum_accum, void* stream) {
    VBG_CHECK_ARG(rows >= 0 && cols >= 0 && ldp % 32 == 0 && ldp >= cols && plane >= (long long)rows * ldp && plane % 8 == 0);
    if (rows == 0 || cols == 0) return VBG_OK;
    VBG_CHECK_ARG(x && out && ((uintptr_t)out & 15) == 0);
    const long long n = (long long)rows * (ldp / 8);
    long long g = (n + 255) / 256;
    size_t lds = 0;
    if (colsum_accum) {          // thread count = a multiple of the chunks per row (a thread keeps one 8-column chunk), <= ~288 blocks
        VBG_CHECK_ARG(ldp * 4 <= 64 * 1024);
        const long long cpr = ldp / 8;
        long long a = cpr, b = 256;
        while (b) { const long long t = a % b; a = b; b = t; }
        const long long unit = cpr / a;
        long long k = 288 / unit;
        if (k < 1) k = 1;
        const long long need = (g + unit - 1) / unit;
        if (k > need) k = need;
        g = k * unit;
        lds = (size_t)ldp * 4;
    } else if (g > 256 * 16) {
        g = 256 * 16;
    }
    VBG_LAUNCH(split_planes_pair_kernel, dim3((unsigned)g), dim3(256), lds, (hipStream_t)stream, x, ldx, rows, cols, out, ldp, plane, amax, colsum_accum);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_col_l1_max(const vbg_l1_entry* table_dev, int n, int max_cols, unsigned* out, void* stream) {
    VBG_CHECK_ARG(n >= 0 && max_cols >= 0);
    if (n == 0 || max_cols == 0) return VBG_OK;
    VBG_CHECK_ARG(table_dev && out);
    VBG_LAUNCH(col_l1_max_kernel, dim3((unsigned)((max_cols + 63) / 64), (unsigned)n), dim3(256), 0, (hipStream_t)stream, table_dev, out);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_split_planes_t(const float* x, long long ldx, int rows, int cols, unsigned short* out, int ldp, long long plane,
                                  void* stream) {
    VBG_CHECK_ARG(rows >= 0 && cols >= 0 && ldp % 32 == 0 && ldp >= rows && plane >= (long long)cols * ldp && plane % 8 == 0);
    if (rows == 0 || cols == 0) return VBG_OK;
    VBG_CHECK_ARG(x && out && ((uintptr_t)out & 15) == 0);
    VBG_LAUNCH(split_planes_t_kernel, dim3(cdiv(ldp, 64), cdiv(cols, 64)), dim3(256), 0, (hipStream_t)stream, x, ldx, rows, cols, out, ldp, plane);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_split_planes_t_batched(const float* src, unsigned short* dst, const long long* tbl_dev, int njobs, int total_tiles,
                                          long long plane, void* stream) {
    VBG_CHECK_ARG(njobs >= 0 && total_tiles >= 0);
    if (njobs == 0 || total_tiles == 0) return VBG_OK;
    VBG_CHECK_ARG(src && dst && tbl_dev && ((uintptr_t)dst & 15) == 0 && plane % 8 == 0);
    VBG_LAUNCH(split_planes_t_batched_kernel<false>, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, src, dst, tbl_dev, njobs, plane);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_split_planes_pair_t_batched(const float* src, unsigned short* dst, const long long* tbl_dev, int njobs, int total_tiles,
                                               long long plane, void* stream) {
    VBG_CHECK_ARG(njobs >= 0 && total_tiles >= 0);
    if (njobs == 0 || total_tiles == 0) return VBG_OK;
    VBG_CHECK_ARG(src && dst && tbl_dev && ((uintptr_t)dst & 15) == 0 && plane % 8 == 0);
    VBG_LAUNCH(split_planes_t_batched_kernel<true>, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, src, dst, tbl_dev, njobs, plane);
    VBG_LAUNCH_RET();
}
