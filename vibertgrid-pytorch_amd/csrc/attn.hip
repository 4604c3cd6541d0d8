// Fused multi-head self-attention on packed variable-length sequences, forward and backward, fp32-grade on the bf16 matrix cores.
//
// Replaces the `[L, L]` score / probability blocks of transformers' BertSelfAttention (model/BERTgrid_generator.py:134: softmax(Q K^T /
// sqrt(dh)) -> dropout -> P V, and its autograd): nothing of size L x L is ever written.  Q, K, V (and dO in the backward kernels)
// arrive as the three bf16 planes the producing GEMM's epilogue wrote (x = hi + mid + lo exactly, csrc/gemm_planes.hip); every
// product is the fp32 sum of the six piece products of order <= 2^-16, exactly as in the plane GEMM, and the probabilities /
// score gradients are split the same way in registers before they enter the second product.
//
// One kernel template, three modes.  A workgroup = 4 waves owns 128 rows of one (sequence, head) -- 32 per wave, held as MFMA
// B-operand fragments in registers ("stationary") -- and streams the other side of the score matrix through LDS in tiles of 32
// rows (LDS-DMA, two stages):
//   FWD : own = queries, stream = (K, V).  S^T = K Q^T (lane = query, registers = keys: the softmax statistics are lane-local),
//         online softmax, O^T += V^T Pd^T (V^T fragments by ds_read_b64_tr_b16).  Writes O and the row statistics (m, 1 / l).
//   DQ  : own = queries, stream = (K, V).  delta = rowsum(dO o O) in the prologue; P^T = exp(S^T - m) / l, dP^T = V dO^T,
//         dS^T = P^T o (dP^T - delta), dQ^T += K^T dS^T; writes the rows' own sum P dP as the delta of the DKV pass.
//   DKV : own = keys,    stream = (Q, dO). P = exp(S - m) / l (lane = key, registers = queries), dP = dO V^T,
//         dV^T += dO^T Pd, dK^T += Q^T dS.
// (Two backward kernels recompute S and dP once each instead of exchanging dS through LDS or accumulating dQ with atomics: seven
// products instead of five, deterministic, every accumulator lane-local.)
// Dropout keeps are read from a bit mask written once per layer and step by attn_mask_kernel in both orientations (bit = key of a
// 32-key block per query, and bit = query of a 32-query block per key), so the three kernels agree by construction and spend no
// VALU on random numbers.
//
// LDS image of a streamed tile: [plane][32 rows][128 B] (64 bf16 of one head), the 16-byte chunk index XOR-ed with
// f(row) = (row>>1 & 1) << 2 | (row>>2 & 1) | (row>>3 & 1) << 1 -- applied on the SOURCE address of the DMA -- which makes both the
// row-fragment ds_read_b128 (score-type products) and the transposing ds_read_b64_tr_b16 (the second product) conflict free.
#include "vbg_common.h"
#include "../../include/vbg.h"

namespace vbg {

typedef unsigned at_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 at_bf16x8 __attribute__((ext_vector_type(8)));
typedef short at_v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) at_v4s* at_lds_v4s;
typedef __attribute__((address_space(3))) void* at_lds_ptr;

constexpr int AT_PL = 4096;                 // bytes of one plane of a 32-row tile
constexpr int AT_OP = 3 * AT_PL;            // one operand tile (three planes)
constexpr int AT_STAGE = 2 * AT_OP;         // two streamed operands per stage
constexpr unsigned AT_INVALID = 0x80000000u;

__device__ __forceinline__ int at_swz(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 1) | (((row >> 3) & 1) << 1); }

// exact three-way split of 16 values held in MFMA D layout (register r = streamed row (r&3) + 8 (r>>2) + 4 (lane>>5)) into the
// B-operand fragments of the second product: k-step ks takes registers 8 ks .. 8 ks + 7 (the A operand is read with the matching
// row order), element pairs packed low half first
__device__ __forceinline__ void at_split16(const float (&x)[16], at_u32x4 (&bp)[3][2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = x[8 * ks + 2 * j], b = x[8 * ks + 2 * j + 1];
            const float ra = a - __uint_as_float(__float_as_uint(a) & 0xffff0000u), rb = b - __uint_as_float(__float_as_uint(b) & 0xffff0000u);
            const float sa = ra - __uint_as_float(__float_as_uint(ra) & 0xffff0000u), sb = rb - __uint_as_float(__float_as_uint(rb) & 0xffff0000u);
            bp[0][ks][j] = __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
            bp[1][ks][j] = __builtin_amdgcn_perm(__float_as_uint(rb), __float_as_uint(ra), 0x07060302u);
            bp[2][ks][j] = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
        }
}

#define AT_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(at_bf16x8, a), __builtin_bit_cast(at_bf16x8, b), c, 0, 0, 0)

template <int MODE, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_kernel(const vbg_attn_desc p) {
    constexpr bool FWD = MODE == VBG_ATTN_FWD, DQ = MODE == VBG_ATTN_DQ, DKV = MODE == VBG_ATTN_DKV;
    constexpr int NS = FWD ? 1 : 2;                                   // stationary operands
    // behind the two tile stages: the dropout keep words of the workgroup's own rows (128 rows x 16 tiles) and, DKV, the statistics
    // (m, 1 / l, delta) of every query of the sequence (3 x 512 floats) -- staged ONCE, so that the tile loop issues no ordinary
    // global load (hipcc waits vmcnt(0) for any such load while LDS-DMA is in flight: it drained the tile pipeline every iteration)
    constexpr int AT_MASK_OFF = 2 * AT_STAGE, AT_STAT_OFF = AT_MASK_OFF + 128 * 16 * 4, AT_SMEM = AT_STAT_OFF + (DKV ? 3 * 512 * 4 : 0);
    __shared__ __attribute__((aligned(1024))) unsigned char smem[AT_SMEM];          // (the ONE LDS object of the kernel)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int seq = p.tasks[2 * blockIdx.x], blk = p.tasks[2 * blockIdx.x + 1], head = blockIdx.y;
    const int L = p.seq_len[seq], row0 = p.seq_row0[seq];
    const int hid = p.heads * 64;
    const int nt = (L + 31) >> 5;
    const int own0 = blk * 128 + wave * 32;                            // this wave's first own row inside the sequence
    const bool active = own0 < L;                                      // (wave-uniform)
    const long long lsoff = (long long)head * p.ntok_pad + p.pad_off[seq];
    const long long mbase = p.mask_off ? p.mask_off[seq] + (long long)head * nt * 32 * nt : 0;

    // ---- operands: streamed (op 0, op 1) and stationary (st 0, st 1) ------------------------------------------------------
    const unsigned short* sbase[2];
    long long splane[2], sld[2];
    const unsigned short* tbase[2];
    long long tplane[2], tld[2];
    {
        const unsigned short* q = p.qkv + (long long)row0 * p.qkv_ld + head * 64;
        const unsigned short* d_o = (FWD ? p.qkv : p.dO) + (FWD ? 0 : (long long)row0 * p.do_ld + head * 64);
        if constexpr (DKV) {
            sbase[0] = q;           splane[0] = p.qkv_plane; sld[0] = p.qkv_ld;          // Q
            sbase[1] = d_o;         splane[1] = p.do_plane;  sld[1] = p.do_ld;           // dO
            tbase[0] = q + hid;     tplane[0] = p.qkv_plane; tld[0] = p.qkv_ld;          // K
            tbase[1] = q + 2 * hid; tplane[1] = p.qkv_plane; tld[1] = p.qkv_ld;          // V
        } else {
            sbase[0] = q + hid;     splane[0] = p.qkv_plane; sld[0] = p.qkv_ld;          // K
            sbase[1] = q + 2 * hid; splane[1] = p.qkv_plane; sld[1] = p.qkv_ld;          // V
            tbase[0] = q;           tplane[0] = p.qkv_plane; tld[0] = p.qkv_ld;          // Q
            tbase[1] = d_o;         tplane[1] = p.do_plane;  tld[1] = p.do_ld;           // dO (DQ only)
        }
    }

    // ---- LDS-DMA of a streamed tile: wave w fills rows 8 w .. 8 w + 7 of every plane of both operands (6 instructions of 1 KiB);
    //      lane l -> row 8 w + l / 8, physical chunk l % 8 <- logical chunk (l % 8) ^ f(row) -----------------------------------
    const int drow = 8 * wave + (lane >> 3);
    const int dchunk = (lane & 7) ^ at_swz(drow);
    unsigned dvo[2][3];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int q = 0; q < 3; ++q) dvo[o][q] = (unsigned)(((long long)q * splane[o] + (long long)drow * sld[o]) * 2 + dchunk * 16);
    auto issue = [&](int stage, int t) {
        const unsigned inv = (t * 32 + drow < L) ? 0u : AT_INVALID;          // rows past the sequence land as zeros
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(sbase[o]), 0, (int)0x80000000u, 0x00020000);
            const int soff = (int)((long long)t * 32 * sld[o] * 2);
#pragma unroll
            for (int q = 0; q < 3; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (at_lds_ptr)(smem + stage * AT_STAGE + o * AT_OP + q * AT_PL + wave * 1024), 16,
                                                         (int)(dvo[o][q] | inv), soff, 0, 0);
        }
    };

    // ---- stationary fragments (B operands of the score-type products): own row lr, k-step ks = 16 B at column 16 ks + 8 lh ----
    at_u32x4 st[NS][3][4];
    {
        const bool ok = own0 + lr < L;
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    at_u32x4 v = {0u, 0u, 0u, 0u};
                    if (ok) v = *reinterpret_cast<const at_u32x4*>(tbase[s] + (long long)q * tplane[s] + (long long)(own0 + lr) * tld[s] + 16 * ks + 8 * lh);
                    st[s][q][ks] = v;
                }
    }

    // ---- LDS fragment addresses ---------------------------------------------------------------------------------------------
    int fra[4];                                  // row fragment (ds_read_b128) of k-step ks: row lr, logical chunk 2 ks + lh
    {
        const int sw = at_swz(lr);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fra[ks] = lr * 128 + (((2 * ks + lh) ^ sw) << 4);
    }
    int tra[2][2];                               // transposing read: [second half of the 8 rows][32-column block]
    {
        const int i = lane & 15, gs = (lane >> 4) & 1, rb = 4 * lh + (i >> 2);
#pragma unroll
        for (int e4 = 0; e4 < 2; ++e4)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const int row = 8 * e4 + rb, c16 = db * 4 + gs * 2 + ((i & 3) >> 1);
                tra[e4][db] = row * 128 + ((c16 ^ at_swz(row)) << 4) + (i & 1) * 8;
            }
    }
    constexpr int qa[6] = {2, 0, 1, 1, 0, 0}, qb[6] = {0, 2, 1, 0, 1, 0};       // piece products, smallest first
    // score-type product: D[streamed row][own row] += X_stream[row][:] . X_own[row][:]
    auto sprod = [&](const unsigned char* img, const at_u32x4 (&sb)[3][4], f32x16& acc) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            at_u32x4 fa[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) fa[q] = *reinterpret_cast<const at_u32x4*>(img + q * AT_PL + fra[ks]);
#pragma unroll
            for (int t = 0; t < 6; ++t) acc = AT_MFMA(fa[qa[t]], sb[qb[t]][ks], acc);
        }
    };
    // second product: D[column d of the streamed operand][own row] += sum over streamed rows X_stream[row][d] E[row][own row]
    auto tprod = [&](const unsigned char* img, const at_u32x4 (&bp)[3][2], f32x16 (&acc)[2]) {
        typedef __attribute__((address_space(3))) unsigned char* lds_bytes;
        lds_bytes im = (lds_bytes)img;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                at_u32x4 fa[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const at_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((at_lds_v4s)(im + q * AT_PL + ks * 2048 + tra[0][db]));
                    const at_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((at_lds_v4s)(im + q * AT_PL + ks * 2048 + tra[1][db]));
                    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                    fa[q] = at_u32x4{l2.x, l2.y, h2.x, h2.y};
                }
#pragma unroll
                for (int t = 0; t < 6; ++t) acc[db] = AT_MFMA(fa[qa[t]], bp[qb[t]][ks], acc[db]);
            }
    };

    // low-precision companion of tprod (first plane of both operands only): the forward's Kbar^T += K^T P^T
    auto kprod = [&](const unsigned char* img, const at_u32x4 (&bh)[2], f32x16 (&acc)[2]) {
        typedef __attribute__((address_space(3))) unsigned char* lds_bytes;
        lds_bytes im = (lds_bytes)img;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const at_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((at_lds_v4s)(im + ks * 2048 + tra[0][db]));
                const at_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((at_lds_v4s)(im + ks * 2048 + tra[1][db]));
                const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                const at_u32x4 fa = at_u32x4{l2.x, l2.y, h2.x, h2.y};
                acc[db] = AT_MFMA(fa, bh[ks], acc[db]);
            }
    };

    // The same two products with element-wise work woven in by hand: `work(pi)` (pi = 0..7: the register pair 2 pi, 2 pi + 1 of the
    // D layout) is issued behind every third MFMA, fenced so that the compiler keeps the order -- a 32 x 32 x 16 MFMA holds the matrix
    // pipe for 32 clocks, ~25 VALU instructions of a pair fit behind three of them.  (hipcc's own schedule puts the ~200 VALU
    // instructions of a tile in one block between the products; with one wave per SIMD nothing then overlaps the matrix pipe.)
    // Fragments are read one group ahead.
    auto sprod_woven = [&](const unsigned char* img, const at_u32x4 (&sb)[3][4], f32x16& acc, auto&& work) {
        at_u32x4 fa[2][3];
#pragma unroll
        for (int q = 0; q < 3; ++q) fa[0][q] = *reinterpret_cast<const at_u32x4*>(img + q * AT_PL + fra[0]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks + 1 < 4) {
#pragma unroll
                for (int q = 0; q < 3; ++q) fa[(ks + 1) & 1][q] = *reinterpret_cast<const at_u32x4*>(img + q * AT_PL + fra[ks + 1]);
            }
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
                for (int t = 3 * hf; t < 3 * hf + 3; ++t) acc = AT_MFMA(fa[ks & 1][qa[t]], sb[qb[t]][ks], acc);
                __builtin_amdgcn_sched_barrier(0);
                work(2 * ks + hf);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto tprod_woven = [&](const unsigned char* img, const at_u32x4 (&bp)[3][2], f32x16 (&acc)[2], auto&& work) {
        typedef __attribute__((address_space(3))) unsigned char* lds_bytes;
        lds_bytes im = (lds_bytes)img;
        at_u32x4 fa[2][3];
        auto rd = [&](int g, at_u32x4 (&f)[3]) {                       // group g = 2 ks + db
            const int ks = g >> 1, db = g & 1;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const at_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((at_lds_v4s)(im + q * AT_PL + ks * 2048 + tra[0][db]));
                const at_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((at_lds_v4s)(im + q * AT_PL + ks * 2048 + tra[1][db]));
                const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                f[q] = at_u32x4{l2.x, l2.y, h2.x, h2.y};
            }
        };
        rd(0, fa[0]);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g + 1 < 4) rd(g + 1, fa[(g + 1) & 1]);
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
                for (int t = 3 * hf; t < 3 * hf + 3; ++t) acc[g & 1] = AT_MFMA(fa[g & 1][qa[t]], bp[qb[t]][g >> 1], acc[g & 1]);
                __builtin_amdgcn_sched_barrier(0);
                work(2 * g + hf);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // exact three-way split of the register pair (2 pi, 2 pi + 1) into its slot of the B-operand fragments (as at_split16)
    auto split_pair = [&](float a, float b, int pi, at_u32x4 (&bp)[3][2]) {
        const float ra = a - __uint_as_float(__float_as_uint(a) & 0xffff0000u), rb = b - __uint_as_float(__float_as_uint(b) & 0xffff0000u);
        const float sa = ra - __uint_as_float(__float_as_uint(ra) & 0xffff0000u), sb2 = rb - __uint_as_float(__float_as_uint(rb) & 0xffff0000u);
        bp[0][pi >> 2][pi & 3] = __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
        bp[1][pi >> 2][pi & 3] = __builtin_amdgcn_perm(__float_as_uint(rb), __float_as_uint(ra), 0x07060302u);
        bp[2][pi >> 2][pi & 3] = __builtin_amdgcn_perm(__float_as_uint(sb2), __float_as_uint(sa), 0x07060302u);
    };

    f32x16 acc0[2], acc1[2];                     // FWD: O^T, Kbar^T; DQ: dQ^T; DKV: dK^T (acc0), dV^T (acc1)
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[db][r] = 0.f; acc1[db][r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;        // FWD: running maximum / this half-wave's partial row sum
    // softmax statistics of a query row: its maximum m and 1 / sum exp(x - m), exactly the two numbers the forward normalised with
    // (a single log-sum-exp rounds to ulp(lse) ~ 5e-7 ABSOLUTE, i.e. a common relative error of the whole recomputed row that
    // the cancellation in dP - delta amplifies)
    const long long lsplane = (long long)p.heads * p.ntok_pad;
    float m_own = 0.f, il_own = 0.f, del_own = 0.f;          // DQ: per own query
    if constexpr (DQ) {
        if (own0 + lr < L) {
            m_own = p.lse[lsoff + own0 + lr];
            il_own = p.lse[lsplane + lsoff + own0 + lr];
            // delta = sum_d dO[q][d] O[q][d] of this head: dO is rebuilt exactly from its three planes (the stationary fragments: this
            // lane holds columns 16 ks + 8 lh .. + 7 of its row), O is read as fp32; the two half-waves of a row add up below
            const float* orow = p.o + (long long)(row0 + own0 + lr) * p.ldk + head * 64 + 8 * lh;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const float4 o0 = *reinterpret_cast<const float4*>(orow + 16 * ks), o1 = *reinterpret_cast<const float4*>(orow + 16 * ks + 4);
                const float ov[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned h = st[1][0][ks][j], m = st[1][1][ks][j], l = st[1][2][ks][j];
                    const float lo_e = (__uint_as_float(h << 16) + __uint_as_float(m << 16)) + __uint_as_float(l << 16);
                    const float hi_e = (__uint_as_float(h & 0xffff0000u) + __uint_as_float(m & 0xffff0000u)) + __uint_as_float(l & 0xffff0000u);
                    del_own = fmaf(lo_e, ov[2 * j], del_own);
                    del_own = fmaf(hi_e, ov[2 * j + 1], del_own);
                }
            }
        }
        del_own += __shfl_xor(del_own, 32, 64);
    }
    const float scale = p.scale, keep_scale = p.keep_scale;
    const bool want_kbar = FWD && p.kbar != nullptr;
    float dsum = 0.f;                            // DQ: this half-wave's part of sum_k P_k dP_k of the own query
    // dropout keep words of the own row: FWD / DQ bit = key of the streamed 32-key block, DKV bit = query of the 32-query block
    unsigned* const mlds = reinterpret_cast<unsigned*>(smem + AT_MASK_OFF) + (wave * 32 + lr) * 16;
    if constexpr (DROP) {
        if (active && lh == 0 && own0 + lr < nt * 32) {
            const unsigned* mrow = (DKV ? p.mask_k : p.mask_q) + mbase + (long long)(own0 + lr) * nt;
            for (int t = 0; t < nt; ++t) mlds[t] = mrow[t];
        }
    }
    float* const slds = reinterpret_cast<float*>(smem + AT_STAT_OFF);
    if constexpr (DKV) {
        for (int i = tid; i < nt * 32; i += 256) {
            slds[i] = p.lse[lsoff + i];
            slds[512 + i] = p.lse[lsplane + lsoff + i];
            slds[1024 + i] = p.delta[lsoff + i];
        }
    }
    unsigned mword = 0xffffffffu;
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    for (int t = 0; t < nt; ++t) {
        const int stage = t & 1;
        if (t + 1 < nt) issue(stage ^ 1, t + 1);
        if constexpr (DROP) mword = mlds[t];                               // (rows past the padded length are never stored)
        // (DKV: the statistics of the streamed queries are read from LDS where they are used -- 48 registers less)
        const float* const stile = slds + t * 32 + 4 * lh;
        if (active) {
            const unsigned char* im0 = smem + stage * AT_STAGE;
            const unsigned char* im1 = im0 + AT_OP;
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            sprod(im0, st[0], s);
            const unsigned mw = mword >> (4 * lh);                        // bit of register r: (r & 3) + 8 (r >> 2)
            if constexpr (FWD) {
                float x[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) x[r] = s[r] * scale;
                if (t * 32 + 32 > L) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh >= L) x[r] = -INFINITY;
                }
                float bm = x[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) bm = fmaxf(bm, x[r]);
                bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
                const float m_new = fmaxf(m_run, bm);
                const float alpha = __expf(m_run - m_new);
                float ps = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { x[r] = __expf(x[r] - m_new); ps += x[r]; }
                l_run = l_run * alpha + ps;
                m_run = m_new;
                if (__any(alpha != 1.0f)) {                              // (exact: the maximum settles after the first tiles)
#pragma unroll
                    for (int db = 0; db < 2; ++db)
#pragma unroll
                        for (int r = 0; r < 16; ++r) { acc0[db][r] *= alpha; acc1[db][r] *= alpha; }
                }
                if (want_kbar) {
                    // Kbar = sum_k P_k K_k to bf16 precision (one piece product): the backward's first-order correction of dQ for the
                    // error of delta needs it to ~1 %, see the DQ epilogue
                    at_u32x4 bh[2];
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            bh[ks][j] = __builtin_amdgcn_perm(__float_as_uint(x[8 * ks + 2 * j + 1]), __float_as_uint(x[8 * ks + 2 * j]), 0x07060302u);
                    kprod(im0, bh, acc1);
                }
                if constexpr (DROP) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) x[r] = ((mw >> ((r & 3) + 8 * (r >> 2))) & 1u) ? x[r] * keep_scale : 0.f;
                }
                at_u32x4 bp[3][2];
                at_split16(x, bp);
                tprod(im1, bp, acc0);
            } else {
                f32x16 dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) dp[r] = 0.f;
                float pr[16], ds[16];
                if constexpr (DQ) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) pr[r] = __expf(s[r] * scale - m_own) * il_own;
                    sprod(im1, st[1], dp);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float g = dp[r];
                        if constexpr (DROP) g = ((mw >> ((r & 3) + 8 * (r >> 2))) & 1u) ? g * keep_scale : 0.f;
                        ds[r] = pr[r] * (g - del_own);
                        dsum = fmaf(pr[r], g, dsum);
                    }
                    at_u32x4 bp[3][2];
                    at_split16(ds, bp);
                    tprod(im0, bp, acc0);                                  // dQ^T += K^T dS^T
                } else {
                    // statistics of the streamed queries: register r = query 8 (r >> 2) + 4 lh + (r & 3) of the tile (loaded an
                    // iteration ahead: a use of a fresh global load in here would drain the tile DMA in flight)
                    // Phase A: P, its dropped / scaled form and the split of that only need S -> woven into the dP product.
                    float pv[16];
                    at_u32x4 bpv[3][2], bpk[3][2];
                    sprod_woven(im1, st[1], dp, [&](int pi) {
                        // registers 2 pi, 2 pi + 1 = queries 8 (pi >> 1) + 2 (pi & 1) (+ 1) of this half-wave's rows
                        const float2 mq = *reinterpret_cast<const float2*>(stile + 8 * (pi >> 1) + 2 * (pi & 1));
                        const float2 iq = *reinterpret_cast<const float2*>(stile + 512 + 8 * (pi >> 1) + 2 * (pi & 1));
                        float pd[2];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int r = 2 * pi + e;
                            pv[r] = __expf(s[r] * scale - (e ? mq.y : mq.x)) * (e ? iq.y : iq.x);
                            pd[e] = pv[r];
                            if constexpr (DROP) pd[e] = ((mw >> ((r & 3) + 8 * (r >> 2))) & 1u) ? pv[r] * keep_scale : 0.f;
                        }
                        split_pair(pd[0], pd[1], pi, bpv);
                    });
                    // Phase B: dS and its split need dP -> woven into the dV product
                    tprod_woven(im1, bpv, acc1, [&](int pi) {                                 // dV^T += dO^T Pd
                        const float2 dq = *reinterpret_cast<const float2*>(stile + 1024 + 8 * (pi >> 1) + 2 * (pi & 1));
                        float dsv[2];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int r = 2 * pi + e;
                            float g = dp[r];
                            if constexpr (DROP) g = ((mw >> ((r & 3) + 8 * (r >> 2))) & 1u) ? g * keep_scale : 0.f;
                            dsv[e] = pv[r] * (g - (e ? dq.y : dq.x));
                        }
                        split_pair(dsv[0], dsv[1], pi, bpk);
                    });
                    tprod(im0, bpk, acc0);                                 // dK^T += Q^T dS
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---- epilogue: accumulators are [d][own row]: lane = own row writes 4 consecutive d per register quad -------------------
    const bool live = active && own0 + lr < L;
    if (FWD || !p.out_amax) { if (!live) return; }             // (backward with out_amax: every lane stays for the wave's maximum)
    const long long orow = (long long)(row0 + own0 + lr) * p.ldo;
    const long long krow = (long long)(row0 + own0 + lr) * p.ldk;
    float vmax = 0.f;                                          // max |value stored into p.out| of this lane (backward, p.out_amax)
    auto store = [&](const f32x16 (&acc)[2], float mul, int col, float* dst) {
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 v = make_float4(acc[db][4 * j] * mul, acc[db][4 * j + 1] * mul, acc[db][4 * j + 2] * mul, acc[db][4 * j + 3] * mul);
                *reinterpret_cast<float4*>(dst + orow + col + db * 32 + 8 * j + 4 * lh) = v;
                if constexpr (!FWD) vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            }
    };
    if constexpr (FWD) {
        const float lt = l_run + __shfl_xor(l_run, 32, 64);
        const float il = 1.0f / lt;
        store(acc0, il, head * 64, p.out);
        if (p.out_planes) {                           // exact three-way split of the stored values, 4 bf16 (8 bytes) per plane and quad
            unsigned short* pr0 = p.out_planes + (long long)(row0 + own0 + lr) * p.op_ld + head * 64;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned h[4], m[4], l[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float e = acc0[db][4 * j + i] * il;
                        const unsigned u = __float_as_uint(e);
                        const float r1 = e - __uint_as_float(u & 0xffff0000u);
                        const unsigned u1 = __float_as_uint(r1);
                        const float r2 = r1 - __uint_as_float(u1 & 0xffff0000u);
                        h[i] = u >> 16; m[i] = u1 >> 16; l[i] = __float_as_uint(r2) >> 16;
                    }
                    unsigned short* o = pr0 + db * 32 + 8 * j + 4 * lh;
                    *reinterpret_cast<uint2*>(o) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
                    *reinterpret_cast<uint2*>(o + p.op_plane) = make_uint2(m[0] | (m[1] << 16), m[2] | (m[3] << 16));
                    *reinterpret_cast<uint2*>(o + 2 * p.op_plane) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
                }
        }
        if (p.out_pair) {                             // the same values as two fp16 pieces (hi, (x - hi) 2^11; round to nearest), 8 bytes per plane and quad
            typedef float f32x2_t __attribute__((ext_vector_type(2)));
            typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
            unsigned short* pq0 = p.out_pair + (long long)(row0 + own0 + lr) * p.oq_ld + head * 64;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x2_t v0 = {acc0[db][4 * j] * il, acc0[db][4 * j + 1] * il}, v1 = {acc0[db][4 * j + 2] * il, acc0[db][4 * j + 3] * il};
                    const f16x2_t h0 = __builtin_convertvector(v0, f16x2_t), h1 = __builtin_convertvector(v1, f16x2_t);
                    const f16x2_t l0 = __builtin_convertvector((v0 - __builtin_convertvector(h0, f32x2_t)) * 2048.f, f16x2_t);
                    const f16x2_t l1 = __builtin_convertvector((v1 - __builtin_convertvector(h1, f32x2_t)) * 2048.f, f16x2_t);
                    unsigned short* o = pq0 + db * 32 + 8 * j + 4 * lh;
                    *reinterpret_cast<uint2*>(o) = make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1));
                    *reinterpret_cast<uint2*>(o + p.oq_plane) = make_uint2(__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1));
                }
        }
        if (want_kbar) store(acc1, il, head * 64, p.kbar + krow - orow);
        if (lh == 0) { p.lse[lsoff + own0 + lr] = m_run; p.lse[lsplane + lsoff + own0 + lr] = il; }
    } else if constexpr (DQ) {
      if (live) {
        // delta = rowsum(dO o O) carries the accumulated rounding of O as an error COMMON to the whole row, which the key
        // contraction cannot average out (the unfused path sums P o dP itself).  The row's own sum_k P_k dP_k is known now; dQ is
        // linear in delta, so  dQ = scale (sum_k P_k (dP_k - delta) K_k - (delta' - delta) Kbar)  holds exactly, and the small
        // correction only needs Kbar to bf16 precision.  delta' replaces delta for the dK / dV pass that follows.
        const float dnew = dsum + __shfl_xor(dsum, 32, 64);
        const float c = dnew - del_own;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 kb = *reinterpret_cast<const float4*>(p.kbar + krow + head * 64 + db * 32 + 8 * j + 4 * lh);
                acc0[db][4 * j] -= c * kb.x; acc0[db][4 * j + 1] -= c * kb.y; acc0[db][4 * j + 2] -= c * kb.z; acc0[db][4 * j + 3] -= c * kb.w;
            }
        store(acc0, scale, head * 64, p.out);
        if (lh == 0) p.delta[lsoff + own0 + lr] = dnew;
      }
    } else {
      if (live) {
        store(acc0, scale, hid + head * 64, p.out);
        store(acc1, 1.0f, 2 * hid + head * 64, p.out);
      }
    }
    if constexpr (!FWD) {
        // the largest magnitude of d(qkv) rides on the kernels that write it (the scale of its fp16-pair planes): one atomic per
        // wave at most, on the wave's own word of the amax slot
        if (p.out_amax) {                                      // (uniform)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
            if (lane == 0 && vmax > 0.f) {
                const unsigned bits = __float_as_uint(vmax);
                unsigned* word = p.out_amax + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & (VBG_AMAX_WORDS - 1)) * VBG_AMAX_STRIDE;
                if (bits > __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(word, bits);
            }
        }
    }
}

// dropout keeps of one layer: for group g = (seq, head), query q, 32-key block kb: mask_q[off + q * nkb + kb] bit j = keep of key
// 32 kb + j; the same bits transposed: mask_k[off + key * nkb + qb] bit i = keep of query 32 qb + i.  A wave = one 32-query block x
// two 32-key blocks; draws are 16-bit slices of the counter hash (rng_u32's 64-bit state), keep <=> draw >= thr16.
// blockIdx.z = layer * ngroups + g: the masks of SEVERAL layers of one step in one launch (layer l: stream id sid + l * sid_stride, words
// at l * layer_words) -- a layer's 24 576 two-word blocks are launch-shaped (19 us for 6 MB), twelve layers in one grid are not.
__global__ __launch_bounds__(64) void attn_mask_kernel(const int* __restrict__ seq_len, const long long* __restrict__ mask_off, int heads,
                                                       int maxlen, unsigned thr16, unsigned long long seed, unsigned long long sid,
                                                       unsigned* __restrict__ mask_q, unsigned* __restrict__ mask_k, int ngroups,
                                                       unsigned long long sid_stride, long long layer_words) {
    const int layer = blockIdx.z / ngroups;
    const int g = blockIdx.z - layer * ngroups, seq = g / heads, head = g % heads;
    sid += (unsigned long long)layer * sid_stride;
    mask_q += (long long)layer * layer_words;
    mask_k += (long long)layer * layer_words;
    const int L = seq_len[seq], nkb = (L + 31) >> 5;
    const int qb = blockIdx.y, kp = blockIdx.x;
    if (qb >= nkb || 2 * kp >= nkb) return;
    const int lane = threadIdx.x, q = qb * 32 + (lane & 31), kb = 2 * kp + (lane >> 5);
    const long long base = mask_off[seq] + (long long)head * nkb * 32 * nkb;
    unsigned w = 0;
    const uint64_t idx0 = (((uint64_t)g * maxlen + q) * ((maxlen + 31) / 32) + kb) * 8;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        uint64_t z = seed + 0x9E3779B97F4A7C15ull * (sid + 1) + (idx0 + c) * 0xBF58476D1CE4E5B9ull;
        z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
        z ^= z >> 27; z *= 0x94D049BB133111EBull;
        z ^= z >> 31;
#pragma unroll
        for (int e = 0; e < 4; ++e) w |= ((unsigned)((z >> (16 * e)) & 0xffffu) >= thr16 ? 1u : 0u) << (4 * c + e);
    }
    if (kb < nkb) mask_q[base + (long long)q * nkb + kb] = w;
    unsigned tw = 0;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const unsigned long long b = __ballot((w >> j) & 1u);
        const unsigned mine = (lane >> 5) ? (unsigned)(b >> 32) : (unsigned)b;
        if ((lane & 31) == j) tw = mine;
    }
    if (kb < nkb) mask_k[base + (long long)(kb * 32 + (lane & 31)) * nkb + qb] = tw;
}

}  // namespace vbg

using namespace vbg;

extern "C" int vbg_attn(const vbg_attn_desc* desc, void* stream) {
    VBG_CHECK_ARG(desc != nullptr);
    const vbg_attn_desc& d = *desc;
    VBG_CHECK_ARG(d.mode >= VBG_ATTN_FWD && d.mode <= VBG_ATTN_DKV && d.heads > 0 && d.ntasks >= 0);
    if (d.ntasks == 0) return VBG_OK;
    VBG_CHECK_ARG(d.tasks && d.seq_len && d.seq_row0 && d.pad_off && d.qkv && d.out && d.lse);
    VBG_CHECK_ARG(d.qkv_ld % 8 == 0 && ((uintptr_t)d.qkv & 15) == 0 && d.qkv_plane % 8 == 0 && 6 * d.qkv_plane < 0x7fffffffll);
    VBG_CHECK_ARG(((uintptr_t)d.kbar & 15) == 0 && d.ldk % 4 == 0);
    if (d.out_pair) VBG_CHECK_ARG(d.mode == VBG_ATTN_FWD && ((uintptr_t)d.out_pair & 7) == 0 && d.oq_ld % 4 == 0 && d.oq_plane % 4 == 0);
    if (d.out_planes) VBG_CHECK_ARG(d.mode == VBG_ATTN_FWD && ((uintptr_t)d.out_planes & 7) == 0 && d.op_ld % 4 == 0 && d.op_plane % 4 == 0);
    VBG_CHECK_ARG(d.ldo % 4 == 0 && ((uintptr_t)d.out & 15) == 0 && d.ntok_pad % 32 == 0 && ((uintptr_t)d.lse & 15) == 0);
    if (d.mode != VBG_ATTN_FWD) {
        VBG_CHECK_ARG(d.dO && d.delta && d.do_ld % 8 == 0 && ((uintptr_t)d.dO & 15) == 0 && d.do_plane % 8 == 0 && 6 * d.do_plane < 0x7fffffffll);
        VBG_CHECK_ARG(((uintptr_t)d.delta & 15) == 0);
        if (d.mode == VBG_ATTN_DQ) VBG_CHECK_ARG(d.kbar != nullptr && d.o != nullptr && ((uintptr_t)d.o & 15) == 0);
    }
    VBG_CHECK_ARG(d.max_len >= 1 && d.max_len <= 512);          // (16 tiles of 32 rows: the size of the staged mask / statistic tables)
    const bool drop = d.mask_q != nullptr;
    if (drop) VBG_CHECK_ARG(d.mask_k && d.mask_off && d.keep_scale >= 1.0f);
    hipStream_t s = (hipStream_t)stream;
    const dim3 g(d.ntasks, d.heads), b(256);
    (void)hipGetLastError();
#define AT_GO(M)                                                                                      \
    do {                                                                                              \
        if (drop) hipLaunchKernelGGL((attn_kernel<M, true>), g, b, 0, s, d);                          \
        else hipLaunchKernelGGL((attn_kernel<M, false>), g, b, 0, s, d);                              \
    } while (0)
    if (d.mode == VBG_ATTN_FWD) AT_GO(VBG_ATTN_FWD);
    else if (d.mode == VBG_ATTN_DQ) AT_GO(VBG_ATTN_DQ);
    else AT_GO(VBG_ATTN_DKV);
#undef AT_GO
    VBG_LAUNCH_RET();
}

extern "C" unsigned vbg_attn_drop_thr16(float drop_p) {
    double t = (double)drop_p * 65536.0 + 0.5;
    if (t < 0) t = 0;
    if (t > 65535.0) t = 65535.0;
    return (unsigned)t;
}

extern "C" int vbg_attn_mask(const int* seq_len, const long long* mask_off, int nseq, int heads, int maxlen, float drop_p,
                             unsigned long long seed, unsigned long long stream_id, unsigned* mask_q, unsigned* mask_k, void* stream) {
    VBG_CHECK_ARG(nseq >= 0 && heads > 0 && maxlen >= 0 && drop_p > 0.f && drop_p < 1.f);
    if (nseq == 0 || maxlen == 0) return VBG_OK;
    VBG_CHECK_ARG(seq_len && mask_off && mask_q && mask_k);
    const int nkb = (maxlen + 31) / 32;
    VBG_LAUNCH(attn_mask_kernel, dim3((nkb + 1) / 2, nkb, nseq * heads), dim3(64), 0, (hipStream_t)stream, seq_len, mask_off, heads, maxlen,
               vbg_attn_drop_thr16(drop_p), seed, stream_id, mask_q, mask_k, nseq * heads, 0ull, 0ll);
    VBG_LAUNCH_RET();
}

extern "C" int vbg_attn_mask_layers(const int* seq_len, const long long* mask_off, int nseq, int heads, int maxlen, float drop_p,
                                    unsigned long long seed, unsigned long long stream_id0, unsigned long long stream_id_stride, int nlayers,
                                    long long layer_words, unsigned* mask_q, unsigned* mask_k, void* stream) {
    VBG_CHECK_ARG(nseq >= 0 && heads > 0 && maxlen >= 0 && drop_p > 0.f && drop_p < 1.f && nlayers >= 0 && layer_words >= 0);
    if (nseq == 0 || maxlen == 0 || nlayers == 0) return VBG_OK;
    VBG_CHECK_ARG(seq_len && mask_off && mask_q && mask_k);
    const int nkb = (maxlen + 31) / 32, ngroups = nseq * heads;
    VBG_CHECK_ARG(ngroups <= 65535);
    const int per = 65535 / ngroups;                       // layers per launch (grid z limit)
    for (int l0 = 0; l0 < nlayers; l0 += per) {
        const int nl = (nlayers - l0 < per) ? (nlayers - l0) : per;
        VBG_LAUNCH(attn_mask_kernel, dim3((nkb + 1) / 2, nkb, ngroups * nl), dim3(64), 0, (hipStream_t)stream, seq_len, mask_off, heads, maxlen,
                   vbg_attn_drop_thr16(drop_p), seed, stream_id0 + (unsigned long long)l0 * stream_id_stride, mask_q + (long long)l0 * layer_words,
                   mask_k + (long long)l0 * layer_words, ngroups, stream_id_stride, layer_words);
    }
    VBG_LAUNCH_RET();
}
