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
#define AT_MFMA_H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(at_f16x8, a), __builtin_bit_cast(at_f16x8, b), c, 0, 0, 0)

// ---- the two-piece fp16 form (FORM 1; round 5) and its one-product cut (FORM 2: `amp`) -------------------------------------------------
// Operands arrive as the fp16-pair planes of csrc/gemm_planes.hip: x = hi + lo' 2^-11, hi = fp16(x), lo' = fp16((x - hi) 2^11) (round to
// nearest both; q / k / v as they are, dO scaled by the power of two its producer derived from a bound: vbg_attn_desc.do_amax).  A product
// is hi hi + 2^-11 (hi lo' + lo' hi): three fp16 MFMAs instead of six bf16 ones, two planes through LDS instead of three.
//   * Score-type products (S = K Q^T, dP = V dO^T: a reduction over the 64 columns of a head): ONE accumulator.  The 2^-11 is folded into
//     the stationary operand (the workgroup's own rows, one row per lane = the N index of the product), which is first scaled by the
//     per-lane power of two c = 2^(13 - exponent(max |row|)) so that the folded pieces stay normal fp16 numbers: fragments hi c,
//     lo' c 2^-11, hi c 2^-11; the product leaves as c S and is scaled back per lane.  (Folding into unscaled fragments -- the first
//     build -- loses the cross terms of every element below 2^-3: hi 2^-11 is an fp16 subnormal there.)
//   * Second products (O^T += V^T P^T, dQ^T += K^T dS^T, ...: a reduction over up to 512 streamed rows): the cross products go into
//     accumulators of their own that live for ONE tile and are folded into the main ones times 2^-11 behind it.  The matrix pipe
//     aligns the products of an instruction to the accumulator it adds them to; against a running sum over hundreds of rows a cross
//     product (2^-12 of a main product, 2^-21 of the sum) arrived with a bit or two -- the second build, measured: switching the
//     cross products off moved O by 1e-6 of an error of 2.4e-4.
//   * Register operands are scaled into fp16's range before they are split (hi, lo' as above): probabilities (<= 1.12) by 2^13, score
//     gradients by a running per-lane power of two derived from a BOUND known before the values are (|dS| <= max |dP| keep + max
//     |delta| over the tile, P <= 1); when the bound grows the lane's accumulators are multiplied by the exact ratio.  Nothing can
//     overflow: every scale comes from a maximum or a bound of what it scales.  The streamed operand is used as it lies in LDS.
typedef _Float16 at_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 at_h2 __attribute__((ext_vector_type(2)));
typedef float at_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float at_hlo(unsigned u) { return (float)__builtin_bit_cast(at_h2, u).x; }
__device__ __forceinline__ float at_hhi(unsigned u) { return (float)__builtin_bit_cast(at_h2, u).y; }

template <int MODE, bool DROP, int FORM>
__global__ __launch_bounds__(256, 2) void attn_kernel(const vbg_attn_desc p) {
    constexpr bool FWD = MODE == VBG_ATTN_FWD, DQ = MODE == VBG_ATTN_DQ, DKV = MODE == VBG_ATTN_DKV;
    constexpr int NS = FWD ? 1 : 2;                                   // stationary operands
    // FORM 0: three bf16 planes, six piece products; FORM 1: two fp16 planes, three; FORM 2: the hi plane alone, one (`amp`)
    constexpr int NP = FORM == 0 ? 3 : (FORM == 1 ? 2 : 1);           // planes of a streamed operand that travel through LDS
    // FORM 1, DKV (two stationary operands, four accumulator sets: the register file is full): the stationary fragments stay as stored
    // (hi, lo': two sets instead of three) and the score products keep their cross terms in a 16-register accumulator of their own
    constexpr bool SXA = FORM == 1 && DKV;
    constexpr int NST = FORM == 2 ? 1 : (SXA ? 2 : 3);                // fragment sets of a stationary operand (FORM 1: hi c, lo' c 2^-11, hi c 2^-11)
    constexpr int NE = FORM == 0 ? 3 : (FORM == 1 ? 2 : 1);           // pieces of a register operand (FORM 1: hi, lo')
    constexpr int AT_OP = NP * AT_PL, AT_STAGE = 2 * AT_OP;           // (shadow the three-plane sizes of the file scope)
    constexpr float ESC = FORM == 0 ? 1.f : 8192.f, IESC = FORM == 0 ? 1.f : 1.f / 8192.f;     // scale of the probabilities as a register operand
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
    unsigned dvo[2][NP];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int q = 0; q < NP; ++q) dvo[o][q] = (unsigned)(((long long)q * splane[o] + (long long)drow * sld[o]) * 2 + dchunk * 16);
    auto issue = [&](int stage, int t) {
        const unsigned inv = (t * 32 + drow < L) ? 0u : AT_INVALID;          // rows past the sequence land as zeros
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(sbase[o]), 0, (int)0x80000000u, 0x00020000);
            const int soff = (int)((long long)t * 32 * sld[o] * 2);
#pragma unroll
            for (int q = 0; q < NP; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (at_lds_ptr)(smem + stage * AT_STAGE + o * AT_OP + q * AT_PL + wave * 1024), 16,
                                                         (int)(dvo[o][q] | inv), soff, 0, 0);
        }
    };

    // ---- stationary fragments (B operands of the score-type products): own row lr, k-step ks = 16 B at column 16 ks + 8 lh ----
    at_u32x4 st[NS][NST][4];
    at_u32x4 dlo[FORM == 1 && DQ ? 4 : 1];          // FORM 1, DQ: the lo' pieces of dO as stored, for delta in the prologue only
    at_u32x4 dhi[FORM != 0 && DQ ? 4 : 1];          // FORM 1 / 2, DQ: the hi pieces of dO as stored
    float cinv[NS];                                 // FORM 1 / 2: 1 / (the power of two this lane's stationary row was scaled by)
    {
        const bool ok = own0 + lr < L;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if constexpr (FORM == 0 || SXA) {
#pragma unroll
                for (int q = 0; q < NP; ++q)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        at_u32x4 v = {0u, 0u, 0u, 0u};
                        if (ok) v = *reinterpret_cast<const at_u32x4*>(tbase[s] + (long long)q * tplane[s] + (long long)(own0 + lr) * tld[s] + 16 * ks + 8 * lh);
                        st[s][q][ks] = v;
                    }
                cinv[s] = 1.f;
            } else {
                at_u32x4 rh[4], rl[4];
                float mx = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    rh[ks] = at_u32x4{0u, 0u, 0u, 0u}; rl[ks] = rh[ks];
                    if (ok) {
                        rh[ks] = *reinterpret_cast<const at_u32x4*>(tbase[s] + (long long)(own0 + lr) * tld[s] + 16 * ks + 8 * lh);
                        if constexpr (FORM == 1) rl[ks] = *reinterpret_cast<const at_u32x4*>(tbase[s] + tplane[s] + (long long)(own0 + lr) * tld[s] + 16 * ks + 8 * lh);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) mx = fmaxf(mx, fmaxf(fabsf(at_hlo(rh[ks][j])), fabsf(at_hhi(rh[ks][j]))));
                }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));                       // (the two half-waves hold the two halves of the row)
                const float2 cs = vbg_pow2_scale(__float_as_uint(mx));        // (max = 0, inf or NaN: no scaling)
                cinv[s] = cs.y;
                const float c0 = cs.x, c1 = cs.x * 0.00048828125f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    if constexpr (DQ) { if (s == 1) { dhi[ks] = rh[ks]; if constexpr (FORM == 1) dlo[ks] = rl[ks]; } }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const at_f2 h = {at_hlo(rh[ks][j]), at_hhi(rh[ks][j])};
                        st[s][0][ks][j] = __builtin_bit_cast(unsigned, __builtin_convertvector(h * c0, at_h2));            // hi c
                        if constexpr (FORM == 1) {
                            const at_f2 l = {at_hlo(rl[ks][j]), at_hhi(rl[ks][j])};
                            st[s][1][ks][j] = __builtin_bit_cast(unsigned, __builtin_convertvector(l * c1, at_h2));        // lo' c 2^-11
                            st[s][2][ks][j] = __builtin_bit_cast(unsigned, __builtin_convertvector(h * c1, at_h2));        // hi c 2^-11
                        }
                    }
                }
            }
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
    // the piece products of one k-step, smallest first, into ONE accumulator (FORM 1: the stationary side carries the 2^-11)
    auto sp3 = [&](const at_u32x4 (&fa)[NP], const at_u32x4 (&sb)[NST][4], int ks, f32x16& acc, f32x16& sx, int first, int last) {
        if constexpr (FORM == 0) {
#pragma unroll
            for (int t = 0; t < 6; ++t) if (t >= first && t < last) acc = AT_MFMA(fa[qa[t]], sb[qb[t]][ks], acc);
        } else if constexpr (SXA) {
            if (first <= 0 && 0 < last) sx = AT_MFMA_H(fa[1], sb[0][ks], sx);             // lo' x hi
            if (first <= 1 && 1 < last) sx = AT_MFMA_H(fa[0], sb[1][ks], sx);             // hi x lo'
            if (first <= 2 && 2 < last) acc = AT_MFMA_H(fa[0], sb[0][ks], acc);           // hi x hi
        } else if constexpr (FORM == 1) {
            if (first <= 0 && 0 < last) acc = AT_MFMA_H(fa[1], sb[2][ks], acc);
            if (first <= 1 && 1 < last) acc = AT_MFMA_H(fa[0], sb[1][ks], acc);
            if (first <= 2 && 2 < last) acc = AT_MFMA_H(fa[0], sb[0][ks], acc);
        } else {
            if (first <= 0 && 0 < last) acc = AT_MFMA_H(fa[0], sb[0][ks], acc);
        }
    };
    auto sprod = [&](const unsigned char* img, const at_u32x4 (&sb)[NST][4], f32x16& acc) {
        f32x16 sx;
        if constexpr (SXA) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sx[r] = 0.f;
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            at_u32x4 fa[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) fa[q] = *reinterpret_cast<const at_u32x4*>(img + q * AT_PL + fra[ks]);
            sp3(fa, sb, ks, acc, sx, 0, 6);
        }
        if constexpr (SXA) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = fmaf(sx[r], 0.00048828125f, acc[r]);
        }
    };
    // second product: D[column d of the streamed operand][own row] += sum over streamed rows X_stream[row][d] E[row][own row]
    // fragments of the streamed operand for the second product, group g = 2 ks + db: the NP planes by transposing reads, used as they lie
    constexpr int NF = NP;
    typedef __attribute__((address_space(3))) unsigned char* lds_bytes;
    auto trd = [&](lds_bytes im, int g, at_u32x4 (&f)[NF]) {
        const int ks = g >> 1, db = g & 1;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const at_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((at_lds_v4s)(im + q * AT_PL + ks * 2048 + tra[0][db]));
            const at_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((at_lds_v4s)(im + q * AT_PL + ks * 2048 + tra[1][db]));
            const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
            f[q] = at_u32x4{l2.x, l2.y, h2.x, h2.y};
        }
    };
    // FORM 1: the two cross products of the SECOND product go into accumulators of their own (`ax`, zero at the start of every tile's
    // product and folded into the main ones times 2^-11 behind it): the main accumulators run over up to 512 streamed rows and the matrix
    // pipe aligns the 16 products of an instruction to the accumulator it adds them to -- cross products 2^-12 of a main product and
    // 2^-21 of the running sum arrived with one or two bits (measured: switching them off changed O by 1e-6 of an error of 2.4e-4).
    // Over one tile the cross sums are of the size of their own terms.  (The score-type products reduce over 64 columns only: there the
    // one accumulator with the folded 2^-11 is accurate, lse to 8e-7.)
    auto tp3 = [&](const at_u32x4 (&fa)[NF], const at_u32x4 (&bp)[NE][2], int ks, f32x16& acc, f32x16& ax, int first, int last) {
        if constexpr (FORM == 0) {
#pragma unroll
            for (int t = 0; t < 6; ++t) if (t >= first && t < last) acc = AT_MFMA(fa[qa[t]], bp[qb[t]][ks], acc);
        } else if constexpr (FORM == 1) {
            if (first <= 0 && 0 < last) ax = AT_MFMA_H(fa[1], bp[0][ks], ax);             // lo' x hi
            if (first <= 1 && 1 < last) ax = AT_MFMA_H(fa[0], bp[1][ks], ax);             // hi x lo'
            if (first <= 2 && 2 < last) acc = AT_MFMA_H(fa[0], bp[0][ks], acc);           // hi x hi
        } else {
            if (first <= 0 && 0 < last) acc = AT_MFMA_H(fa[0], bp[0][ks], acc);
        }
    };
    auto tfold = [&](f32x16 (&acc)[2], const f32x16 (&ax)[FORM == 1 ? 2 : 1]) {
        if constexpr (FORM == 1) {
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[db][r] = fmaf(ax[db][r], 0.00048828125f, acc[db][r]);
        }
    };
    auto tprod = [&](const unsigned char* img, const at_u32x4 (&bp)[NE][2], f32x16 (&acc)[2]) {
        lds_bytes im = (lds_bytes)img;
        f32x16 ax[FORM == 1 ? 2 : 1];
        if constexpr (FORM == 1) {
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) ax[db][r] = 0.f;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            at_u32x4 fa[NF];
            trd(im, g, fa);
            tp3(fa, bp, g >> 1, acc[g & 1], ax[FORM == 1 ? (g & 1) : 0], 0, 6);
        }
        if constexpr (FORM == 1) tfold(acc, ax);
    };

    // low-precision companion of tprod (first plane of both operands only): the forward's Kbar^T += K^T P^T
    auto kprod = [&](const unsigned char* img, const at_u32x4 (&bh)[2], f32x16 (&acc)[2]) {
        lds_bytes im = (lds_bytes)img;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const at_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((at_lds_v4s)(im + ks * 2048 + tra[0][db]));
                const at_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((at_lds_v4s)(im + ks * 2048 + tra[1][db]));
                const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                const at_u32x4 fa = at_u32x4{l2.x, l2.y, h2.x, h2.y};
                if constexpr (FORM == 0) acc[db] = AT_MFMA(fa, bh[ks], acc[db]);
                else acc[db] = AT_MFMA_H(fa, bh[ks], acc[db]);
            }
    };

    // The same two products with element-wise work woven in by hand: `work(pi)` (pi = 0..7: the register pair 2 pi, 2 pi + 1 of the
    // D layout) is issued behind every third MFMA, fenced so that the compiler keeps the order -- a 32 x 32 x 16 MFMA holds the matrix
    // pipe for 32 clocks, ~25 VALU instructions of a pair fit behind three of them.  (hipcc's own schedule puts the ~200 VALU
    // instructions of a tile in one block between the products; with one wave per SIMD nothing then overlaps the matrix pipe.)
    // Fragments are read one group ahead.
    // (the six / three / one piece products of a k-step are cut in two groups with a slot for element-wise work behind each)
    constexpr int NPR = FORM == 0 ? 6 : (FORM == 1 ? 3 : 1), CUT = FORM == 0 ? 3 : (FORM == 1 ? 2 : 1);
    auto sprod_woven = [&](const unsigned char* img, const at_u32x4 (&sb)[NST][4], f32x16& acc, auto&& work) {
        at_u32x4 fa[2][NP];
        f32x16 sx;
        if constexpr (SXA) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sx[r] = 0.f;
        }
#pragma unroll
        for (int q = 0; q < NP; ++q) fa[0][q] = *reinterpret_cast<const at_u32x4*>(img + q * AT_PL + fra[0]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks + 1 < 4) {
#pragma unroll
                for (int q = 0; q < NP; ++q) fa[(ks + 1) & 1][q] = *reinterpret_cast<const at_u32x4*>(img + q * AT_PL + fra[ks + 1]);
            }
            sp3(fa[ks & 1], sb, ks, acc, sx, 0, CUT);
            __builtin_amdgcn_sched_barrier(0);
            work(2 * ks);
            __builtin_amdgcn_sched_barrier(0);
            sp3(fa[ks & 1], sb, ks, acc, sx, CUT, NPR);
            __builtin_amdgcn_sched_barrier(0);
            work(2 * ks + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (SXA) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = fmaf(sx[r], 0.00048828125f, acc[r]);
        }
    };
    auto tprod_woven = [&](const unsigned char* img, const at_u32x4 (&bp)[NE][2], f32x16 (&acc)[2], auto&& work) {
        lds_bytes im = (lds_bytes)img;
        at_u32x4 fa[2][NF];
        f32x16 ax[FORM == 1 ? 2 : 1];
        if constexpr (FORM == 1) {
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) ax[db][r] = 0.f;
        }
        trd(im, 0, fa[0]);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g + 1 < 4) trd(im, g + 1, fa[(g + 1) & 1]);
            tp3(fa[g & 1], bp, g >> 1, acc[g & 1], ax[FORM == 1 ? (g & 1) : 0], 0, CUT);
            __builtin_amdgcn_sched_barrier(0);
            work(2 * g);
            __builtin_amdgcn_sched_barrier(0);
            tp3(fa[g & 1], bp, g >> 1, acc[g & 1], ax[FORM == 1 ? (g & 1) : 0], CUT, NPR);
            __builtin_amdgcn_sched_barrier(0);
            work(2 * g + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (FORM == 1) tfold(acc, ax);
    };
    // exact three-way split of the register pair (2 pi, 2 pi + 1) into its slot of the B-operand fragments (as at_split16)
    auto split_pair = [&](float a, float b, int pi, at_u32x4 (&bp)[NE][2]) {
        if constexpr (FORM == 0) {
            const float ra = a - __uint_as_float(__float_as_uint(a) & 0xffff0000u), rb = b - __uint_as_float(__float_as_uint(b) & 0xffff0000u);
            const float sa = ra - __uint_as_float(__float_as_uint(ra) & 0xffff0000u), sb2 = rb - __uint_as_float(__float_as_uint(rb) & 0xffff0000u);
            bp[0][pi >> 2][pi & 3] = __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
            bp[1][pi >> 2][pi & 3] = __builtin_amdgcn_perm(__float_as_uint(rb), __float_as_uint(ra), 0x07060302u);
            bp[2][pi >> 2][pi & 3] = __builtin_amdgcn_perm(__float_as_uint(sb2), __float_as_uint(sa), 0x07060302u);
        } else {                                    // fp16 pieces of values the caller scaled to the top of fp16's range; FORM 2: the hi piece
            const at_h2 h = __builtin_convertvector((at_f2){a, b}, at_h2);
            bp[0][pi >> 2][pi & 3] = __builtin_bit_cast(unsigned, h);
            if constexpr (FORM == 1) {
                const at_f2 hf = __builtin_convertvector(h, at_f2);
                bp[1][pi >> 2][pi & 3] = __builtin_bit_cast(unsigned, __builtin_convertvector((at_f2){(a - hf.x) * 2048.f, (b - hf.y) * 2048.f}, at_h2));      // lo' = remainder 2^11
            }
        }
    };
    auto split16 = [&](const float (&x)[16], at_u32x4 (&bp)[NE][2]) {
#pragma unroll
        for (int pi = 0; pi < 8; ++pi) split_pair(x[2 * pi], x[2 * pi + 1], pi, bp);
    };
    // scale of dO's planes: the power of two its producer derived from the amax slot (FORM 0 planes are exact: 1)
    float dinv = 1.f;
    if constexpr (!FWD && FORM != 0) {
        if (p.do_amax) dinv = vbg_pow2_scale(vbg_amax_read(p.do_amax)).y;
    }
    // running per-lane scale of the score gradients as a register operand (FORM 1 / 2): the power of two that brings the largest BOUND
    // seen so far to [2^13, 2^14); esc = 0: no tile yet
    float esc = 0.f, bmax = 0.f;
    auto ds_scale = [&](float bound, f32x16 (&acc)[2]) {                 // bound >= every |dS| of this lane in this tile (both half-waves)
        bound = fmaxf(bound, __shfl_xor(bound, 32, 64));
        if (bound > bmax) {
            bmax = bound;
            const float ne = vbg_pow2_scale(__float_as_uint(bound)).x;
            if (ne != esc) {
                if (esc != 0.f) {
                    const float f = ne / esc;                            // (a power of two: exact)
#pragma unroll
                    for (int db = 0; db < 2; ++db)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[db][r] *= f;
                }
                esc = ne;
            }
        }
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
                    float lo_e, hi_e;
                    if constexpr (FORM == 0) {
                        const unsigned h = st[1][0][ks][j], m = st[1][1][ks][j], l = st[1][2][ks][j];
                        lo_e = (__uint_as_float(h << 16) + __uint_as_float(m << 16)) + __uint_as_float(l << 16);
                        hi_e = (__uint_as_float(h & 0xffff0000u) + __uint_as_float(m & 0xffff0000u)) + __uint_as_float(l & 0xffff0000u);
                    } else {                      // dO 2^e = hi + lo' 2^-11 (FORM 2: the hi piece alone is what the products see)
                        const unsigned h = dhi[ks][j];
                        lo_e = at_hlo(h); hi_e = at_hhi(h);
                        if constexpr (FORM == 1) {
                            lo_e = fmaf(at_hlo(dlo[ks][j]), 0.00048828125f, lo_e);
                            hi_e = fmaf(at_hhi(dlo[ks][j]), 0.00048828125f, hi_e);
                        }
                    }
                    del_own = fmaf(lo_e, ov[2 * j], del_own);
                    del_own = fmaf(hi_e, ov[2 * j + 1], del_own);
                }
            }
        }
        del_own = (del_own + __shfl_xor(del_own, 32, 64)) * dinv;          // (FORM 1 / 2: dO's planes carry the scale 2^e)
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
                for (int r = 0; r < 16; ++r) x[r] = s[r] * (scale * cinv[0]);
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
                        for (int j = 0; j < 4; ++j) {
                            if constexpr (FORM == 0) bh[ks][j] = __builtin_amdgcn_perm(__float_as_uint(x[8 * ks + 2 * j + 1]), __float_as_uint(x[8 * ks + 2 * j]), 0x07060302u);
                            else bh[ks][j] = __builtin_bit_cast(unsigned, __builtin_convertvector((at_f2){x[8 * ks + 2 * j], x[8 * ks + 2 * j + 1]}, at_h2));
                        }
                    kprod(im0, bh, acc1);
                }
                if constexpr (DROP) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) x[r] = ((mw >> ((r & 3) + 8 * (r >> 2))) & 1u) ? x[r] * (keep_scale * ESC) : 0.f;
                } else if constexpr (FORM != 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) x[r] *= ESC;
                }
                at_u32x4 bp[NE][2];
                split16(x, bp);
                tprod(im1, bp, acc0);
            } else {
                f32x16 dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) dp[r] = 0.f;
                float pr[16], ds[16];
                if constexpr (DQ) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) pr[r] = __expf(s[r] * (scale * cinv[0]) - m_own) * il_own;
                    sprod(im1, st[1], dp);
                    const float dmul = dinv * cinv[1];
                    float gmax = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float g = dp[r] * dmul;
                        if constexpr (DROP) g = ((mw >> ((r & 3) + 8 * (r >> 2))) & 1u) ? g * keep_scale : 0.f;
                        ds[r] = pr[r] * (g - del_own);
                        dsum = fmaf(pr[r], g, dsum);
                        gmax = fmaxf(gmax, fabsf(g));
                    }
                    if constexpr (FORM != 0) {                           // |dS| <= |g| + |delta| (P <= 1): the lane's scale before the values exist
                        ds_scale(gmax + fabsf(del_own), acc0);
#pragma unroll
                        for (int r = 0; r < 16; ++r) ds[r] *= esc;
                    }
                    at_u32x4 bp[NE][2];
                    split16(ds, bp);
                    tprod(im0, bp, acc0);                                  // dQ^T += K^T dS^T
                } else {
                    // statistics of the streamed queries: register r = query 8 (r >> 2) + 4 lh + (r & 3) of the tile (loaded an
                    // iteration ahead: a use of a fresh global load in here would drain the tile DMA in flight)
                    // Phase A: P, its dropped / scaled form and the split of that only need S -> woven into the dP product.
                    float pv[16];
                    at_u32x4 bpv[NE][2], bpk[NE][2];
                    sprod_woven(im1, st[1], dp, [&](int pi) {
                        // registers 2 pi, 2 pi + 1 = queries 8 (pi >> 1) + 2 (pi & 1) (+ 1) of this half-wave's rows
                        const float2 mq = *reinterpret_cast<const float2*>(stile + 8 * (pi >> 1) + 2 * (pi & 1));
                        const float2 iq = *reinterpret_cast<const float2*>(stile + 512 + 8 * (pi >> 1) + 2 * (pi & 1));
                        float pd[2];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int r = 2 * pi + e;
                            pv[r] = __expf(s[r] * (scale * cinv[0]) - (e ? mq.y : mq.x)) * (e ? iq.y : iq.x);
                            pd[e] = pv[r] * ESC;
                            if constexpr (DROP) pd[e] = ((mw >> ((r & 3) + 8 * (r >> 2))) & 1u) ? pv[r] * (keep_scale * ESC) : 0.f;
                        }
                        split_pair(pd[0], pd[1], pi, bpv);
                    });
                    const float dmul = dinv * cinv[1];
                    if constexpr (FORM != 0) {
                        // the lane's scale of dS from a bound: |dS| <= max |dP| keep_scale + max |delta| over the tile's queries (P <= 1)
                        float gmax = 0.f, dmax = 0.f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) gmax = fmaxf(gmax, fabsf(dp[r]));
#pragma unroll
                        for (int pi = 0; pi < 8; ++pi) {
                            const float2 dq = *reinterpret_cast<const float2*>(stile + 1024 + 8 * (pi >> 1) + 2 * (pi & 1));
                            dmax = fmaxf(dmax, fmaxf(fabsf(dq.x), fabsf(dq.y)));
                        }
                        ds_scale(gmax * (dmul * keep_scale) + dmax, acc0);
                    }
                    const float dse = FORM == 0 ? 1.f : esc;
                    // Phase B: dS and its split need dP -> woven into the dV product
                    tprod_woven(im1, bpv, acc1, [&](int pi) {                                 // dV^T += dO^T Pd
                        const float2 dq = *reinterpret_cast<const float2*>(stile + 1024 + 8 * (pi >> 1) + 2 * (pi & 1));
                        float dsv[2];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int r = 2 * pi + e;
                            float g = dp[r] * dmul;
                            if constexpr (DROP) g = ((mw >> ((r & 3) + 8 * (r >> 2))) & 1u) ? g * keep_scale : 0.f;
                            dsv[e] = pv[r] * (g - (e ? dq.y : dq.x)) * dse;
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

    // (FORM 1 / 2: the score-gradient accumulators carry this lane's running scale; no tile seen: they are zero)
    const float dsi = (FORM == 0 || FWD) ? 1.f : (esc != 0.f ? 1.f / esc : 0.f);
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
        const float ilk = 1.0f / lt;                 // of Kbar (its probabilities were not scaled)
        const float il = ilk * IESC;                 // of O: the probabilities entered the second product times ESC
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
        if (want_kbar) store(acc1, ilk, head * 64, p.kbar + krow - orow);
        if (lh == 0) { p.lse[lsoff + own0 + lr] = m_run; p.lse[lsplane + lsoff + own0 + lr] = ilk; }
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
                acc0[db][4 * j] = acc0[db][4 * j] * dsi - c * kb.x; acc0[db][4 * j + 1] = acc0[db][4 * j + 1] * dsi - c * kb.y;
                acc0[db][4 * j + 2] = acc0[db][4 * j + 2] * dsi - c * kb.z; acc0[db][4 * j + 3] = acc0[db][4 * j + 3] * dsi - c * kb.w;
            }
        store(acc0, scale, head * 64, p.out);
        if (lh == 0) p.delta[lsoff + own0 + lr] = dnew;
      }
    } else {
      if (live) {
        store(acc0, scale * dsi, hid + head * 64, p.out);
        store(acc1, dinv * IESC, 2 * hid + head * 64, p.out);
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
                                                       unsigned long long sid_stride, long long layer_words, int g0) {
    // (ngroups = (sequence, head) groups per layer IN THIS LAUNCH, g0 = the first of them: more than 65 535 groups are cut into
    //  several launches by the callers -- grid z limit)
    const int layer = blockIdx.z / ngroups;
    const int g = g0 + (int)blockIdx.z - layer * ngroups, seq = g / heads, head = g % heads;
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
    VBG_CHECK_ARG(d.mode >= VBG_ATTN_FWD && d.mode <= VBG_ATTN_DKV && d.heads > 0 && d.ntasks >= 0 && d.form >= 0 && d.form <= 2);
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
#define AT_GO2(M, F)                                                                                  \
    do {                                                                                              \
        if (drop) hipLaunchKernelGGL((attn_kernel<M, true, F>), g, b, 0, s, d);                       \
        else hipLaunchKernelGGL((attn_kernel<M, false, F>), g, b, 0, s, d);                           \
    } while (0)
#define AT_GO(M)                                                                                      \
    do {                                                                                              \
        if (d.form == 0) AT_GO2(M, 0);                                                                \
        else if (d.form == 1) AT_GO2(M, 1);                                                           \
        else AT_GO2(M, 2);                                                                            \
    } while (0)
    if (d.mode == VBG_ATTN_FWD) AT_GO(VBG_ATTN_FWD);
    else if (d.mode == VBG_ATTN_DQ) AT_GO(VBG_ATTN_DQ);
    else AT_GO(VBG_ATTN_DKV);
#undef AT_GO
#undef AT_GO2
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
    const int nkb = (maxlen + 31) / 32, ngroups = nseq * heads;
    for (int g0 = 0; g0 < ngroups; g0 += 65535) {
        const int ng = (ngroups - g0 < 65535) ? (ngroups - g0) : 65535;
        VBG_LAUNCH(attn_mask_kernel, dim3((nkb + 1) / 2, nkb, ng), dim3(64), 0, (hipStream_t)stream, seq_len, mask_off, heads, maxlen,
                   vbg_attn_drop_thr16(drop_p), seed, stream_id, mask_q, mask_k, ng, 0ull, 0ll, g0);
    }
    VBG_LAUNCH_RET();
}

extern "C" int vbg_attn_mask_layers(const int* seq_len, const long long* mask_off, int nseq, int heads, int maxlen, float drop_p,
                                    unsigned long long seed, unsigned long long stream_id0, unsigned long long stream_id_stride, int nlayers,
                                    long long layer_words, unsigned* mask_q, unsigned* mask_k, void* stream) {
    VBG_CHECK_ARG(nseq >= 0 && heads > 0 && maxlen >= 0 && drop_p > 0.f && drop_p < 1.f && nlayers >= 0 && layer_words >= 0);
    if (nseq == 0 || maxlen == 0 || nlayers == 0) return VBG_OK;
    VBG_CHECK_ARG(seq_len && mask_off && mask_q && mask_k);
    const int nkb = (maxlen + 31) / 32, ngroups = nseq * heads;
    if (ngroups > 65535) {                                 // a layer alone exceeds the grid z limit: per layer, groups in chunks
        for (int l = 0; l < nlayers; ++l)
            for (int g0 = 0; g0 < ngroups; g0 += 65535) {
                const int ng = (ngroups - g0 < 65535) ? (ngroups - g0) : 65535;
                VBG_LAUNCH(attn_mask_kernel, dim3((nkb + 1) / 2, nkb, ng), dim3(64), 0, (hipStream_t)stream, seq_len, mask_off, heads, maxlen,
                           vbg_attn_drop_thr16(drop_p), seed, stream_id0 + (unsigned long long)l * stream_id_stride, mask_q + (long long)l * layer_words,
                           mask_k + (long long)l * layer_words, ng, 0ull, 0ll, g0);
            }
        VBG_LAUNCH_RET();
    }
    const int per = 65535 / ngroups;                       // layers per launch (grid z limit)
    for (int l0 = 0; l0 < nlayers; l0 += per) {
        const int nl = (nlayers - l0 < per) ? (nlayers - l0) : per;
        VBG_LAUNCH(attn_mask_kernel, dim3((nkb + 1) / 2, nkb, ngroups * nl), dim3(64), 0, (hipStream_t)stream, seq_len, mask_off, heads, maxlen,
                   vbg_attn_drop_thr16(drop_p), seed, stream_id0 + (unsigned long long)l0 * stream_id_stride, mask_q + (long long)l0 * layer_words,
                   mask_k + (long long)l0 * layer_words, ngroups, stream_id_stride, layer_words, 0);
    }
    VBG_LAUNCH_RET();
}
