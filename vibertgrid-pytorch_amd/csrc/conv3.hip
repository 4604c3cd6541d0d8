// 3x3 / stride 1 / pad 1 convolution over NHWC activations with ROW REUSE across the three horizontal taps (gfx950).
//
// Replaces, for the wide trunk / FPN / segmentation-head convolutions (reference model/ResNetFPN_ViBERTgrid.py:478-508, 612-648 and
// model/semantic_segmentation_head.py conv stacks, torch.nn.Conv2d(k=3, s=1, p=1)), the generic implicit GEMM of gemm.hip, whose
// k-loop fetches and splits the 128 x 16 activation tile once per filter tap: nine times per (row of taps x channel chunk).  Every
// matrix kernel of this library levels off where a CU ingests ~12.5 B / clk from L2, so the lever is bytes per
// product.  Here an output tile is 128 consecutive pixels = whole image rows (W <= 128) or a piece of one row; for a filter row kh and a chunk
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
#include <stdlib.h>
#include "../../include/vbg.h"

namespace vbg {

typedef unsigned c3_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 c3_bf16x8 __attribute__((ext_vector_type(8)));
constexpr unsigned C3_INVALID = 0x80000000u;       // outside every descriptor: the load returns 0 without touching memory

struct conv3_args {
    const float* X;        // [B, H, W, Cs]
    const float* Wt;       // [N, 3, 3, Cs]
    const unsigned short* Wp; // PW kernels: the filter as fp16-pair planes in k-tile order (conv3_wprep_kernel below), Wt unused
    const float* bias;     // [N] or null
    float* Y;              // [B, H, W, N]
    double* stats;         // BatchNorm slot workspace [slots][2][N] or null
    int stats_slots;
    int H, W, wsh, Cs, N, M;
    int roi;               // 0, or the number of 7 x 7 images: pixel tiles are 8 x 8 SLOT grids over compactly stored 7 x 7 images (below)
    int ksplit, csplit;    // reduction split over gridDim.z = ksplit * csplit blocks per tile: filter rows (1 or 3) x channel groups
    float* slab;           // [tiles][gridDim.z][BM * BN] partial tiles of the split form
    unsigned* tickets;     // [tiles] arrival counters of the split form, zero between launches
    int accumulate;
    const unsigned* a_amax; // F16 form: bits of max |X| (a device word written by the producer of X), or null: X is used as it is
    int dbg_no_ring_guard; // VBG_DEBUG_CONV3_NO_RING_GUARD=1: leave out the round-6 barrier behind the read of k-tile 0 (tests prove they can see the race)
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
typedef _Float16 c3_f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 c3_f16x2 __attribute__((ext_vector_type(2)));
// F16 form: two fp16 pieces of a pair of floats (a in the low half): hi = fp16(x), lo = fp16((x - hi) * 2^11), both rounded to
// nearest -- the remainder is exact in fp32 and its scaling keeps it clear of fp16's subnormals -- so x = hi + lo * 2^-11 up to
// 2^-23 |x| (11 + 1 + 11 bits: the remainder carries a sign).  |x| >= 65520 becomes inf (and the products NaN / inf): an operand
// outside fp16's range is visible, never silently clipped.
__device__ __forceinline__ void c3_split2(float a, float b, unsigned& hi, unsigned& lo) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {a, b};
    const f16x2_t h = __builtin_convertvector(v, f16x2_t);                        // v_cvt_pk_f16_f32: round to nearest even
    const f16x2_t l = __builtin_convertvector((v - __builtin_convertvector(h, f32x2_t)) * 2048.f, f16x2_t);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
// the value of an amax slot (VBG_AMAX_WORDS = 64 words, see include/vbg.h): the max over its words, uniform across the wave
__device__ __forceinline__ unsigned c3_amax_read(const unsigned* slot) {
    unsigned v = slot[(threadIdx.x & (VBG_AMAX_WORDS - 1)) * VBG_AMAX_STRIDE];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o, 64));
    return __builtin_amdgcn_readfirstlane(v);
}
// power-of-two scale that brings a tensor whose largest magnitude has the bit pattern `amax_bits` to [2^13, 2^14): (scale, 1 / scale).
// Exact in fp32 (the scaled operand's pieces see the same significand bits); amax = 0 or below 2^-100: no scaling.
__device__ __forceinline__ float2 c3_pow2_scale(unsigned amax_bits) {
    const int e = (int)((amax_bits >> 23) & 0xffu);            // biased exponent of amax
    if (e < 27 || e > 240) return make_float2(1.f, 1.f);
    return make_float2(__uint_as_float((unsigned)(267 - e) << 23), __uint_as_float((unsigned)(e - 13) << 23));
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

// software-pipelined loop: after MFMA M one LDS read (the next k-tile's fragments) while any are left, then its share of the NV VALU
// and ND LDS-write instructions of the region
template <int M, int NM, int NRD, int NV, int ND>
struct c3_pipe2 {
    static __device__ __forceinline__ void run() {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (M < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        constexpr int v = ((M + 1) * NV) / NM - (M * NV) / NM;
        if constexpr (v > 0) __builtin_amdgcn_sched_group_barrier(0x002, v, 0);
        constexpr int w = ((M + 1) * ND) / NM - (M * ND) / NM;
        if constexpr (w > 0) __builtin_amdgcn_sched_group_barrier(0x200, w, 0);
        if constexpr (M + 1 < NM) c3_pipe2<M + 1, NM, NRD, NV, ND>::run();
    }
};

// BM = 128 pixels per tile, or 64 for the late stages whose pixel count would leave half the chip without a 128-pixel tile
// F16: the arithmetic form -- false: three bf16 pieces per operand, six piece products (the library's default everywhere); true: two
// fp16 pieces, three piece products with the two cross products in their own accumulators (scaled by 2^11), for operands inside fp16's
// range (forward activations and filters; NOT gradients).  Measured against fp64 on the same data the two forms are equally accurate
// (1.7e-7 vs 2.5e-7 of the summed magnitudes) and F16 needs half the matrix-core work: 457 vs 701 us at 256 x 256, 128 x 128 pixels.
//
// BN = 128 filters per tile, or 64 (the 64-channel stage of the trunk: waves 2 x 2 of BM / 2 pixels x 32 filters).
//
// ROI mode (p.roi = number of images, H = W = 7: the [N * 49, C] region-of-interest maps of the field-type head, reference
// model/field_type_classification_head.py:64-75): a 128-row tile is TWO images, each laid out as an 8 x 8 grid of slots whose
// eighth row and column do not exist in memory -- their loads carry the invalid offset and read 0, so slot x = 7 is the zero pixel
// between image rows that the W >= 16 layout inserts explicitly, and slot row y = 7 the zero row under the image.  Slots map to the
// compact rows (image * 49 + y * 7 + x) in the loader and in the epilogue (invalid slots are neither stored nor counted in the
// statistics); 49 of 64 MFMA rows are useful, against which the generic kernel's nine-fold re-fetch and re-split costs more.
//
// PW ("pre-split weights", F16 form, 128-pixel tiles): the filter operand does not pass through registers at all.  conv3_wprep_kernel
// writes it ONCE per weight version as fp16-pair planes in the order this kernel walks it: one contiguous 64 * BN byte block per
// (filter tile, tap, 16-channel chunk) = [plane][32-row block][8-channel half][row][8 x fp16] -- the image a lane-linear LDS-DMA
// lands conflict free for the ds_read_b128 fragment pattern (a 16-lane group reads 16 different 16-byte bank quads).  The k-loop then
// issues BN / 64 `buffer_load ... lds` per wave and k-tile into a ring of NSB stages (counted vmcnt, D = NSB - 1 tiles in flight) and
// spends no VALU and no ds_write on the filter: 4 bytes per element from L2 as before (two fp16 pieces = one fp32), in full 128-byte
// lines instead of 64-byte row pieces, and three quarters of the kernel's split work gone (the activation super-tile, loaded every
// third k-tile, is the rest).
// PWM = 2: the same with a software-pipelined k-loop -- the fragments of k-tile t + 1 are read from LDS in the gaps of the MFMAs of
// k-tile t (two register sets), the activation super-tile is stored one k-tile earlier (loaded at kw = 0, split and stored at kw = 1,
// first read during kw = 2), and the DMA runs one k-tile further ahead of its readers.  A wave's k-tile is then MFMAs + one barrier: what
// the one-round launches of the trunk (one wave per SIMD, nobody to hide the LDS latency behind) were missing.  Same products in the
// same order per accumulator: bit-identical.
// ONEP (`amp`, PWM = 2 only): ONE product -- the hi pieces of both operands (x rounded to fp16: the operand of the reference's autocast
// convolution; gradients scaled into range as in the pair form), fp32 accumulation.  The lo pieces are neither written, loaded (128-filter
// tiles: the hi plane is the first half of a k-tile's block) nor read.
#ifndef VBG_C3_NSB64
#define VBG_C3_NSB64 4          // ring stages of the pipelined pre-split kernel on 64-filter tiles (4 KB each); 8 measured: no change (round 6)
#endif
#ifndef VBG_C3_ASMW
#define VBG_C3_ASMW 0           // 1: the pipelined loop's activation pieces go to LDS through inline-asm ds_write_b64 (c3_lds_store_b64); measured: no change
#endif
// (Round 6 experiment, compiled out by default: -DVBG_C3_ASMW=1 / -DVBG_C3_NSB64=8.  Both were built on the reading that the one-round
//  64-filter launches wait for their filter DMA; four builds, bit-identical results, every cfg2 shape within +-1 % and the step within
//  0.2 % -- tools/calls/r6_call09.sh, gpurun_out/r6c09_*: the reading was wrong, the switches stay for the record.)
// An LDS store the compiler does not see as one.  hipcc puts `s_waitcnt vmcnt(0)` in front of every LDS WRITE that follows an LDS-DMA in
// program order (it cannot prove that the DMA's destination and the store do not overlap), i.e. in front of the activation pieces of
// every third k-tile -- which waits for the filter DMA issued ONE k-tile earlier to land: the ring's lead is gone exactly where a
// 64-filter k-tile (six MFMAs) needs it most (ISA of round 5: `s_waitcnt vmcnt(0)` + 4 ds_write_b64 in the kw = 1 tile).  The stores
// go to the activation buffers, the DMA to the filter ring: disjoint by construction.  Ordering against the readers is what it was: the
// explicit `s_waitcnt lgkmcnt(0)` + s_barrier that ends every k-tile.
__device__ __forceinline__ void c3_lds_store_b64(unsigned* dst, const uint2& v) {
    typedef __attribute__((address_space(3))) unsigned* lds_u32;
    const unsigned addr = (unsigned)(size_t)(lds_u32)dst;
    const unsigned long long d = ((unsigned long long)v.y << 32) | v.x;
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(d) : "memory");
}

template <int BM, int BN, bool F16, int PWM = 0, bool ONEP = false>
__global__ __launch_bounds__(256, 2) void conv3x3_kernel(const conv3_args p) {
    constexpr bool PW = PWM != 0;
    static_assert(!PW || (F16 && BM == 128), "pre-split weights: the fp16-pair form on 128-pixel tiles");
    static_assert(!ONEP || PWM == 2, "the one-product form exists for the pipelined pre-split kernels");
    constexpr int NT = 256, SKH = 24;
    constexpr int NSB = (PWM == 2 && BN == 64) ? VBG_C3_NSB64 : 4, DPF = NSB - 1;      // PW: stages of the filter ring / k-tiles in flight
    constexpr int BTILE = 64 * BN;                             // PW: bytes of one k-tile of the filter (2 planes x BN rows x 32 B)
    constexpr int NDB = BN / 64;                               // PW: 1 KiB DMA units per wave and k-tile
    constexpr int NDBE = (ONEP && NDB == 2) ? 1 : NDB;         // ... that are issued (ONEP at 128 filters: the hi plane = units 0-3 only)
    constexpr int TNF = BN / 64;                               // 32-column fragments per wave
    constexpr int NBI = BN * 4 / NT;                           // filter float4s per thread and tile
    constexpr int NPL = F16 ? 2 : 3;                           // planes per operand tile
    constexpr int NQ = ONEP ? 1 : NPL;                         // planes that are read
    constexpr int NAI = BM * 4 / NT;                           // activation float4s per thread and super-tile
    constexpr int TM = BM / 64;                                // 32-row fragments per wave (waves 2 x 2: BM / 2 pixels x 64 filters each)
    constexpr int AROWS = BM + BM / 8;                         // + a zero pixel on either side of each image row (W >= 16)
    constexpr int PA = AROWS * SKH / 2, PB = BN * SKH / 2;     // one bf16 plane (dwords)
    constexpr int ASZ = NPL * PA, BSZ = NPL * PB;
    constexpr int CTS = BN + 4;
    constexpr int LOOPSZ = PW ? 2 * ASZ + NSB * BTILE / 4 : 2 * (ASZ + BSZ);
    constexpr int SMEM = LOOPSZ > BM * CTS ? LOOPSZ : BM * CTS;                        // 76 KB at BM = 128, three planes: two workgroups per CU
    // (the ONE LDS object of the kernel: with a second one the LDS lowering tags every access with alias scopes, and the compiler then
    //  puts `s_waitcnt vmcnt(0)` in front of every LDS read that follows an LDS-DMA into the same object -- the ring would never run
    //  ahead.  The split form's arrival ticket lives in the last word.)
    __shared__ __attribute__((aligned(16))) unsigned smem[SMEM + 4];
    unsigned* const As = smem;
    unsigned* const Bs = smem + 2 * ASZ;

    const int tid = threadIdx.x;
    // F16 form on an operand outside fp16's range (a gradient): X is multiplied by a power of two derived from its largest magnitude
    // on the way into LDS and the result by the inverse -- exact, and the pieces then sit in the middle of fp16's range
    float2 a_sc = make_float2(1.f, 1.f);
    if constexpr (F16) { if (p.a_amax) a_sc = c3_pow2_scale(c3_amax_read(p.a_amax)); }
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
    const int roi = p.roi;
    const int nb = roi ? (int)tile_m * (BM / 64) : m0 / HW;    // the tile lies inside one image (H*W % 128 == 0); ROI: its first image
    const int p0 = roi ? 0 : m0 - nb * HW;
    const float* const a_img = p.X + (long long)nb * HW * Cs;
    const int kc = (tid & 3) * 4;                              // this thread's 4 channels of a 16-channel chunk
    const int psh = roi ? 31 : wsh;                            // (ROI: no explicit zero pixels between the rows)
    int a_y[NAI], a_x[NAI], a_lrow[NAI], a_off[NAI];
    unsigned avo[NAI], bvo[NBI];
#pragma unroll
    for (int i = 0; i < NAI; ++i) {
        const int r = (tid + i * NT) >> 2;
        const int pix = p0 + r;
        a_x[i] = roi ? (r & 7) : (pix & (W - 1));
        a_y[i] = roi ? ((r >> 3) & 7) : (pix >> wsh);
        a_off[i] = roi ? ((r >> 6) * 49 + ((nb + (r >> 6) < roi && (r & 7) < 7 && ((r >> 3) & 7) < 7) ? 0 : (1 << 24))) : 0;   // image of the slot (or: no such pixel)
        a_lrow[i] = r + 1 + 2 * (r >> psh);
    }
#pragma unroll
    for (int i = 0; i < NBI; ++i) {
        const int r = (tid + i * NT) >> 2;
        bvo[i] = (n0 + r < N) ? (unsigned)((r * K + kc) * 4) : C3_INVALID;
    }
    // PW: unit u = wave + 4 i of a k-tile's filter block (1 KiB: plane u / (BN / 32), row block u % (BN / 32)); lane l moves bytes
    // [16 l, 16 l + 16) of the unit, source and LDS image alike
    typedef __attribute__((address_space(3))) void* c3_lds_ptr;
    unsigned char* const ring = reinterpret_cast<unsigned char*>(smem + 2 * ASZ);
    unsigned dvo[NDB];
    int dlds[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i) {
        dvo[i] = (unsigned)(((tid >> 6) + 4 * i) * 1024 + (tid & 63) * 16);
        dlds[i] = __builtin_amdgcn_readfirstlane(((tid >> 6) + 4 * i) * 1024);
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
            avo[i] = ((unsigned)sy < (unsigned)H && a_off[i] < (1 << 24)) ? (unsigned)(((a_off[i] + sy * W + a_x[i]) * Cs + kc) * 4) : C3_INVALID;
        }
        const int sy = h_y + kh - 1;
        hvo = (wide && tid < 8 && (unsigned)sy < (unsigned)H && (unsigned)h_x < (unsigned)W) ? (unsigned)(((sy * W + h_x) * Cs + kc) * 4) : C3_INVALID;
    };
    const float* const wbase = p.Wt + (long long)n0 * K;
    float4 ra[NAI], rb[NBI];
    // Split form (few pixel tiles: the late stages of the trunk): gridDim.z blocks share a tile, each reducing over one filter row
    // (ksplit = 3) and / or one group of channels; they meet in the epilogue (slabs + arrival ticket, below)
    const int nz = (int)gridDim.z, zz = (int)blockIdx.z;
    const int kh0 = p.ksplit == 3 ? zz % 3 : 0;
    const int cw = Cs / p.csplit, cbeg = (p.ksplit == 3 ? zz / 3 : zz) * cw, cend = cbeg + cw;
    int a_kh = kh0, a_c0 = cbeg;               // next activation super-tile (filter row, channel chunk) to load
    int b_kh = kh0, b_c0 = cbeg, b_kw = 0;     // next weight tile to load: order (kh, chunk, kw)
    auto load_a = [&]() {
        const __amdgpu_buffer_rsrc_t r = c3_rsrc(a_img + a_c0);
#pragma unroll
        for (int i = 0; i < NAI; ++i) ra[i] = c3_load(r, avo[i]);
        if (wide) rh = c3_load(r, hvo);
        a_c0 += 16;
        if (a_c0 >= cend) { a_c0 = cbeg; ++a_kh; set_a(a_kh); }
    };
    auto load_b = [&]() {
        const __amdgpu_buffer_rsrc_t r = c3_rsrc(wbase + (b_kh * 3 + b_kw) * Cs + b_c0);
#pragma unroll
        for (int i = 0; i < NBI; ++i) rb[i] = c3_load(r, bvo[i]);
        if (++b_kw == 3) {
            b_kw = 0; b_c0 += 16;
            if (b_c0 >= cend) { b_c0 = cbeg; ++b_kh; }
        }
    };
    // PW: the DMA of the next filter k-tile in walking order (kh, chunk, kw) into ring stage `d_stage`; past the block's last k-tile
    // every lane carries the invalid offset (zeros into a stage nobody reads, no memory traffic): the vmcnt arithmetic stays uniform
    const int nchk = Cs >> 4;
    int d_left = 0, d_stage = 0;               // k-tiles still to issue / the stage the next one goes to
    auto issue_b = [&]() {
        const unsigned inv = d_left > 0 ? 0u : C3_INVALID;
        const unsigned short* src = p.Wp + (size_t)(((int)tile_n * 9 + b_kh * 3 + b_kw) * nchk + (b_c0 >> 4)) * (BTILE / 2);
        const __amdgpu_buffer_rsrc_t r = c3_rsrc(reinterpret_cast<const float*>(src));
#pragma unroll
        for (int i = 0; i < NDBE; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (c3_lds_ptr)(ring + d_stage * BTILE + dlds[i]), 16, (int)(dvo[i] | inv), 0, 0, 0);
        --d_left;
        d_stage = d_stage + 1 == NSB ? 0 : d_stage + 1;
        if (++b_kw == 3) {
            b_kw = 0; b_c0 += 16;
            if (b_c0 >= cend) { b_c0 = cbeg; ++b_kh; }
        }
    };
    auto store4 = [&](unsigned* dst, int PL, int row, const float4& v) {
        const int o = row * (SKH / 2) + kc / 2;
        if constexpr (F16) {
            uint2 h, l;
            c3_split2(v.x, v.y, h.x, l.x);
            c3_split2(v.z, v.w, h.y, l.y);
            *reinterpret_cast<uint2*>(&dst[o]) = h;
            *reinterpret_cast<uint2*>(&dst[o + PL]) = l;
        } else {
            uint2 h, m, l;
            c3_split3(v.x, v.y, h.x, m.x, l.x);
            c3_split3(v.z, v.w, h.y, m.y, l.y);
            *reinterpret_cast<uint2*>(&dst[o]) = h;
            *reinterpret_cast<uint2*>(&dst[o + PL]) = m;
            *reinterpret_cast<uint2*>(&dst[o + 2 * PL]) = l;
        }
    };
    auto scaled = [&](const float4& v) { return make_float4(v.x * a_sc.x, v.y * a_sc.x, v.z * a_sc.x, v.w * a_sc.x); };
    auto store_a = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NAI; ++i) store4(As + buf * ASZ, PA, a_lrow[i], F16 ? scaled(ra[i]) : ra[i]);
        if (wide && tid < 8) store4(As + buf * ASZ, PA, h_row, F16 ? scaled(rh) : rh);
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NBI; ++i) store4(Bs + buf * BSZ, PB, (tid + i * NT) >> 2, rb[i]);
    };

    // ---------------- main loop ---------------------------------------------------------------
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lk = lane >> 5;
    f32x16 acc[TM][TNF], acx[F16 ? TM : 1][TNF];               // (F16: the cross products hi lo + lo hi, scaled by 2^11)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TNF; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][j][r] = 0.f;
                if constexpr (F16) acx[i][j][r] = 0.f;
            }
    int arow[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * (BM / 2) + i * 32 + lr;
        arow[i] = r + 2 * (r >> psh);                         // LDS row of pixel x - 1 (tap kw adds kw)
    }
    const int brow = wn * (BN / 2) + lr;

    using yes_t = std::integral_constant<bool, true>;
    using no_t = std::integral_constant<bool, false>;
    // one (tap, chunk) k-tile: fragments of this tile from LDS -> loads of the next tile -> first third of the piece products (covers
    // the load latency) -> the rest of the products with the split + LDS writes of the loaded tile in their gaps -> barrier
    auto k_tile = [&](auto loada_tag, auto more_tag, int abuf, int bbuf, int kw) {
        constexpr bool LOADA = decltype(loada_tag)::value, MORE = decltype(more_tag)::value;
        const c3_u32x4* as = reinterpret_cast<const c3_u32x4*>(As + abuf * ASZ);
        const c3_u32x4* bs = reinterpret_cast<const c3_u32x4*>(Bs + bbuf * BSZ);
        c3_u32x4 fa[NPL][TM], fb[NPL][TNF];
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[q][i] = as[q * (PA / 4) + (arow[i] + kw) * (SKH / 8) + lk];
#pragma unroll
            for (int j = 0; j < TNF; ++j) {
                if constexpr (PW) fb[q][j] = *reinterpret_cast<const c3_u32x4*>(ring + bbuf * BTILE + q * (BTILE / 2) + (wn * TNF + j) * 1024 + lk * 512 + lr * 16);
                else fb[q][j] = bs[q * (PB / 4) + (brow + j * 32) * (SKH / 8) + lk];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PW) {
            if constexpr (LOADA) load_a();     // (in front of the DMA: waiting for these registers must not wait for the DMA behind them)
            issue_b();
        } else {
            if constexpr (MORE) load_b();
            if constexpr (LOADA) load_a();
        }
        __builtin_amdgcn_sched_barrier(0);
        // piece products, smallest first: (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi); F16: (lo,hi) (hi,lo) -> cross sums, (hi,hi)
        constexpr int qa[6] = {F16 ? 1 : 2, 0, 1, 1, 0, 0}, qb[6] = {0, F16 ? 1 : 2, F16 ? 0 : 1, 0, 1, 0};
        auto mma_range = [&](auto t0_tag, auto t1_tag) {
#pragma unroll
            for (int t = decltype(t0_tag)::value; t < decltype(t1_tag)::value; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int n = 0; n < TNF; ++n) {
                        if constexpr (F16) {
                            f32x16& d = t < 2 ? acx[i][n] : acc[i][n];
                            d = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(c3_f16x8, fa[t == 0 ? 1 : 0][i]),
                                                                       __builtin_bit_cast(c3_f16x8, fb[t == 1 ? 1 : 0][n]), d, 0, 0, 0);
                        } else {
                            acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(c3_bf16x8, fa[qa[t]][i]),
                                                                                __builtin_bit_cast(c3_bf16x8, fb[qb[t]][n]), acc[i][n], 0, 0, 0);
                        }
                    }
        };
        using i0 = std::integral_constant<int, 0>;
        using ih = std::integral_constant<int, F16 ? 1 : 2>;
        using i1 = std::integral_constant<int, F16 ? 3 : 6>;
        if constexpr (MORE && PW) {
            mma_range(i0{}, ih{});
            __builtin_amdgcn_sched_barrier(0);
            mma_range(ih{}, i1{});
            if constexpr (LOADA) {
                store_a(abuf ^ 1);
                c3_pipe<0, 2 * TNF * TM, NAI * 10, NAI * 2>::run();
            }
            __builtin_amdgcn_sched_barrier(0);
            // this wave's share of the NEXT k-tile has landed (the DPF - 1 tiles behind it stay in flight), then everybody's
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DPF - 1) * NDB) : "memory");
            __syncthreads();
        } else if constexpr (MORE) {
            mma_range(i0{}, ih{});
            __builtin_amdgcn_sched_barrier(0);
            mma_range(ih{}, i1{});
            store_b(bbuf ^ 1);
            if constexpr (LOADA) store_a(abuf ^ 1);
            constexpr int NL = LOADA ? NBI + NAI : NBI;
            if constexpr (F16) c3_pipe<0, 2 * TNF * TM, NL * 10, NL * 2>::run();
            else c3_pipe<0, 4 * TNF * TM, NL * 22, NL * 3>::run();
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        } else {
            mma_range(i0{}, i1{});
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // the zero pixels beside the image rows are written once; the loop only ever stores the pixel rows
    for (int e = tid; e < 2 * ASZ / 4; e += NT) reinterpret_cast<uint4*>(As)[e] = make_uint4(0u, 0u, 0u, 0u);
    set_a(kh0);
    const int nsup = (p.ksplit == 3 ? 1 : 3) * cw / 16;
    if constexpr (PWM == 2) {
        // Ring of NSB stages (4); the DMA of k-tile t + NSB is issued at the END of k-tile t into the
        // stage of k-tile t itself (its fragments were read during k-tile t - 1), and must have landed by the end of k-tile t + NSB - 2:
        // NSB - 2 k-tiles of lead.
        // What hipcc adds on its own (measured in the ISA): nothing in front of an LDS READ that follows an LDS-DMA as long as the
        // kernel has ONE LDS object, but `s_waitcnt vmcnt` for the most recent LDS-DMA in front of every LDS WRITE -- so the
        // activation super-tile is split in registers in the gaps of the MFMAs and WRITTEN at the end of its k-tile, in front of that
        // k-tile's DMA issue, where the most recent DMA is a whole k-tile old.  No fence-carrying __syncthreads() in the loop (its
        // release fence drains vmcnt to 0): counted waits and bare barriers.
        d_left = 3 * nsup;
        load_a();
        __syncthreads();                       // (the zero fill above)
        store_a(0);
#pragma unroll
        for (int d = 0; d < NSB; ++d) issue_b();
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NSB - 2) * NDBE) : "memory");    // k-tiles 0 and 1 have landed
        __builtin_amdgcn_s_barrier();
        c3_u32x4 fA[2][NPL][TM], fB[2][NPL][TNF];
        auto read_frags = [&](auto par_tag, int abuf, int stage, int kw) {
            constexpr int P = decltype(par_tag)::value;
            const c3_u32x4* as = reinterpret_cast<const c3_u32x4*>(As + abuf * ASZ);
            const unsigned char* bsb = ring + stage * BTILE + (wn * TNF) * 1024 + lk * 512 + lr * 16;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fA[P][q][i] = as[q * (PA / 4) + (arow[i] + kw) * (SKH / 8) + lk];
#pragma unroll
                for (int j = 0; j < TNF; ++j) fB[P][q][j] = *reinterpret_cast<const c3_u32x4*>(bsb + q * (BTILE / 2) + j * 1024);
            }
        };
        auto mma_set = [&](auto par_tag) {
            constexpr int P = decltype(par_tag)::value;
#pragma unroll
            for (int t = ONEP ? 2 : 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int n = 0; n < TNF; ++n) {
                        f32x16& d = t < 2 ? acx[i][n] : acc[i][n];
                        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(c3_f16x8, fA[P][t == 0 ? 1 : 0][i]),
                                                                   __builtin_bit_cast(c3_f16x8, fB[P][t == 1 ? 1 : 0][n]), d, 0, 0, 0);
                    }
        };
        // the activation super-tile in two steps: pieces into registers (VALU, in the MFMA gaps), then the LDS writes
        uint2 sh[NAI + 1], sl[NAI + 1];
        auto split_a = [&]() {
#pragma unroll
            for (int i = 0; i < NAI; ++i) {
                const float4 v = scaled(ra[i]);
                c3_split2(v.x, v.y, sh[i].x, sl[i].x);
                c3_split2(v.z, v.w, sh[i].y, sl[i].y);
            }
            const float4 v = scaled(rh);
            c3_split2(v.x, v.y, sh[NAI].x, sl[NAI].x);
            c3_split2(v.z, v.w, sh[NAI].y, sl[NAI].y);
        };
        auto write_a = [&](int buf) {
            unsigned* dst = As + buf * ASZ;
#pragma unroll
            for (int i = 0; i < NAI; ++i) {
                const int o = a_lrow[i] * (SKH / 2) + kc / 2;
#if VBG_C3_ASMW
                c3_lds_store_b64(&dst[o], sh[i]);
                if constexpr (!ONEP) c3_lds_store_b64(&dst[o + PA], sl[i]);
#else
                *reinterpret_cast<uint2*>(&dst[o]) = sh[i];
                if constexpr (!ONEP) *reinterpret_cast<uint2*>(&dst[o + PA]) = sl[i];
#endif
            }
            if (wide && tid < 8) {
                const int o = h_row * (SKH / 2) + kc / 2;
#if VBG_C3_ASMW
                c3_lds_store_b64(&dst[o], sh[NAI]);
                if constexpr (!ONEP) c3_lds_store_b64(&dst[o + PA], sl[NAI]);
#else
                *reinterpret_cast<uint2*>(&dst[o]) = sh[NAI];
                if constexpr (!ONEP) *reinterpret_cast<uint2*>(&dst[o + PA]) = sl[NAI];
#endif
            }
        };
        using p0 = std::integral_constant<int, 0>;
        using p1 = std::integral_constant<int, 1>;
        int bst = 0;                           // ring stage of the current k-tile
        // one k-tile: P = register set holding its fragments, KW = its tap column; `s` its super-tile
        auto tile = [&](auto par_tag, auto kw_tag, int s) {
            constexpr int P = decltype(par_tag)::value, KW = decltype(kw_tag)::value;
            const int nst = bst + 1 == NSB ? 0 : bst + 1;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (P == 0) read_frags(p1{}, KW < 2 ? (s & 1) : ((s + 1) & 1), nst, KW < 2 ? KW + 1 : 0);
            else read_frags(p0{}, KW < 2 ? (s & 1) : ((s + 1) & 1), nst, KW < 2 ? KW + 1 : 0);
            if constexpr (P == 0) mma_set(p0{}); else mma_set(p1{});
            if constexpr (KW == 1) split_a();
            constexpr int NM = (ONEP ? 1 : 3) * TM * TNF, NRD = NQ * (TM + TNF);
            c3_pipe2<0, NM, NRD, KW == 1 ? (NAI + 1) * (ONEP ? 6 : 10) : 0, 0>::run();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (KW == 1) write_a((s + 1) & 1);   // super-tile s + 1: read from k-tile (s, 2) on
            // (activation loads in FRONT of the DMA: loads return in order, and waiting for these registers in the next k-tile must not
            //  mean waiting for the DMA issued behind them)
            if constexpr (KW == 0) load_a();                // super-tile s + 1 (past the end: zeros or rows nobody uses)
            issue_b();                                      // k-tile t + 4 into this k-tile's stage
            if constexpr (KW == 0) {
                // k-tile t + 2 has landed: t + 3, t + 4 and the activation loads just issued may stay in flight
                if (wide) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NSB - 2) * NDBE + NAI + 1) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NSB - 2) * NDBE + NAI) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NSB - 2) * NDBE) : "memory");
            }
            __builtin_amdgcn_s_barrier();
            bst = nst;
        };
        read_frags(p0{}, 0, 0, 0);
        // Round 6 -- the one place where the ring's rule ("a stage is refilled only after a barrier behind its last reader") did not hold:
        // k-tile 0 is read HERE, behind the prologue's barrier, and the first k-tile of the loop ends with the DMA of k-tile 4 into the
        // same stage 0 -- with no barrier in between.  A wave that fell one k-tile behind its siblings right after the prologue barrier (a
        // 64-filter k-tile is six MFMAs: ~300 clocks) read k-tile 4's filter slice as k-tile 0: one wave's 64 x 32 block of ONE tile off by one
        // k-tile's contribution (1/6 of its magnitude), seen in 5-12 % of cfg2 steps once three streams shared the chip -- the "conv
        // weight-gradient stream" outlier of profiles/r05_stream_race.txt (tools/stream_race_check.py --trace named the tensor, the wave
        // and the size).  Every wave now holds k-tile 0's fragments in registers before anybody may refill the stage.
        if (!p.dbg_no_ring_guard) {                            // (uniform: a kernel argument)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        using k0 = std::integral_constant<int, 0>;
        using k1 = std::integral_constant<int, 1>;
        using k2 = std::integral_constant<int, 2>;
        for (int s = 0; s < nsup; s += 2) {
            tile(p0{}, k0{}, s); tile(p1{}, k1{}, s); tile(p0{}, k2{}, s);
            if (s + 1 < nsup) { tile(p1{}, k0{}, s + 1); tile(p0{}, k1{}, s + 1); tile(p1{}, k2{}, s + 1); }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the zero-writing DMAs past the end must not land in the staged tile
    } else if constexpr (PW) {
        d_left = 3 * nsup;
        load_a();
        __syncthreads();                       // (the zero fill above)
#pragma unroll
        for (int d = 0; d < DPF; ++d) issue_b();
        store_a(0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DPF - 1) * NDB) : "memory");       // k-tile 0 has landed
        __syncthreads();
        int bst = 0;
        auto nx = [&]() { const int c = bst; bst = bst + 1 == NSB ? 0 : bst + 1; return c; };
        for (int s = 0; s + 1 < nsup; ++s) {
            k_tile(no_t{}, yes_t{}, s & 1, nx(), 0);
            k_tile(no_t{}, yes_t{}, s & 1, nx(), 1);
            k_tile(yes_t{}, yes_t{}, s & 1, nx(), 2);
        }
        {
            const int s = nsup - 1;
            k_tile(no_t{}, yes_t{}, s & 1, nx(), 0);
            k_tile(no_t{}, yes_t{}, s & 1, nx(), 1);
            k_tile(no_t{}, no_t{}, s & 1, nx(), 2);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the zero-writing DMAs past the end must not land in the staged tile
    } else {
    load_a();
    load_b();
    __syncthreads();
    store_a(0);
    store_b(0);
    __syncthreads();
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
    }

    // ---------------- epilogue: staged through LDS, float4 row pieces, optional bias / accumulate / BatchNorm statistics ---------
    const float* bias = p.bias;
    const int accumulate = p.accumulate;
    __syncthreads();
    float* const Ct = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < TNF; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Ct[(wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * CTS + wn * (BN / 2) + j * 32 + lr] =
                    ONEP ? acc[i][j][r] * a_sc.y : (F16 ? (acc[i][j][r] + acx[i][j][r] * (1.f / 2048.f)) * a_sc.y : acc[i][j][r]);
    __syncthreads();
    constexpr int QN = BN / 4;
    // Split form: the block's partial tile goes to its slab (write-through stores), every wave drains its stores, and one relaxed
    // agent-scope ticket per block decides who finishes the tile: the LAST arriver (one agent-scope acquire) adds the gridDim.z slabs in
    // block order -- its own included, so the sum does not depend on who came last -- and runs the epilogue.  Nobody waits.
    const float* part = nullptr;
    if (nz > 1) {
        unsigned& ticket_s = smem[SMEM];
        const size_t tile = (size_t)tile_n * gx + tile_m;
        float* const mine = p.slab + (tile * nz + zz) * (size_t)(BM * BN);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(mine, 0, BM * BN * 4, 0x00020000);
#pragma unroll 4
        for (int q = 0; q < BM * QN / NT; ++q) {
            const int idx = tid + q * NT;
            const int row = idx / QN, c = (idx % QN) * 4;
            const c3_u32x4 v = *reinterpret_cast<const c3_u32x4*>(&Ct[row * CTS + c]);
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, (row * BN + c) * 4, 0, 16 /* sc1: write-through */);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) ticket_s = __hip_atomic_fetch_add(p.tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (ticket_s != (unsigned)(nz - 1)) return;            // (uniform) somebody else finishes this tile
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(p.tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // ready for the next launch
        }
        __syncthreads();
        part = p.slab + tile * nz * (size_t)(BM * BN);
    }
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f), cq = cs;
#pragma unroll 4
    for (int q = 0; q < BM * QN / NT; ++q) {
        const int idx = tid + q * NT;
        const int row = idx / QN, c = (idx % QN) * 4;
        int gm = m0 + row;
        const int gn = n0 + c;
        if (gn >= N) continue;                                 // (N % 4 == 0: a float4 is inside or outside)
        if (roi) {                                             // slot -> compact row; the eighth row / column and images past the last do not exist
            const int im = nb + (row >> 6), sy = (row >> 3) & 7, sx = row & 7;
            if (im >= roi || sy == 7 || sx == 7) continue;
            gm = im * 49 + sy * 7 + sx;
        }
        float4 v;
        if (part) {
            v = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int b = 0; b < nz; ++b) {
                const float4 o = *reinterpret_cast<const float4*>(part + (size_t)b * (BM * BN) + row * BN + c);
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
        } else {
            v = *reinterpret_cast<const float4*>(&Ct[row * CTS + c]);
        }
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

// Filter planes of the PW kernels, for every 3x3 filter of a model in ONE launch (blockIdx.y = table entry).  Entry e describes one
// image: the filter itself (flip = 0: rows = output channels, reduction = input channels) or the filter of the input gradient
// (flip = 1: turned by 180 degrees, channel roles swapped -- rows = input channels, reduction = output channels; what conv3_wflip_kernel
// writes as fp32).  Image layout (BN = bn rows per filter tile): block ((tile * 9 + tap) * (K / 16) + chunk) of 64 * bn bytes =
// [plane: hi, lo][row block of 32][8-channel half][row][8 x fp16]; rows past `rows` are zeros.  One thread = one (row, half) group of 8
// elements: hi = fp16(w), lo = fp16((w - hi) * 2^11), both rounded to nearest (c3_split2).
__global__ __launch_bounds__(256) void conv3_wprep_kernel(const vbg_conv3_wprep_entry* __restrict__ tab) {
    const vbg_conv3_wprep_entry e = tab[blockIdx.y];
    const int bn = e.bn, rows = e.flip ? e.Cin : e.Cout, K = e.flip ? e.Cout : e.Cin;
    const int gpb = bn * 2;                                    // groups per filter block
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long nblocks = (long long)((rows + bn - 1) / bn) * 9 * (K / 16);
    if (g >= nblocks * gpb) return;
    const long long blk = g / gpb;
    const int in = (int)(g - blk * gpb);                       // = (rb * 2 + half) * 32 + row
    const int row32 = in & 31, half = (in >> 5) & 1, rb = in >> 6;
    const int chunk = (int)(blk % (K / 16));
    const int tap = (int)((blk / (K / 16)) % 9);
    const int tile = (int)(blk / ((long long)(K / 16) * 9));
    const int r = tile * bn + rb * 32 + row32;
    const int k0 = chunk * 16 + half * 8;
    float v[8];
    if (r < rows) {
        if (!e.flip) {
            const float4* src = reinterpret_cast<const float4*>(e.w + ((long long)r * 9 + tap) * e.Cin + k0);
            const float4 a = src[0], b = src[1];
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = e.w[((long long)(k0 + j) * 9 + (8 - tap)) * e.Cin + r];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
    }
    uint4 h, l;
    c3_split2(v[0], v[1], h.x, l.x);
    c3_split2(v[2], v[3], h.y, l.y);
    c3_split2(v[4], v[5], h.z, l.z);
    c3_split2(v[6], v[7], h.w, l.w);
    unsigned char* dst = reinterpret_cast<unsigned char*>(e.out) + blk * (64ll * bn) + (long long)rb * 1024 + half * 512 + row32 * 16;
    *reinterpret_cast<uint4*>(dst) = h;
    *reinterpret_cast<uint4*>(dst + 32 * bn) = l;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Weight gradient of the same convolution: dW[co][kh][kw][ci] += sum_p dY[p][co] * X[p + (kh-1) W + (kw-1)][ci].
// The reduction index is the pixel, i.e. the ROW index of both NHWC operands.  A k-tile is a chunk of 16 consecutive pixels of one
// image row; its dY rows [16][CO] and the X pixels the nine taps touch ([3 rows][18 pixels][CI], borders read 0) are split once
// and stored as they lie in memory ([pixel][channel] bf16, three planes).  The MFMA fragments -- 8 consecutive pixels of one channel --
// come out of that image through ds_read_b64_tr_b16 (a 16-lane group reads a [4 pixels][16 channels] block and receives it
// transposed), so nothing is transposed on the way into LDS, and a filter tap is a ROW offset (kh * 18 + kw) of the X image: one
// load and one split of X serve all nine taps, the dY fragments of a k-tile serve 54 MFMAs.  Per product the kernel moves 0.025 B
// from L2 (generic CONV_R gather of gemm.hip: 0.0625 B) and splits a quarter of the elements.  A workgroup owns a [CO x 9 x CI]
// block of dW (CO x CI = 128 x 32 or 64 x 64: 144 accumulator registers) and a strip of pixel chunks; strips meet in float atomics.
// Row strides of the LDS images are odd multiples of 64 bytes: the four pixel rows of a transposing read land on disjoint banks.
// ------------------------------------------------------------------------------------------------------------------------------
struct conv3w_args {
    const float* dY;       // [B, H, W, Cout]
    const float* X;        // [B, H, W, Cs]
    float* dW;             // [Cout, 3, 3, Cs], accumulated into with float atomics (slab == null) ...
    float* slab;           // ... or [strips][Cout, 3, 3, Cs]: every strip stores its partial block plainly, conv3_wgrad_reduce_kernel adds them up
    int H, W, Cs, Cout;
    int nchunks, per;      // 16-pixel chunks in total / per strip
    int roi;               // 1: H = W = 7 images stored compactly; a chunk is two rows of the image's 8 x 8 slot grid (see conv3x3_kernel)
    const unsigned* dy_amax;   // F16 form: bit patterns of max |dY| and max |X| (device words), the operands' power-of-two scales
    const unsigned* x_amax;
};

typedef short c3_v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) c3_v4s* c3_lds_v4s;
typedef __attribute__((address_space(3))) unsigned char* c3_lds_bytes;

// F16: the arithmetic form -- false: three bf16 pieces per operand, six piece products; true: two fp16 pieces (hi = fp16(v s), lo =
// fp16(v s - hi), round to nearest) of both operands after scaling each by the power of two s that brings its tensor's largest magnitude
// to [2^13, 2^14) (exact), three piece products (lo hi, hi lo, hi hi) into the ONE accumulator of a tap (nine taps x two accumulators
// would not fit the register file), the result scaled back in the epilogue.  Elements down to 2^-17 of the tensor's maximum keep 22
// significant bits (lo is a normal fp16 there); below that the absolute error is <= 2^-25 (2^-39 of the maximum).
// ONEP (`amp`): the hi pieces only, one product per tap (the lo planes are neither written nor read).
template <int WCO, bool F16, bool ROI = false, bool ONEP = false>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_kernel(const conv3w_args p) {
    static_assert(!ONEP || F16, "the one-product form is the fp16 form's hi x hi product");
    constexpr int NQ = ONEP ? 1 : (F16 ? 2 : 3);               // planes that are written and read
    constexpr int NT = 256, WCI = 4 / WCO, CO = 32 * WCO, CI = 32 * WCI;
    constexpr int NPL = F16 ? 2 : 3;
    // bytes of a pixel row of the dY / X image: an odd multiple of 64 B (16 banks), so that the four consecutive pixel rows a
    // transposing read touches -- 32 bytes each for either 16-channel half -- lie on eight disjoint 8-bank groups
    constexpr int RSA = CO * 2 + 64, RSB = CI == 32 ? 64 : CI * 2 + 64;
    constexpr int PA = 16 * RSA, PB = 54 * RSB;                // one plane
    constexpr int STAGE = NPL * (PA + PB);
    constexpr int QA = CO / 4, QB = CI / 4;                    // float4 per pixel
    constexpr int NDA = 16 * QA / NT, NXB = 54 * QB, NDB = (NXB + NT - 1) / NT;
    static_assert(16 * QA % NT == 0, "dY chunk divides over the threads");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = p.H, W = p.W, Cs = p.Cs, Cout = p.Cout;
    const int co0 = blockIdx.z * CO, ci0 = blockIdx.y * CI;
    const int c0 = blockIdx.x * p.per, c1 = min(p.nchunks, c0 + p.per);
    if (c0 >= c1) return;
    float2 sc_a = make_float2(1.f, 1.f), sc_b = sc_a;
    if constexpr (F16) {
        sc_a = c3_pow2_scale(c3_amax_read(p.dy_amax));
        sc_b = c3_pow2_scale(c3_amax_read(p.x_amax));
    }

    // ---- loader: per-thread constants, per-chunk scalars -------------------------------------------------------------------
    // ROI mode: the k-tile is 16 SLOTS = two rows of an image's 8 x 8 slot grid (slot x = 7 / row 7: no such pixel, read as 0), the X
    // image the 34 slots from (row - 1, x - 1) on at pitch 8 -- the zero slot x = 7 doubles as the padding left of x = 0 -- so a tap is
    // again a row offset, kh * 8 + kw.  49 of 64 reduction rows are real pixels.
    constexpr bool roi = ROI;
    constexpr int pitch = roi ? 8 : 18;
    unsigned avo[NDA], bvo[NDB];
    int alds[NDA], blds[NDB], b_r3[NDB], b_px[NDB], a_yy[NDA];
#pragma unroll
    for (int i = 0; i < NDA; ++i) {
        const int e = tid + i * NT, px = e / QA, c4 = e % QA;
        avo[i] = roi ? (unsigned)((((px >> 3) * 7 + (px & 7)) * Cout + co0 + 4 * c4) * 4) : (unsigned)((px * Cout + co0 + 4 * c4) * 4);
        a_yy[i] = (px & 7) == 7 ? 100 : (px >> 3);             // (ROI: row of the slot inside the chunk; 100: no such slot)
        alds[i] = px * RSA + c4 * 8;
    }
#pragma unroll
    for (int i = 0; i < NDB; ++i) {
        const int e = tid + i * NT;
        if constexpr (roi) {
            const int j = e / QB, c4 = e % QB, d = j - 9;       // slot j of the X image = slot d relative to the chunk's first
            const bool ok = e < NXB && j < 34;
            b_r3[i] = ok ? (d >> 3) : 1000000;                 // image row relative to the chunk's first (-2 .. 3)
            b_px[i] = d & 7;
            bvo[i] = ok ? (unsigned)(((((d >> 3) + 2) * 7 + (d & 7)) * Cs + ci0 + 4 * c4) * 4) : 0u;
            blds[i] = e < NXB ? j * RSB + c4 * 8 : 0;
        } else {
            const int r3 = e / (18 * QB), rem = e - r3 * (18 * QB), px = rem / QB, c4 = rem % QB;
            const bool ok = e < NXB;
            b_r3[i] = ok ? r3 : 1000000;                       // (never inside the image)
            b_px[i] = px;
            bvo[i] = (unsigned)(((r3 * W + px) * Cs + ci0 + 4 * c4) * 4);
            blds[i] = ok ? (r3 * 18 + px) * RSB + c4 * 8 : 0;
        }
    }
    // chunk c = pixels [16 c, 16 c + 16) of the flattened [B, H, W] index: (image n, row y, first column x0) walk in scalars
    int pix = c0 * 16;
    int n = pix / (H * W);
    int y = (pix - n * H * W) / W;
    int x0 = pix - (n * H + y) * W;
    int chunk = c0;
    float4 ra[NDA], rb[NDB];
    auto load_chunk = [&](unsigned inv) {
        if constexpr (roi) {
            const int img = chunk >> 2, cc = chunk & 3;         // four chunks (slot rows 0-1, 2-3, 4-5, 6-7) per image
            const __amdgpu_buffer_rsrc_t rA = c3_rsrc(p.dY + (long long)(img * 49 + 14 * cc) * Cout);
#pragma unroll
            for (int i = 0; i < NDA; ++i) ra[i] = c3_load(rA, (a_yy[i] + 2 * cc < 7) ? (avo[i] | inv) : C3_INVALID);
            // descriptor base: pixel (first row - 2, 0) of the image (lanes whose slot is no pixel carry the invalid offset)
            const __amdgpu_buffer_rsrc_t rB = c3_rsrc(p.X + (long long)(img * 49 + (2 * cc - 2) * 7) * Cs);
#pragma unroll
            for (int i = 0; i < NDB; ++i) {
                const bool ok = (unsigned)(2 * cc + b_r3[i]) < 7u && b_px[i] < 7;
                rb[i] = c3_load(rB, ok ? (bvo[i] | inv) : C3_INVALID);
            }
            ++chunk;
            return;
        }
        const __amdgpu_buffer_rsrc_t rA = c3_rsrc(p.dY + (long long)pix * Cout);
#pragma unroll
        for (int i = 0; i < NDA; ++i) ra[i] = c3_load(rA, avo[i] | inv);
        // descriptor base: pixel (y - 1, x0 - 1) of image n (lanes whose pixel lies outside the image carry the invalid offset)
        const __amdgpu_buffer_rsrc_t rB = c3_rsrc(p.X + ((long long)(n * H + y - 1) * W + (x0 - 1)) * Cs);
#pragma unroll
        for (int i = 0; i < NDB; ++i) {
            const bool ok = (unsigned)(y + b_r3[i] - 1) < (unsigned)H && (unsigned)(x0 + b_px[i] - 1) < (unsigned)W;
            rb[i] = c3_load(rB, ok ? (bvo[i] | inv) : C3_INVALID);
        }
        pix += 16; x0 += 16;
        if (x0 >= W) { x0 = 0; if (++y >= H) { y = 0; ++n; } }
    };
    auto store3 = [&](unsigned char* dst, int PL, const float4& v) {
        uint2 h, m, l;
        c3_split3(v.x, v.y, h.x, m.x, l.x);
        c3_split3(v.z, v.w, h.y, m.y, l.y);
        *reinterpret_cast<uint2*>(dst) = h;
        *reinterpret_cast<uint2*>(dst + PL) = m;
        *reinterpret_cast<uint2*>(dst + 2 * PL) = l;
    };
    auto store2 = [&](unsigned char* dst, int PL, const float4& v, float sc) {
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
        const f32x2_t v0 = {v.x * sc, v.y * sc}, v1 = {v.z * sc, v.w * sc};
        const f16x2_t h0 = __builtin_convertvector(v0, f16x2_t), h1 = __builtin_convertvector(v1, f16x2_t);
        const f16x2_t l0 = __builtin_convertvector(v0 - __builtin_convertvector(h0, f32x2_t), f16x2_t);
        const f16x2_t l1 = __builtin_convertvector(v1 - __builtin_convertvector(h1, f32x2_t), f16x2_t);
        *reinterpret_cast<uint2*>(dst) = make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1));
        if constexpr (!ONEP) *reinterpret_cast<uint2*>(dst + PL) = make_uint2(__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1));
    };
    auto store_chunk = [&](int stage) {
        unsigned char* sa = smem + stage * STAGE;
        unsigned char* sb = sa + NPL * PA;
#pragma unroll
        for (int i = 0; i < NDA; ++i) {
            if constexpr (F16) store2(sa + alds[i], PA, ra[i], sc_a.x); else store3(sa + alds[i], PA, ra[i]);
        }
#pragma unroll
        for (int i = 0; i < NDB; ++i)
            if (NXB % NT == 0 || i + 1 < NDB || tid + i * NT < NXB) {
                if constexpr (F16) store2(sb + blds[i], PB, rb[i], sc_b.x); else store3(sb + blds[i], PB, rb[i]);
            }
    };

    // ---- fragments -----------------------------------------------------------------------------------------------------------
    const int wco = wave / WCI, wci = wave % WCI;
    const int lr = lane & 31, lk = lane >> 5;
    const int i16 = lane & 15, tg = (lane >> 4) & 1;
    const int ta = (8 * lk + (i16 >> 2)) * RSA + (32 * wco + 16 * tg + 4 * (i16 & 3)) * 2;
    const int tb = (8 * lk + (i16 >> 2)) * RSB + (32 * wci + 16 * tg + 4 * (i16 & 3)) * 2;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    auto frag = [&](c3_lds_bytes base, int RS) {
        const c3_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((c3_lds_v4s)(base));
        const c3_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((c3_lds_v4s)(base + 4 * RS));
        const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
        return c3_u32x4{l2.x, l2.y, h2.x, h2.y};
    };

    load_chunk(0u);
    store_chunk(0);
    __syncthreads();
    const int nch = c1 - c0;
    for (int it = 0; it < nch; ++it) {
        const int stage = it & 1;
        load_chunk(it + 1 < nch ? 0u : C3_INVALID);             // (past the strip: every lane reads 0, no traffic; stored into the idle stage)
        __builtin_amdgcn_sched_barrier(0);
        c3_lds_bytes as = (c3_lds_bytes)(smem + stage * STAGE) + ta;
        c3_lds_bytes bs = (c3_lds_bytes)(smem + stage * STAGE + NPL * PA) + tb;
        c3_u32x4 fa[NPL];
#pragma unroll
        for (int q = 0; q < NQ; ++q) fa[q] = frag(as + q * PA, RSA);
        // piece products, smallest first; F16: (lo,hi) (hi,lo) (hi,hi)
        constexpr int qa[6] = {F16 ? 1 : 2, 0, F16 ? 0 : 1, 1, 0, 0}, qb[6] = {0, F16 ? 1 : 2, F16 ? 0 : 1, 0, 1, 0};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            c3_u32x4 fb[NPL];
#pragma unroll
            for (int q = 0; q < NQ; ++q) fb[q] = frag(bs + q * PB + ((tap / 3) * pitch + tap % 3) * RSB, RSB);
#pragma unroll
            for (int t = ONEP ? 2 : 0; t < (F16 ? 3 : 6); ++t) {
                if constexpr (F16)
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(c3_f16x8, fa[qa[t]]), __builtin_bit_cast(c3_f16x8, fb[qb[t]]),
                                                                      acc[tap], 0, 0, 0);
                else
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(c3_bf16x8, fa[qa[t]]), __builtin_bit_cast(c3_bf16x8, fb[qb[t]]),
                                                                       acc[tap], 0, 0, 0);
            }
        }
        store_chunk(stage ^ 1);
        __syncthreads();
    }
    // ---- epilogue (lanes = 32 consecutive ci: 128-byte rows): the strip's partial block goes to its slab as plain stores (float
    //      atomics run at the L2's atomic rate: ~70 us for the 9.4 M partials of a 256 x 256 filter), or straight into dW ---------
    const float unscale = sc_a.y * sc_b.y;                     // (1 in the bf16 form; a power of two: exact)
    if (p.slab) {
        float* const out = p.slab + (long long)blockIdx.x * Cout * 9 * Cs;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + 32 * wco + (r & 3) + 8 * (r >> 2) + 4 * lk;
                out[((long long)co * 9 + tap) * Cs + ci0 + 32 * wci + lr] = F16 ? acc[tap][r] * unscale : acc[tap][r];
            }
        return;
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + 32 * wco + (r & 3) + 8 * (r >> 2) + 4 * lk;
            unsafeAtomicAdd(p.dW + ((long long)co * 9 + tap) * Cs + ci0 + 32 * wci + lr, F16 ? acc[tap][r] * unscale : acc[tap][r]);
        }
}

// dW[i] += sum over the strips' slabs (fixed order: deterministic)
__global__ void conv3_wgrad_reduce_kernel(const float4* __restrict__ slab, int nsplit, long long n4, float4* __restrict__ dW) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 s = dW[i];
    for (int k = 0; k < nsplit; ++k) {
        const float4 v = slab[k * n4 + i];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    dW[i] = s;
}

// the same for many strips over a small dW (the [64 x 9 x 64] blocks of the first trunk stage: 256 strips, 9216 float4 -- 36 blocks of the
// kernel above, every thread 256 dependent-in-order loads: 64 us): a block is 64 float4 columns x 16 strip groups (one wave each), every
// thread has its nsplit / 16 loads in flight together, the groups meet in LDS and are added in group order -- a fixed order as well
__global__ __launch_bounds__(1024) void conv3_wgrad_reduce_par_kernel(const float4* __restrict__ slab, int nsplit, long long n4,
                                                                      float4* __restrict__ dW) {
    __shared__ float4 sh[16][64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + c;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) {
        int k = g;
        for (; k + 48 < nsplit; k += 64) {
            float4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = slab[(long long)(k + 16 * j) * n4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) { a.x += v[j].x; a.y += v[j].y; a.z += v[j].z; a.w += v[j].w; }
        }
        for (; k < nsplit; k += 16) {
            const float4 v = slab[(long long)k * n4 + i];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    }
    sh[g][c] = a;
    __syncthreads();
    if (g == 0 && i < n4) {
        float4 s = dW[i];
#pragma unroll
        for (int j = 0; j < 16; ++j) { s.x += sh[j][c].x; s.y += sh[j][c].y; s.z += sh[j][c].z; s.w += sh[j][c].w; }
        dW[i] = s;
    }
}

}  // namespace vbg

extern "C" int vbg_conv3x3_wgrad_strips(int B, int H, int W, int Cs, int Cout) {
    const long long nchunks = (H == 7 && W == 7) ? (long long)B * 4 : (long long)B * H * W / 16;
    const bool wide = Cout % 128 == 0;
    const long long tiles = wide ? (long long)(Cout / 128) * (Cs / 32) : (long long)(Cout / 64) * (Cs / 64);
    if (tiles <= 0 || nchunks <= 0) return 0;
    // workgroups: whole rounds of the chip -- two per CU when a strip then still holds >= 256 k-tiles (the wide convolutions at
    // 1/4 resolution: 717 vs 767 us), else one per CU (fewer, longer strips and half the slab traffic: 72 vs 87 us at 128 x 128
    // channels, 64 x 64 pixels); measured with tools/conv3w_sweep.py.  VBG_CONV3W_BLOCKS / VBG_CONV3W_SLAB_MB override (tuning).
    static const long long blocks_env = getenv("VBG_CONV3W_BLOCKS") ? atoll(getenv("VBG_CONV3W_BLOCKS")) : 0;
    static const long long slab_mb = getenv("VBG_CONV3W_SLAB_MB") ? atoll(getenv("VBG_CONV3W_SLAB_MB")) : 96;
    long long blocks = blocks_env;
    if (blocks <= 0) blocks = (nchunks / ((512 + tiles - 1) / tiles) >= 256) ? 512 : 256;
    long long nsplit = (blocks + tiles - 1) / tiles;
    const long long cap = (slab_mb << 20) / ((long long)Cout * 9 * Cs * 4);    // bound on the slab scratch
    if (nsplit > cap) nsplit = cap < 1 ? 1 : cap;
    if (nsplit > nchunks / 8) nsplit = nchunks / 8 < 1 ? 1 : nchunks / 8;      // at least 8 k-tiles per strip
    const long long per = (nchunks + nsplit - 1) / nsplit;
    return (int)((nchunks + per - 1) / per);
}

extern "C" int vbg_conv3x3_wgrad(const float* dy, const float* x, float* dw, float* slab, int B, int H, int W, int Cs, int Cout, int form,
                                 const unsigned* dy_amax, const unsigned* x_amax, void* stream) {
    const bool roi = H == 7 && W == 7;                          // [B, 7, 7, C] region maps
    VBG_CHECK_ARG(dy && x && dw && B > 0 && H > 0 && (roi || (W >= 16 && W % 16 == 0)));
    VBG_CHECK_ARG(form == 0 || ((form == 1 || form == 2) && dy_amax && x_amax));          // (2: `amp`, the hi x hi product only)
    VBG_CHECK_ARG(Cs % 32 == 0 && Cout % 64 == 0);
    VBG_CHECK_ARG((((uintptr_t)x) & 15) == 0 && (((uintptr_t)dy) & 15) == 0);
    VBG_CHECK_ARG((long long)(3 * W + 18) * Cs < (1ll << 28) && (long long)16 * Cout < (1ll << 28));
    const long long M = (long long)B * H * W;
    VBG_CHECK_ARG(M < (1ll << 31));
    vbg::conv3w_args a;
    a.dY = dy; a.X = x; a.dW = dw; a.slab = slab; a.H = H; a.W = W; a.Cs = Cs; a.Cout = Cout;
    a.dy_amax = dy_amax; a.x_amax = x_amax;
    a.roi = roi ? 1 : 0;
    a.nchunks = roi ? B * 4 : (int)(M / 16);
    const bool wide = Cout % 128 == 0;                         // [128 co x 32 ci] blocks, else [64 co x 64 ci]
    VBG_CHECK_ARG(wide || Cs % 64 == 0);
    const int nsplit = vbg_conv3x3_wgrad_strips(B, H, W, Cs, Cout);
    VBG_CHECK_ARG(nsplit >= 1 && (!slab || (((uintptr_t)slab) & 15) == 0) && (((uintptr_t)dw) & 15) == 0);
    a.per = (a.nchunks + nsplit - 1) / nsplit;
    if (roi) {
        VBG_CHECK_ARG(wide);
        if (form == 2) { VBG_LAUNCH((vbg::conv3x3_wgrad_kernel<4, true, true, true>), dim3(nsplit, Cs / 32, Cout / 128), dim3(256), 0, (hipStream_t)stream, a); }
        else if (form == 1) { VBG_LAUNCH((vbg::conv3x3_wgrad_kernel<4, true, true>), dim3(nsplit, Cs / 32, Cout / 128), dim3(256), 0, (hipStream_t)stream, a); }
        else { VBG_LAUNCH((vbg::conv3x3_wgrad_kernel<4, false, true>), dim3(nsplit, Cs / 32, Cout / 128), dim3(256), 0, (hipStream_t)stream, a); }
    } else if (wide) {
        if (form == 2) { VBG_LAUNCH((vbg::conv3x3_wgrad_kernel<4, true, false, true>), dim3(nsplit, Cs / 32, Cout / 128), dim3(256), 0, (hipStream_t)stream, a); }
        else if (form == 1) { VBG_LAUNCH((vbg::conv3x3_wgrad_kernel<4, true>), dim3(nsplit, Cs / 32, Cout / 128), dim3(256), 0, (hipStream_t)stream, a); }
        else { VBG_LAUNCH((vbg::conv3x3_wgrad_kernel<4, false>), dim3(nsplit, Cs / 32, Cout / 128), dim3(256), 0, (hipStream_t)stream, a); }
    } else {
        if (form == 2) { VBG_LAUNCH((vbg::conv3x3_wgrad_kernel<2, true, false, true>), dim3(nsplit, Cs / 64, Cout / 64), dim3(256), 0, (hipStream_t)stream, a); }
        else if (form == 1) { VBG_LAUNCH((vbg::conv3x3_wgrad_kernel<2, true>), dim3(nsplit, Cs / 64, Cout / 64), dim3(256), 0, (hipStream_t)stream, a); }
        else { VBG_LAUNCH((vbg::conv3x3_wgrad_kernel<2, false>), dim3(nsplit, Cs / 64, Cout / 64), dim3(256), 0, (hipStream_t)stream, a); }
    }
    if (slab) {
        const long long n4 = (long long)Cout * 9 * Cs / 4;
        // (many strips over a small dW: strips in parallel as well; VBG_CONV3W_REDUCE_PAR=0 keeps the one-thread-per-element kernel)
        static const int par_from = getenv("VBG_CONV3W_REDUCE_PAR") ? atoi(getenv("VBG_CONV3W_REDUCE_PAR")) : 32;
        if (par_from > 0 && nsplit >= par_from)
            hipLaunchKernelGGL(vbg::conv3_wgrad_reduce_par_kernel, dim3((unsigned)((n4 + 63) / 64)), dim3(1024), 0, (hipStream_t)stream,
                               reinterpret_cast<const float4*>(slab), nsplit, n4, reinterpret_cast<float4*>(dw));
        else
            hipLaunchKernelGGL(vbg::conv3_wgrad_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                               reinterpret_cast<const float4*>(slab), nsplit, n4, reinterpret_cast<float4*>(dw));
    }
    VBG_LAUNCH_RET();
}

// blocks per 128 x 128 tile the forward / input-gradient kernel wants for this shape (1: no split).  The late stages of the trunk have
// 64-128 tiles whose reduction is 2304-4608 long: one block per tile leaves most of the chip idle for the ~75 us a lone block needs
// for its k-loop.  Three blocks per tile (one filter row each) or four (a quarter of the channels each) put >= 256 blocks on the
// chip.  VBG_CONV3_SPLIT_Z overrides (tuning; multiples of 3 split the filter rows).
extern "C" int vbg_conv3x3_split(int B, int H, int W, int Cs, int N) {
    static const int forced = getenv("VBG_CONV3_SPLIT_Z") ? atoi(getenv("VBG_CONV3_SPLIT_Z")) : 0;
    if (W < 16 || (W & (W - 1)) != 0 || ((long long)H * W) % 128 != 0 || N % 128 != 0 || Cs % 16 != 0 || W >= 128) return 1;
    const long long tiles = ((long long)B * H * W / 128) * (N / 128);
    // (round 6: down to 4 tiles -- a single document's last stage is 8 tiles of 4608-long reductions: 92 us on the generic 64 x 64 tiles, 32 workgroups)
    if (tiles >= 240 || tiles < 4) return 1;
    int nz = tiles * 3 >= 256 ? 3 : tiles * 4 >= 256 ? 4 : tiles * 6 >= 256 ? 6 : 12;      // (measured: 512 channels at 16 x 16 pixels, 4: 63 us, 6: 66, 3: 71, 1: 137)
    if (tiles < 16) nz = 4;          // (one document: 8 tiles -- 4: 45 us, 6: 46, 12: 63, 1: 86; tools/conv3_small_sweep.py)
    if (forced > 0) nz = forced;
    const int cs = nz % 3 == 0 ? nz / 3 : nz;
    if (Cs % cs != 0 || (Cs / cs) % 16 != 0) return 1;
    return nz;
}

// rows per filter tile of the plane image of a filter with `rows` output rows (vbg_conv3x3_wprep / vbg_conv3x3_pw agree on it)
static int conv3_pw_bn(int rows) { return (rows % 128 != 0 && rows % 64 == 0) ? 64 : 128; }

extern "C" long long vbg_conv3x3_wprep_bytes(int Cout, int Cin, int flip, int bn_req) {
    const int rows = flip ? Cin : Cout, K = flip ? Cout : Cin;
    if (rows <= 0 || K <= 0 || K % 16 != 0 || !(bn_req == 0 || bn_req == 64 || bn_req == 128)) return 0;
    const int bn = bn_req ? bn_req : conv3_pw_bn(rows);
    return (long long)((rows + bn - 1) / bn) * 9 * (K / 16) * 64 * bn;
}

extern "C" int vbg_conv3x3_wprep(const vbg_conv3_wprep_entry* table_dev, const vbg_conv3_wprep_entry* table_host, int n, void* stream) {
    VBG_CHECK_ARG(n >= 0 && (n == 0 || (table_dev && table_host)));
    if (n == 0) return VBG_OK;
    long long most = 0;
    for (int i = 0; i < n; ++i) {
        const vbg_conv3_wprep_entry& e = table_host[i];
        const int rows = e.flip ? e.Cin : e.Cout, K = e.flip ? e.Cout : e.Cin;
        VBG_CHECK_ARG(e.w && e.out && rows > 0 && K > 0 && K % 16 == 0 && e.Cin % 8 == 0 && (e.bn == 64 || e.bn == 128));
        VBG_CHECK_ARG((((uintptr_t)e.w) & 15) == 0 && (((uintptr_t)e.out) & 15) == 0);
        const long long groups = (long long)((rows + e.bn - 1) / e.bn) * 9 * (K / 16) * e.bn * 2;
        most = groups > most ? groups : most;
    }
    VBG_CHECK_ARG((most + 255) / 256 < (1ll << 31));
    VBG_LAUNCH(vbg::conv3_wprep_kernel, dim3((unsigned)((most + 255) / 256), (unsigned)n), dim3(256), 0, (hipStream_t)stream, table_dev);
    VBG_LAUNCH_RET();
}

static int conv3x3_impl(const float* x, const float* w, const unsigned short* wp, const float* bias, float* y, double* stats, int stats_slots,
                        int B, int H, int W, int Cs, int N, int accumulate, int form, const unsigned* x_amax, float* split_slab,
                        unsigned* split_tickets, int nsplit, void* stream, int bn_req = 0);

extern "C" int vbg_conv3x3(const float* x, const float* w, const float* bias, float* y, double* stats, int stats_slots, int B, int H,
                           int W, int Cs, int N, int accumulate, int form, const unsigned* x_amax, float* split_slab,
                           unsigned* split_tickets, int nsplit, void* stream) {
    VBG_CHECK_ARG(w);
    return conv3x3_impl(x, w, nullptr, bias, y, stats, stats_slots, B, H, W, Cs, N, accumulate, form, x_amax, split_slab, split_tickets, nsplit, stream);
}

extern "C" int vbg_conv3x3_pw(const float* x, const void* w_planes, const float* bias, float* y, double* stats, int stats_slots, int B, int H,
                              int W, int Cs, int N, int accumulate, const unsigned* x_amax, float* split_slab, unsigned* split_tickets,
                              int nsplit, int bn, void* stream) {
    VBG_CHECK_ARG(w_planes && (((uintptr_t)w_planes) & 15) == 0 && (bn == 0 || bn == 64 || bn == 128));
    return conv3x3_impl(x, nullptr, (const unsigned short*)w_planes, bias, y, stats, stats_slots, B, H, W, Cs, N, accumulate, 1, x_amax, split_slab,
                        split_tickets, nsplit, stream, bn);
}

extern "C" int vbg_conv3x3_pw_amp(const float* x, const void* w_planes, const float* bias, float* y, double* stats, int stats_slots, int B, int H,
                                  int W, int Cs, int N, int accumulate, const unsigned* x_amax, float* split_slab, unsigned* split_tickets,
                                  int nsplit, int bn, void* stream) {
    VBG_CHECK_ARG(w_planes && (((uintptr_t)w_planes) & 15) == 0 && (bn == 0 || bn == 64 || bn == 128));
    return conv3x3_impl(x, nullptr, (const unsigned short*)w_planes, bias, y, stats, stats_slots, B, H, W, Cs, N, accumulate, 2, x_amax, split_slab,
                        split_tickets, nsplit, stream, bn);
}

static int conv3x3_impl(const float* x, const float* w, const unsigned short* wp, const float* bias, float* y, double* stats, int stats_slots,
                        int B, int H, int W, int Cs, int N, int accumulate, int form, const unsigned* x_amax, float* split_slab,
                        unsigned* split_tickets, int nsplit, void* stream, int bn_req) {
    VBG_CHECK_ARG(form == 0 || form == 1 || (form == 2 && wp));          // (2: the one-product `amp` form of the pre-split kernels)
    VBG_CHECK_ARG(bn_req == 0 || wp);
    VBG_CHECK_ARG(!x_amax || form >= 1);
    VBG_CHECK_ARG(x && (w || wp) && y && B > 0 && H > 0);
    const bool roi = H == 7 && W == 7;                          // [B, 7, 7, C] region maps: two images per 128-slot tile
    VBG_CHECK_ARG(roi || (W >= 16 && W <= 4096 && (W & (W - 1)) == 0));
    VBG_CHECK_ARG(roi || ((long long)H * W) % 64 == 0);
    VBG_CHECK_ARG((long long)H * W * Cs < (1ll << 29) && (!roi || (long long)B * 49 * Cs < (1ll << 29)));
    VBG_CHECK_ARG(Cs >= 16 && Cs % 16 == 0 && N >= 4 && N % 4 == 0);
    VBG_CHECK_ARG((((uintptr_t)x) & 15) == 0 && (((uintptr_t)w) & 15) == 0 && (((uintptr_t)y) & 15) == 0);
    VBG_CHECK_ARG(!stats || (stats_slots >= 1 && !accumulate));
    vbg::conv3_args a;
    a.X = x; a.Wt = w; a.Wp = wp; a.bias = bias; a.Y = y; a.stats = stats; a.stats_slots = stats_slots;
    a.H = H; a.W = W; a.wsh = roi ? 3 : 31 - __builtin_clz((unsigned)W); a.Cs = Cs; a.N = N;
    const long long M = (long long)B * H * W;
    VBG_CHECK_ARG(M < (1ll << 31));
    a.M = (int)M; a.accumulate = accumulate; a.a_amax = x_amax; a.roi = roi ? B : 0;
    static const int no_ring_guard = [] { const char* e = getenv("VBG_DEBUG_CONV3_NO_RING_GUARD"); return (e && e[0] == '1') ? 1 : 0; }();
    a.dbg_no_ring_guard = no_ring_guard;
    // split form: nsplit blocks per tile meet in split_slab [tiles][nsplit][128 * 128] / split_tickets [tiles] (zero; left zero)
    const bool split = nsplit > 1;
    VBG_CHECK_ARG(nsplit >= 1 && (!split || (split_slab && split_tickets && !roi && N % (bn_req ? bn_req : 128) == 0 && ((long long)H * W) % 128 == 0)));
    a.ksplit = split && nsplit % 3 == 0 ? 3 : 1;
    a.csplit = split ? nsplit / a.ksplit : 1;
    a.slab = split_slab; a.tickets = split_tickets;
    VBG_CHECK_ARG(Cs % a.csplit == 0 && (Cs / a.csplit) % 16 == 0 && (((uintptr_t)split_slab) & 15) == 0);
    // filters per tile: 128, or 64 where the filter count is an odd multiple of 64 (the 64-channel stage)
    // 128-pixel tiles once they fill the chip (or the image does not divide into 64-pixel tiles any better), else 64-pixel tiles
    auto is_big = [&](int bn_) { return roi || (((long long)H * W) % 128 == 0 && ((M / 128) * vbg::cdiv(N, bn_) >= 240 || W >= 128)); };
    // bn_req (PW launches): the caller's choice of filters per tile -- the plane image was written for it -- on 128-pixel tiles
    const bool n64 = bn_req ? bn_req == 64 : (N % 128 != 0 && N % 64 == 0 && is_big(64));
    const int bn = n64 ? 64 : 128;
    const bool big = bn_req ? (roi || ((long long)H * W) % 128 == 0) : (split || is_big(bn));
    const dim3 g(roi ? (unsigned)((B + 1) / 2) : (unsigned)(M / (big ? 128 : 64)), (unsigned)vbg::cdiv(N, bn), (unsigned)nsplit);
    if (wp) {
        // the plane image was written for conv3_pw_bn(N) rows per filter tile: the launch must walk it with the same tile
        VBG_CHECK_ARG(big && (bn_req || bn == conv3_pw_bn(N)));
        // VBG_CONV3_PIPE=0: the lockstep k-loop of the PW kernels (A/B switch of the software-pipelined loop)
        static const bool pipe = !(getenv("VBG_CONV3_PIPE") && atoi(getenv("VBG_CONV3_PIPE")) == 0);
        if (form == 2) {
            if (n64) { VBG_LAUNCH((vbg::conv3x3_kernel<128, 64, true, 2, true>), g, dim3(256), 0, (hipStream_t)stream, a); }
            else { VBG_LAUNCH((vbg::conv3x3_kernel<128, 128, true, 2, true>), g, dim3(256), 0, (hipStream_t)stream, a); }
        } else if (pipe) {
            if (n64) { VBG_LAUNCH((vbg::conv3x3_kernel<128, 64, true, 2>), g, dim3(256), 0, (hipStream_t)stream, a); }
            else { VBG_LAUNCH((vbg::conv3x3_kernel<128, 128, true, 2>), g, dim3(256), 0, (hipStream_t)stream, a); }
        } else {
            if (n64) { VBG_LAUNCH((vbg::conv3x3_kernel<128, 64, true, 1>), g, dim3(256), 0, (hipStream_t)stream, a); }
            else { VBG_LAUNCH((vbg::conv3x3_kernel<128, 128, true, 1>), g, dim3(256), 0, (hipStream_t)stream, a); }
        }
        VBG_LAUNCH_RET();
    }
    if (n64) {
        if (form == 1) { VBG_LAUNCH((vbg::conv3x3_kernel<128, 64, true>), g, dim3(256), 0, (hipStream_t)stream, a); }
        else { VBG_LAUNCH((vbg::conv3x3_kernel<128, 64, false>), g, dim3(256), 0, (hipStream_t)stream, a); }
    } else if (form == 1) {
        if (big) { VBG_LAUNCH((vbg::conv3x3_kernel<128, 128, true>), g, dim3(256), 0, (hipStream_t)stream, a); }
        else { VBG_LAUNCH((vbg::conv3x3_kernel<64, 128, true>), g, dim3(256), 0, (hipStream_t)stream, a); }
    } else {
        if (big) { VBG_LAUNCH((vbg::conv3x3_kernel<128, 128, false>), g, dim3(256), 0, (hipStream_t)stream, a); }
        else { VBG_LAUNCH((vbg::conv3x3_kernel<64, 128, false>), g, dim3(256), 0, (hipStream_t)stream, a); }
    }
    VBG_LAUNCH_RET();
}

extern "C" int vbg_conv3x3_wflip(const float* w, int Cout, int Cin, float* out, void* stream) {
    VBG_CHECK_ARG(w && out && Cout > 0 && Cin > 0);
    dim3 g((unsigned)vbg::cdiv(Cin, 32), (unsigned)vbg::cdiv(Cout, 32), 9);
    VBG_LAUNCH(vbg::conv3_wflip_kernel, g, dim3(32, 8), 0, (hipStream_t)stream, w, Cout, Cin, out);
    VBG_LAUNCH_RET();
}
