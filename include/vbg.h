/* libvbg — C-ABI of the MI355X-native ViBERTgrid hot path (gfx950 only).
 *
 * The reference (ZeningLin/ViBERTgrid-PyTorch) is 100 % Python: it has no FFI, so there is no
 * existing binding to mirror.  Each entry point below replaces the third-party library call the
 * reference makes at the cited file:line (paths relative to the reference root); the Python side
 * that binds them with ctypes is vibertgrid-pytorch_amd/vbg/lib.py, and INTEGRATION.md shows the
 * stub a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless named `h_*`;
 *   - activations are NHWC / row-major [rows, channels] fp32; index tensors are int32 unless noted;
 *   - `stream` is a hipStream_t; nothing uses the default stream, nothing synchronises the device;
 *   - the caller owns every buffer including workspaces; the library keeps no mutable global state;
 *   - return value: 0 ok, negative = argument error (VBG_EARG = -1), positive = hipError_t;
 *   - re-entrant: may be called concurrently from autograd worker threads on different streams.
 */
#ifndef VBG_H
#define VBG_H

#ifdef __cplusplus
extern "C" {
#endif

#define VBG_VERSION 100

int vbg_version(void);

/* ------------------------------------------------------------------------------------------
 * GEMM / implicit-GEMM convolution (fp32 MFMA).  C[M,N] (+)= opA[M,K] * opB[K,N].
 * Replaces: torch.nn.Linear / Conv2d forward+backward inside transformers.BertModel
 * (model/BERTgrid_generator.py:134), model/ResNetFPN_ViBERTgrid.py:478-508 & 612-648,
 * model/semantic_segmentation_head.py:66-78, model/field_type_classification_head.py:64-75,
 * 181-188, 564-575, and torch.cat + conv1x1 / Linear fusions at ResNetFPN_ViBERTgrid.py:317-318,
 * 505-506, field_type_classification_head.py:185-188 (K segments read in place: no concat).
 * ------------------------------------------------------------------------------------------ */
enum {
    VBG_OP_DENSE_K = 0, /* elem(row,k) = p[row*ld + k]      (k contiguous; A may have K segments)  */
    VBG_OP_DENSE_R = 1, /* elem(row,k) = p[k*ld + row]      (row contiguous)                        */
    VBG_OP_CONV_K = 2,  /* A only: row = output pixel, k = (tap, channel) gathered from NHWC source */
    VBG_OP_CONV_R = 3,  /* B only: k = output pixel, col = (tap, ci) gathered from NHWC source      */
    VBG_OP_WT_R = 4     /* B only: conv weight [Cout][taps][Cin] read as k = (tap, co), col = ci    */
};
enum { VBG_EPI_NONE = 0, VBG_EPI_RELU = 1, VBG_EPI_GELU_DUAL = 2 /* C = x, C2 = gelu_erf(x) */,
       VBG_EPI_MUL_GELU_GRAD = 3 /* vbg_plane_gemm only: C = x * gelu_erf'(C2), C2 = h is an INPUT (GELU backward in the data-gradient product) */ };

typedef struct vbg_conv_geo {
    int Hs, Ws, Cs;      /* gather-source tensor [*, Hs, Ws, Cs] (X for fwd/wgrad, dY for dgrad)   */
    int Hr, Wr;          /* spatial dims of the row space (output pixels; dX pixels for dgrad)      */
    int kh, kw, stride, pad;
    int dgrad;           /* 0: src = row*stride - pad + tap;  1: src = (row + pad - tap) / stride   */
} vbg_conv_geo;

typedef struct vbg_gemm_desc {
    int M, N, K;
    const float* A; long long lda; int a_kind; int a_vec;   /* a_vec: 16-byte vector loads legal   */
    /* DENSE_K A may be split along K into up to 4 segments, each its own tensor read in place;     */
    /* shift>0: the segment lives at 1/2^shift resolution (nearest-upsampled on the fly), rows are  */
    /* pixels of an [*, a_H, a_W] map.                                                               */
    int a_nseg; const float* a_seg_ptr[4]; int a_seg_kend[4]; long long a_seg_ld[4]; int a_seg_shift[4];
    int a_H, a_W;
    int a_prologue; float a_scale;                          /* 1: a = max(a,0)*a_scale              */
    const float* B; long long ldb; int b_kind; int b_vec;
    vbg_conv_geo geo;
    float* C; long long ldc; float* C2; const float* bias;  /* bias[N] or NULL                      */
    int epi; float alpha; int accumulate;                   /* C += result (atomic only if splitk>1)*/
    int splitk;                                             /* >1 requires accumulate; 1 = never split; 0 = the library may zero C and
                                                               split a long reduction with few output tiles (linear epilogues only;
                                                               atomic order is not reproducible run to run)                         */
    int tile;                                               /* 0 auto, BM*1000+BN: 128128/128064/64064 */
    /* grouped problems: grp[g*8 + {0..6}] = M, N, K, offA, offB, offC, offBias (elements; offsets */
    /* are relative to A/B/C/bias and may be negative); NULL = single problem                       */
    const long long* grp; int ngroups; int grp_maxM, grp_maxN;
    int bk;                                                 /* k-tile depth: 0 auto, 16 or 32       */
    /* optional BatchNorm statistics of the OUTPUT, fused into the epilogue (the conv in front of a training-mode BatchNorm,
       model/ResNetFPN_ViBERTgrid.py:116-123): per column sum and sum of squares of the stored values are added (fp64 atomics) into
       slot row (row-tile index % stats_slots) of stats[stats_slots][2*N]; needs splitk == 1, no accumulate, ldc % 4 == 0, C 16-B aligned */
    double* stats; int stats_slots;
    /* arithmetic form of the products.
       0: v_mfma_f32_32x32x2_f32, exact fp32 products.
       1: amp -- operands stay fp32 in memory, are rounded to bf16 (nearest even) inside the kernel and multiplied with
          v_mfma_f32_32x32x16_bf16, fp32 accumulation; replaces torch.cuda.amp.autocast of pipeline/train_val_utils.py:264 for
          these ops.
       3: fp32-grade on the bf16 matrix cores -- every operand element is split exactly into three bf16 pieces inside the kernel and
          each product is the fp32 sum of the six piece products of order <= 2^-16 (error <= 2^-23 of the product).
       2 (round 6): fp32-grade on the fp16 matrix cores for operands INSIDE fp16's range -- two pieces per element (hi = fp16(x), lo' =
          fp16((x - hi) 2^11), round to nearest), three piece products: half the matrix-core work of form 3, the same measured error
          against fp64; |x| >= 65520 becomes inf, never a clipped value.  The FORWARD kinds only (A DENSE_K / CONV_K, B DENSE_K: activations
          and weights) where they run 64 x 64 tiles; any other kind or tile given form 2 runs form 3 (on the 128 x 128 tiles it measured slower).
       Forms 1 and 3 need 16-byte aligned operands (a_vec, b_vec); form 3 is not available for row-contiguous A (the library uses
       form 0 there), and products whose geometry forces 16-deep k-tiles run as form 0.  Results of all forms agree to fp32
       rounding for 0 / 3 and to bf16 operand rounding for 1. */
    int bf16;
    /* round 6, optional: DETERMINISTIC split of the reduction.  splitk > 1 with slab_stride > 0 (elements): split s of the k range STORES its
       partial product at C + s * slab_stride (plain stores, no atomics; no bias / epilogue / accumulate / groups / statistics), and
       vbg_slab_reduce adds the slabs in split order, with the bias and an optional ReLU.  For products with a handful of output tiles and a
       very long reduction: the first layer of the field-type classifier on ONE document is 128 x 1024 outputs over k = 13 312
       (model/field_type_classification_head.py:78-110) -- 32 tiles of 416 k-tiles otherwise. */
    long long slab_stride;
} vbg_gemm_desc;

int vbg_gemm(const vbg_gemm_desc* desc, void* stream);
/* Measurement hooks (bench.py `roofline`; no reference counterpart): the same launch with the dispatch's own begin / end
 * timestamps delivered to two events (hipExtLaunchKernel start / stop events: no barrier packets around the kernel, the figure
 * rocprofv3's kernel trace reports), plus create / destroy / elapsed for those events (hipEvent_t passed as void*). */
int vbg_gemm_timed(const vbg_gemm_desc* desc, void* stream, void* start_event, void* stop_event);
/* out[m][n] = (relu ? max(., 0) : .)(sum_{s < nslabs, in order} slabs[s * slab_stride + m * lds + n] + bias[n])  (bias may be NULL; N % 4 == 0) */
int vbg_slab_reduce(const float* slabs, int nslabs, long long slab_stride, int M, int N, long long lds, const float* bias, int relu,
                    float* out, long long ldc, void* stream);
int vbg_timer_create(void** event);
int vbg_timer_destroy(void* event);
int vbg_timer_elapsed_ms(void* start_event, void* stop_event, float* ms);

/* ------------------------------------------------------------------------------------------
 * Plane GEMM: the same fp32-grade NT product as vbg_gemm form 3, from operands that were split into bf16 planes ONCE
 * (by vbg_split_planes / vbg_split_planes_t or by a producing epilogue) instead of inside every block of every product.
 * A plane operand is [3][rows][ld] bf16 (hi, mid, lo: x = hi + mid + lo exactly), K-contiguous, ld a multiple of 8 and
 * >= K, K a multiple of 32 (the split kernels zero-pad).  C[M,N] (+)= alpha * A[M,K] * B[N,K]^T (+ bias[N]).
 * Replaces: torch.nn.functional.linear inside transformers BertModel (model/BERTgrid_generator.py:134) and its autograd
 * products (dgrad with the transposed weight planes, wgrad with transposed activation planes), the MLP heads
 * (model/field_type_classification_head.py:78-110).
 * ------------------------------------------------------------------------------------------ */
#define VBG_PLANE_MAX_GROUPS 4
typedef struct vbg_plane_group {                                  /* one problem of a grouped plane GEMM launch */
    const unsigned short* A; const unsigned short* B; float* C;
    long long a_plane, lda, b_plane, ldb, ldc;
    int M, N;
    int tiles_m, tiles_n;                                         /* filled by the library */
    const unsigned* a_amax;                                       /* form 1: amax slot whose power of two A was scaled by, or NULL */
} vbg_plane_group;
typedef struct vbg_plane_gemm_desc {
    int M, N, K;
    const unsigned short* A; long long a_plane; long long lda;   /* plane stride / row stride in elements */
    const unsigned short* B; long long b_plane; long long ldb;
    float* C; long long ldc; float* C2; const float* bias;       /* C may be NULL when only Cp is wanted   */
    /* optional: the stored value (after bias / ReLU / GELU) split into planes [3][M][ldp] -- the A operand of the next product */
    unsigned short* Cp; long long c_plane; long long ldp;
    int epi; float alpha; int accumulate;                         /* C += result (atomics only if splitk > 1) */
    int splitk;                                                   /* >1 requires accumulate */
    int tile;                                                     /* 0 auto; 256256 / 256128 / 128128 / 128129 / 128130 / 128064 / 64064;
                                                                     64004: 64 x 64 with four LDS stages (forms 1, 2: cold weights, few tiles) */
    /* trans != 0 ("TN"): C[M,N] (+)= alpha * sum_k A[k,M] * B[k,N] -- the reduction index is the operands' ROW index: A planes
       [3][K][lda], B planes [3][K][ldb], lda / ldb multiples of 32 (zero padded), K any length.  The weight gradient dW = dY^T X
       straight from the planes of dY and X that the data-gradient / forward products already use (LDS transpose reads). */
    int trans;
    /* ngroups > 0: a grouped launch -- grp[0..ngroups) are independent problems of ONE reduction length K (and one `trans`,
       `accumulate`, `alpha`) that share the grid, so their partial rounds of output tiles fill the chip together (the four weight
       gradients of an encoder layer: 432 tiles in 2 rounds instead of 4 launches of <= 144 tiles on 256 CUs).  No bias / epilogue. */
    int ngroups;
    vbg_plane_group grp[VBG_PLANE_MAX_GROUPS];
    /* optional stream-K tail (NT products on the 8-wave 128 x 128 tiles, no split-K, not grouped): sk_blocks = number of CUs,
       sk_ws = fp32 workspace of sk_blocks * 2 * 128 * 128 elements, sk_cnt = int32 [sk_blocks] that is ZERO on entry (and left
       zero).  The whole rounds of tiles run one block per tile; the k-tile sequence of the last, partly filled round is cut into
       sk_blocks equal ranges whose partial tiles are summed (in block order: deterministic) by the last block to finish each tile.
       The workspace must not be shared by launches that may run concurrently.  NULL = plain rounds. */
    float* sk_ws; int* sk_cnt; int sk_blocks;
    int sk_full, sk_tiles_m, sk_tiles_n;                         /* filled by the library */
    /* optional: colsum[n] += sum_m (stored value)[m][n] (atomics, one per column and row tile): the bias gradient of the layer whose
       output gradient this product produces (NT, no split-K / groups / accumulate / GELU-dual / stream-K / 256256 tile) */
    float* colsum;
    /* form 1 (NT, tiles 128129 / 128130 / 256128, no split-K / groups): A and B are TWO fp16 planes [2][rows][ld] written by
       vbg_split_planes_pair, a producer's pair output or Cq below: hi = fp16(x), lo' = fp16((x - hi) 2^11), round to nearest; three
       piece products (hi hi, and lo' hi + hi lo' scaled by 2^-11 in the epilogue).  Half the matrix-core work and 4 instead of 6
       operand bytes per element for operands inside fp16's range (the forward products: LayerNorm / GELU outputs and weights);
       |x| >= 65520 becomes inf.  form 0: three bf16 planes, six piece products.
       form 2 (`amp`: the reference's autocast linears, pipeline/train_val_utils.py:264): the same fp16-pair operands, ONE product on their hi
       planes (x rounded to fp16; a_amax as in form 1), fp32 accumulation; tiles 128129 / 256128, NT and the weight-gradient products. */
    int form;
    /* optional: the stored value (after bias / GELU) also as fp16-pair planes [2][M][ldq] (plane stride q_plane elements) */
    unsigned short* Cq; long long q_plane; long long ldq;
    /* form 1, optional: the A planes hold x * 2^e, e = the power of two that brings the value of this amax slot to [2^13, 2^14)
       (vbg_split_planes_pair with the same slot): the product is scaled back by 2^-e -- a GRADIENT as the A operand.  With trans != 0
       form 1 also covers the weight-gradient products (single and grouped: vbg_plane_group.a_amax per problem). */
    const unsigned* a_amax;
    /* optional (NT): amax slot that receives max |stored value| of this launch (zeroed by the caller) */
    unsigned* c_amax;
    /* optional (NT 8-wave tiles, with Cq; round 4): the Cq planes hold (stored value) * 2^e where e comes from a rigorous BOUND of the
       stored values instead of their measured maximum -- a GRADIENT leaves the epilogue as pair planes without a split pass of its own:
       bound = value(cq_ref_in: an amax slot holding max |A operand|, unscaled) * value(*cq_l1_in: float bits of max_j sum_k |B[j][k]|, the
       largest row L1 norm of the NT B operand; NULL: 1) * cq_mul (the epilogue's Lipschitz constant, e.g. 1.13 for the GELU gradient, and
       the rounding margin); |sum_k a_k b_jk| <= max |a| * sum_k |b_jk| holds for every element.  The bound's bit pattern is written to
       word 0 of the zeroed amax slot cq_ref_out: consumers pass that slot as a_amax.  The two-piece form keeps 22 bits over ~30 binades,
       so a bound 2^5 ... 2^10 above the true maximum costs no precision. */
    const unsigned* cq_ref_in; const unsigned* cq_l1_in; float cq_mul; unsigned* cq_ref_out;
} vbg_plane_gemm_desc;
/* out[i] = max(out[i], bits of max_c sum_r |w_i[r][c]|) for n matrices (w_i [rows_i][cols_i], row stride ld_i): the largest column L1
 * norm -- with w = the nn.Linear weight [out, in] of a layer, the bound factor of the data gradient dY W (a cq_l1_in word).  One launch
 * for a table of matrices (device memory; max_cols = the largest cols_i); `out` words are maxed into: zero them first. */
typedef struct vbg_l1_entry { const float* w; long long ld; int rows, cols; } vbg_l1_entry;
int vbg_col_l1_max(const vbg_l1_entry* table_dev, int n, int max_cols, unsigned* out, void* stream);
int vbg_plane_gemm(const vbg_plane_gemm_desc* desc, void* stream);
int vbg_plane_gemm_timed(const vbg_plane_gemm_desc* desc, void* stream, void* start_event, void* stop_event);
/* x [rows][cols] fp32 (row stride ldx) -> planes [3][rows][ldp] (plane stride `plane` elements), columns cols..ldp-1 zero;
 * relu != 0: the pieces of max(x, 0); colsum_accum != NULL: colsum_accum[c] += sum_r x[r][c] in the same pass (the bias gradient
 * of a linear layer: torch's autograd `grad_output.sum(0)` for F.linear) */
int vbg_split_planes(const float* x, long long ldx, int rows, int cols, unsigned short* out, int ldp, long long plane, int relu,
                     float* colsum_accum, void* stream);
/* x [rows][cols] fp32 -> TRANSPOSED planes [3][cols][ldp], ldp >= rows (multiple of 32), entries rows..ldp-1 zero */
int vbg_split_planes_t(const float* x, long long ldx, int rows, int cols, unsigned short* out, int ldp, long long plane,
                       void* stream);
/* x [rows][cols] fp32 -> fp16-pair planes [2][rows][ldp] (hi, lo' as above; columns cols..ldp-1 zero): operands of form-1 products */
int vbg_split_planes_pair(const float* x, long long ldx, int rows, int cols, unsigned short* out, int ldp, long long plane,
                          const unsigned* amax, float* colsum_accum, void* stream);
/* amax (optional): x is multiplied by the power of two that brings the slot's value to [2^13, 2^14) first (gradient operands);
 * colsum_accum (optional): colsum_accum[c] += sum_r x[r][c] of the unscaled values, as in vbg_split_planes */
/* many matrices of one fp32 buffer in one launch: tbl_dev = device int64 [njobs][6] = {source offset (elements), rows, cols,
 * destination offset (elements), ldp (multiple of 32, >= rows), index of the job's first 64 x 64 tile}; job j writes the planes of
 * its matrix TRANSPOSED ([3][cols][ldp]) at dst + offset, plane stride `plane` for all jobs.  (The W^T planes of every weight of a
 * flat parameter buffer, refreshed once per optimizer step.) */
int vbg_split_planes_t_batched(const float* src, unsigned short* dst, const long long* tbl_dev, int njobs, int total_tiles,
                               long long plane, void* stream);
/* the same as fp16-pair planes [2][cols][ldp] per job (the W^T operands of the form-1 data-gradient products) */
int vbg_split_planes_pair_t_batched(const float* src, unsigned short* dst, const long long* tbl_dev, int njobs, int total_tiles,
                                    long long plane, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused self-attention on packed variable-length sequences (head width 64), fp32-grade on the bf16 matrix cores; no [L, L]
 * score / probability block is ever written.  Replaces transformers' BertSelfAttention inside BertModel
 * (model/BERTgrid_generator.py:134): softmax(Q K^T * scale) -> dropout -> P V, and its autograd.
 * Q, K, V are columns [0, hid), [hid, 2 hid), [2 hid, 3 hid) (hid = heads * 64) of the plane tensor `qkv` [3][ntok][qkv_ld]
 * (vbg_plane_gemm's Cp output); dO (backward) is a plane tensor [3][ntok][do_ld].  Sequence s owns token rows
 * [seq_row0[s], seq_row0[s] + seq_len[s]); tasks = int32 [ntasks][2] = (sequence, 128-row block) pairs; grid = ntasks x heads.
 *   mode FWD: out = O [ntok][ldo] (columns head * 64 ..); lse [2][heads][ntok_pad]: [0][head][pad_off[s] + row] = maximum m of the scaled
 *             scores of the row, [1][..] = 1 / sum exp(x - m) (the two numbers the row was normalised with; the backward modes read them);
 *   mode DQ : out = dqkv [ntok][ldo]: columns [0, hid) <- dQ; forms delta = rowsum(dO o O) from the dO planes and `o`, WRITES the rows' own
 *             sum_k P_k dP_k into `delta` (zero in the padding rows on entry) and corrects dQ for the difference with kbar;
 *   mode DKV: columns [hid, 3 hid) <- dK, dV; run it after DQ.
 * delta: fp32 [heads][ntok_pad]; pad_off[s] a multiple of 32 with room for roundup(seq_len, 32) rows, rows past a
 * sequence's length ZERO.  Dropout: mask_q / mask_k from vbg_attn_mask (NULL = none), keep_scale =
 * 65536 / (65536 - vbg_attn_drop_thr16(p)); mask_off[s] = first word of sequence s (heads * roundup(len,32) * ceil(len/32) words). */
enum { VBG_ATTN_FWD = 0, VBG_ATTN_DQ = 1, VBG_ATTN_DKV = 2 };
typedef struct vbg_attn_desc {
    int mode, heads, ntasks, max_len;                 /* max_len: longest sequence (<= 512: BERT windows) */
    const int* tasks; const int* seq_len; const int* seq_row0; const int* pad_off; long long ntok_pad;
    const unsigned short* qkv; long long qkv_plane, qkv_ld;
    const unsigned short* dO; long long do_plane, do_ld;
    float* out; long long ldo;
    float* lse; float* delta;
    unsigned short* out_planes; long long op_plane, op_ld;   /* FWD, optional: the three bf16 planes [3][ntok][op_ld] of O */
    float* kbar; long long ldk;     /* [ntok][ldk]: FWD writes sum_k P_k K_k (bf16 precision; NULL = skip), DQ reads it */
    const float* o;                 /* DQ: the forward's O as fp32 [ntok][ldk] (delta = rowsum(dO o O) is formed in the kernel) */
    const unsigned* mask_q; const unsigned* mask_k; const long long* mask_off;
    float scale, keep_scale;
    unsigned* out_amax;             /* DQ / DKV, optional: amax slot (zeroed by the caller) that receives max |value written to out| */
    unsigned short* out_pair; long long oq_plane, oq_ld;     /* FWD, optional (round 4): O also as fp16-pair planes [2][ntok][oq_ld] -- the B operand
                                                                of the output projection's weight gradient, saved for backward instead of split there */
    int form;                       /* round 5.  0: qkv / dO are three bf16 planes, six piece products per product (above);
                                       1: qkv / dO are fp16-pair planes [2][ntok][ld] (hi, (x - hi) 2^11; vbg_plane_gemm's Cq output), three fp16 piece
                                          products per product into one accumulator set (one operand of each cross product carries the 2^-11);
                                       2: the hi planes of the same tensors alone, one product (`amp`: what fp16 autocast multiplies) */
    const unsigned* do_amax;        /* forms 1 / 2, DQ / DKV: the amax slot dO's planes were scaled with (their producer's bound; NULL: unscaled) */
} vbg_attn_desc;
int vbg_attn(const vbg_attn_desc* desc, void* stream);
/* dropout keeps of one layer and step, both orientations (torch.nn.Dropout(attention_probs_dropout_prob) on the probabilities):
 * keep <=> a 16-bit slice of the counter hash >= thr16 = round(p * 65536), i.e. drop rate thr16 / 65536 */
unsigned vbg_attn_drop_thr16(float drop_p);
int vbg_attn_mask(const int* seq_len, const long long* mask_off, int nseq, int heads, int maxlen, float drop_p,
                  unsigned long long seed, unsigned long long stream_id, unsigned* mask_q, unsigned* mask_k, void* stream);
/* the same for `nlayers` encoder layers of one step in ONE launch (transformers BertSelfAttention.dropout of every layer of
 * model/BERTgrid_generator.py:134's encoder): layer l draws from stream id stream_id0 + l * stream_id_stride and owns the words
 * [l * layer_words, (l + 1) * layer_words) of mask_q / mask_k; bit-identical to nlayers calls of vbg_attn_mask */
int vbg_attn_mask_layers(const int* seq_len, const long long* mask_off, int nseq, int heads, int maxlen, float drop_p,
                         unsigned long long seed, unsigned long long stream_id0, unsigned long long stream_id_stride, int nlayers,
                         long long layer_words, unsigned* mask_q, unsigned* mask_k, void* stream);

/* column sums: out[n] (+)= sum_m x[m*ld + n]   (bias gradients) */
int vbg_colsum(const float* x, long long ld, int M, int N, float* out, int accumulate, void* stream);
/* the same with every partial sum in fp64 (ws: N doubles of caller scratch, overwritten): bias gradients of the 1x1 segmentation
 * classifiers (model/semantic_segmentation_head.py:66-78: torch's pairwise sum over ~1e6 pixel terms that cancel to 1e-3 of their
 * running magnitude) and every other bias gradient that does not ride on a split pass */
int vbg_colsum_f64(const float* x, long long ld, int M, int N, float* out, int accumulate, double* ws, void* stream);

/* 3x3 / stride 1 / pad 1 convolution, NHWC, fp32-grade split form, activation rows reused across the three horizontal taps
 * (csrc/conv3.hip): y[B,H,W,N] (+)= conv(x[B,H,W,Cs], w[N,3,3,Cs]) (+ bias); stats: BatchNorm slot workspace [slots][2][N] fp64 that
 * receives the per-channel sum / sum of squares of y (not with accumulate).  Replaces torch.nn.Conv2d(k=3, s=1, p=1) forward
 * (model/ResNetFPN_ViBERTgrid.py:478-508, 612-648; model/semantic_segmentation_head.py) and, with the filter written by
 * vbg_conv3x3_wflip, its input gradient.  Requires W a power of two >= 16, H*W % 64 == 0, Cs % 16 == 0, N % 4 == 0 -- or H = W = 7:
 * the [B,7,7,C] region-of-interest maps of the field-type head (model/field_type_classification_head.py:64-75), stored compactly, two
 * images per 128-row tile.  Filter counts that are odd multiples of 64 run 64-filter tiles.
 * nsplit > 1 (vbg_conv3x3_split(...) says how many: the late trunk stages with 64-128 tiles of 2304-4608 long reductions): nsplit
 * workgroups share a tile, each reducing one filter row (nsplit % 3 == 0) and / or one channel group; they meet through split_slab
 * ([tiles][nsplit][128*128] floats of caller scratch) and split_tickets ([tiles] words, zero on entry and left zero), the last
 * arriver adding the partial tiles in a fixed order (deterministic).  nsplit = 1: both may be NULL. */
int vbg_conv3x3_split(int B, int H, int W, int Cs, int N);
int vbg_conv3x3(const float* x, const float* w, const float* bias, float* y, double* stats, int stats_slots, int B, int H, int W,
                int Cs, int N, int accumulate, int form, const unsigned* x_amax, float* split_slab, unsigned* split_tickets,
                int nsplit, void* stream);
/* form 0: three bf16 pieces per operand, six piece products (any operands); form 1: two fp16 pieces (round to nearest), three piece
 * products -- same measured accuracy against fp64 and half the matrix-core work, for operands inside fp16's range; a magnitude of
 * 65520 or more becomes inf, never a silently clipped value.  x_amax (form 1, optional): amax slot (below) holding the bit pattern of
 * max |x| (vbg_amax, or the amax output of vbg_bn_bwd_apply): x is multiplied by the power of two that brings that maximum to
 * [2^13, 2^14) on its way into the kernel and the result by its inverse (exact), which puts a GRADIENT operand inside fp16's range:
 * the input gradient of the convolution in form 1. */
/* An "amax slot" is VBG_AMAX_WORDS = 64 device words VBG_AMAX_STRIDE = 32 words (128 bytes, one L2 line) apart -- 8 KB in all; its
 * value -- the bit pattern of a tensor's largest magnitude (non-negative floats order like their bit patterns) -- is the max over the
 * words.  Producers max INTO a slot (their blocks spread over the words: atomics on one L2 line serialise at ~10 ns each), the caller
 * zeroes it; consumers reduce the 64 words with one wave.  vbg_amax: slot = max(slot, max |x[i]|). */
#define VBG_AMAX_WORDS 64
#define VBG_AMAX_STRIDE 32
int vbg_amax(const float* x, long long n, unsigned* amax, void* stream);
/* out[ci][2-kh][2-kw][co] = w[co][kh][kw][ci]: the filter with which the input gradient of a 3x3 / s1 / p1 convolution is the
 * same convolution of dy */
int vbg_conv3x3_wflip(const float* w, int Cout, int Cin, float* out, void* stream);

/* The same convolution in form 1 with the FILTER PRE-SPLIT (round 4): `w_planes` = the fp16-pair image of the filter that
 * vbg_conv3x3_wprep wrote (flip = 0: the forward of torch.nn.Conv2d(k=3, s=1, p=1); flip = 1: the filter of the input gradient, i.e.
 * vbg_conv3x3_wflip + split in one pass).  The kernel streams the filter through LDS-DMA in 128-byte lines and spends no VALU on it;
 * results equal vbg_conv3x3(form 1) bit for bit (same pieces, same products, same order).  Needs 128-pixel tiles (the shapes for which
 * vbg_conv3x3 picks them, incl. every nsplit > 1 and the 7x7 region maps); other shapes are argument errors.
 * vbg_conv3x3_wprep: ONE launch for a table of filters (the caller keeps the table in device AND host memory; entries must stay valid
 * until the launch has run): entry = {w [Cout,3,3,Cin] fp32, out, Cout, Cin, flip, bn}; `out` receives vbg_conv3x3_wprep_bytes(Cout,
 * Cin, flip) bytes; bn = rows per filter tile = 64 when the image's row count (flip ? Cin : Cout) is an odd multiple of 64, else 128.
 * The reduction width (flip ? Cout : Cin) must be a multiple of 16. */
typedef struct vbg_conv3_wprep_entry {
    const float* w;
    void* out;
    int Cout, Cin, flip, bn;
} vbg_conv3_wprep_entry;
long long vbg_conv3x3_wprep_bytes(int Cout, int Cin, int flip, int bn);
int vbg_conv3x3_wprep(const vbg_conv3_wprep_entry* table_dev, const vbg_conv3_wprep_entry* table_host, int n, void* stream);
int vbg_conv3x3_pw(const float* x, const void* w_planes, const float* bias, float* y, double* stats, int stats_slots, int B, int H, int W,
                   int Cs, int N, int accumulate, const unsigned* x_amax, float* split_slab, unsigned* split_tickets, int nsplit,
                   int bn, void* stream);
/* `amp` form of vbg_conv3x3_pw (the reference's `amp: True`, pipeline/train_val_utils.py:264: autocast convolutions): ONE product on the
 * hi pieces -- x (scaled into range by x_amax as above) and the filter rounded to fp16, fp32 accumulation; the lo planes of the image are
 * not loaded at 128 filters per tile.  Same arguments, same image. */
int vbg_conv3x3_pw_amp(const float* x, const void* w_planes, const float* bias, float* y, double* stats, int stats_slots, int B, int H, int W,
                       int Cs, int N, int accumulate, const unsigned* x_amax, float* split_slab, unsigned* split_tickets, int nsplit,
                       int bn, void* stream);
/* bn (vbg_conv3x3_pw / vbg_conv3x3_wprep_bytes): filters per tile the image was written for -- 0: the library's rule above; 64 / 128: the
 * caller's choice (64-filter tiles double the tile count of a launch: the late trunk stages, whose 128-filter tiles do not fill the chip);
 * the launch then runs 128-pixel tiles whatever the tile count, and nsplit > 1 needs N % bn == 0 (slabs of 128 * bn floats). */

/* weight gradient of that convolution: dw[Cout,3,3,Cs] += sum over pixels dy[B,H,W,Cout]^T * shifted x[B,H,W,Cs] (csrc/conv3.hip:
 * operands stay [pixel][channel] in LDS, fragments through transposing LDS reads, one load + split of x serves all nine taps).
 * The pixel range is cut into vbg_conv3x3_wgrad_strips(...) strips; slab = [strips][Cout,3,3,Cs] scratch -> each strip stores its
 * partial result plainly and a second launch adds them into dw in a fixed order (deterministic); slab = NULL -> float atomics.
 * Replaces the weight gradient autograd forms for torch.nn.Conv2d(k=3, s=1, p=1).  Requires W % 16 == 0, Cs % 32 == 0 and
 * Cout % 128 == 0 (or Cout % 64 == 0 and Cs % 64 == 0); or H = W = 7 (region maps, Cout % 128 == 0): a k-tile is then two rows of the
 * image's 8 x 8 slot grid. */
int vbg_conv3x3_wgrad_strips(int B, int H, int W, int Cs, int Cout);
int vbg_conv3x3_wgrad(const float* dy, const float* x, float* dw, float* slab, int B, int H, int W, int Cs, int Cout, int form,
                      const unsigned* dy_amax, const unsigned* x_amax, void* stream);
/* form 0: three bf16 pieces per operand, six piece products; form 1: two fp16 pieces, three piece products, both operands scaled by
 * the powers of two that bring max |dy| and max |x| (bit patterns in the device words dy_amax / x_amax: vbg_amax, or the amax outputs
 * of vbg_bn_bwd_apply / vbg_bn_apply) to [2^13, 2^14), the result scaled back: exact scaling, half the matrix-core work */

/* stem im2col: NHWC [B,H,W,C] -> [B*Ho*Wo, Kpad] with k = (dy*kw+dx)*C + c, zero padded to Kpad */
int vbg_im2col(const float* x, int B, int H, int W, int C, int kh, int kw, int stride, int pad, int Kpad,
               float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * a1. input transform  (pipeline/transform.py:104-171, 225-271)
 * ------------------------------------------------------------------------------------------ */
/* one image CHW [3,h,w] -> normalised, bilinearly resized (align_corners=False, torch
 * recompute_scale_factor semantics: src = (dst+0.5)*(in/out)-0.5) into batch slot b of an NHWC
 * [B,H,W,3] buffer that the caller zero-filled (padding).  oh==h && ow==w is an exact copy. */
int vbg_normalize_resize(const float* img, int h, int w, int oh, int ow, const float* h_mean3, const float* h_std3,
                         float* batch_nhwc, int b, int H, int W, void* stream);
/* boxes int64 [S,4] -> int32 [S,4]: cols 0,2 *= ratio_h, cols 1,3 *= ratio_w in fp32, truncate */
int vbg_rescale_boxes(const long long* in, int S, float ratio_h, float ratio_w, int* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * a3. BERT encoder pieces (transformers BertModel; model/BERTgrid_generator.py:134)
 * ------------------------------------------------------------------------------------------ */
/* out[t] = dropout(LN(word[ids[t]] + pos[pos_ids[t]] + type[0])) ; saves xhat/rstd for backward */
int vbg_embed_ln_fwd(const int* ids, const int* pos_ids, int ntok, int hidden, const float* word, const float* pos,
                     const float* type0, const float* gamma, const float* beta, float eps, float drop_p,
                     unsigned long long seed, unsigned long long stream_id, float* out, float* xhat, float* rstd,
                     void* stream);
int vbg_embed_ln_bwd(const float* dout, const float* xhat, const float* rstd, const int* ids, const int* pos_ids,
                     int ntok, int hidden, const float* gamma, float drop_p, unsigned long long seed,
                     unsigned long long stream_id, float* dword, float* dpos, float* dtype0, float* dgamma,
                     float* dbeta, void* stream);
/* y = LN(dropout(x) + res) * gamma + beta ; x already holds the dense output + bias */
int vbg_dropout_add_ln_fwd(const float* x, const float* res, int rows, int hidden, const float* gamma,
                           const float* beta, float eps, float drop_p, unsigned long long seed,
                           unsigned long long stream_id, float* y, float* xhat, float* rstd, void* stream);
/* the same, and y's three bf16 planes [3][rows][ldp] (plane stride `plane`; ldp == hidden leaves no padding columns to zero, a wider
 * ldp expects them zero already) -- the A operand of the plane GEMM that consumes y: saves the separate split pass */
int vbg_dropout_add_ln_fwd_planes(const float* x, const float* res, int rows, int hidden, const float* gamma,
                                  const float* beta, float eps, float drop_p, unsigned long long seed,
                                  unsigned long long stream_id, float* y, float* xhat, float* rstd, unsigned short* y_planes, int ldp,
                                  long long plane, unsigned short* y_pair, int ldq, long long qplane, void* stream);
/* (y_pair, optional: y also as fp16-pair planes [2][rows][ldq] -- the A operand of the form-1 forward products) */
/* dx (to the dense output), dres, dgamma/dbeta +=.  `slots_ws` (optional): fp32 workspace [vbg_ln_bwd_ws_rows(rows)][2][hidden]; it
 * need not be initialised and holds nothing afterwards.  With it every block of the kernel stores its column sums as one plain row and a
 * second small launch adds the rows in a fixed order (deterministic; round 5: the 516 x 2304 float atomics of the earlier slot scheme
 * were a ~9 us tail at the L2's atomic rate); without it (few rows) the sums go straight into dgamma / dbeta as atomics */
int vbg_ln_bwd_ws_rows(int rows);
int vbg_dropout_add_ln_bwd(const float* dy, const float* xhat, const float* rstd, int rows, int hidden,
                           const float* gamma, float drop_p, unsigned long long seed, unsigned long long stream_id,
                           float* dx, float* dres, float* dgamma, float* dbeta, float* slots_ws, unsigned* dx_amax, void* stream);
/* dx_amax (optional): amax slot (zeroed by the caller) that receives max |dx| -- the scale of dx as an fp16-pair operand */
/* the same with dx delivered as bf16 planes [3][rows][ldp] (not as fp32) and its column sums added into dbias_accum[hidden]: dx is the
 * gradient of the dense output in front of the LayerNorm, which is only ever a plane operand of that layer's gradient products, and
 * its column sums are that layer's bias gradient.  slots3_ws: fp32 [vbg_ln_bwd_ws_rows(rows)][3][hidden], as above (mandatory). */
int vbg_dropout_add_ln_bwd_planes(const float* dy, const float* xhat, const float* rstd, int rows, int hidden,
                                  const float* gamma, float drop_p, unsigned long long seed, unsigned long long stream_id,
                                  unsigned short* dx_planes, int ldp, long long plane, float* dres, float* dgamma, float* dbeta,
                                  float* dbias_accum, float* slots3_ws, void* stream);
/* the same with dx delivered as fp16-pair planes [2][rows][ldp] scaled by the power of two of a rigorous bound of |dx| (round 4: no split
 * pass for this gradient): bound = max |dy| (amax slot dy_amax) * max |gamma| * max rstd * (2 + sqrt(hidden)) / keep * 1.01; its bit
 * pattern goes to word 0 of the zeroed slot dx_bound (the consumers' a_amax), max |dx| itself into the zeroed slot dx_amax (optional) */
int vbg_dropout_add_ln_bwd_pair(const float* dy, const float* xhat, const float* rstd, int rows, int hidden,
                                const float* gamma, float drop_p, unsigned long long seed, unsigned long long stream_id,
                                unsigned short* dx_pair, int ldp, long long plane, float* dres, float* dgamma, float* dbeta,
                                float* dbias_accum, float* slots3_ws, const unsigned* dy_amax, unsigned* dx_amax, unsigned* dx_bound,
                                void* stream);
/* attention probabilities, in place on the grouped score buffer: for group g (= seq*heads + head)
 * rows L=len[g/heads], row stride ldp[g/heads], block offset off[g]; P = softmax(S*scale);
 * dropped entries are stored NEGATED (sign bit = dropped), kept entries unscaled; pad columns = 0. */
int vbg_softmax_fwd(float* s, const long long* off, const int* len, const int* ldp, int ngroups, int heads,
                    int maxlen, float scale, float drop_p, unsigned long long seed, unsigned long long stream_id,
                    void* stream);
/* dS (in place on dp) from sign-encoded P and dP_drop: dS = scale * P * (keep*dP/(1-p) - sum_j P_drop*dP_drop) */
int vbg_softmax_bwd(const float* p, float* dp, const long long* off, const int* len, const int* ldp, int ngroups,
                    int heads, int maxlen, float scale, float drop_p, void* stream);
/* y[r,:] = softmax(x[r,:]) for the returned class probabilities (field_type_classification_head.py:587) */
int vbg_row_softmax(const float* x, int rows, int cols, float* y, void* stream);
int vbg_gelu_bwd(const float* h, float* dg_inout, long long n, void* stream);      /* dh = dg * gelu'(h) */
int vbg_relu_bwd(const float* y, float* dy_inout, long long n, void* stream);      /* dx = dy * (y > 0)  */

/* ------------------------------------------------------------------------------------------
 * One encoder layer forward in ONE call (round 6; csrc/encoder.hip).  Replaces one transformers BertLayer.forward inside BertModel
 * (model/BERTgrid_generator.py:134: BertSelfAttention, BertSelfOutput, BertIntermediate, BertOutput) = the seven launches
 *   vbg_plane_gemm (stacked Q/K/V projection -> planes) -> vbg_attn(FWD) -> vbg_plane_gemm (output projection) ->
 *   vbg_dropout_add_ln_fwd_planes -> vbg_plane_gemm (FFN1, GELU) -> vbg_plane_gemm (FFN2) -> vbg_dropout_add_ln_fwd_planes
 * with exactly the descriptors the caller would have passed one by one (bit-identical results); what it saves is host time per launch.
 * A vbg_planes_ref names a plane tensor (buf NULL = absent): three bf16 planes for a form-0 product, fp16-pair planes for forms 1 / 2.
 *   form_qkv / form_ao / form_ffn: arithmetic form of the products (vbg_plane_gemm_desc.form); form_attn: of the attention (vbg_attn_desc.form;
 *   != 0: pqkv receives fp16-pair planes, else three bf16 planes).
 *   xa = planes of x in the form of form_qkv; wqkv [3 hidden][hidden], wo, wi, wo2 = weight planes in the form of their product.
 *   form_ao != 0 reads pctxq (the attention writes it), else pctx; form_ffn != 0 reads px1q / pgq, else px1 / pg; every present output
 *   plane tensor is written (the caller keeps them for backward / hands py / pyq to the next layer).
 *   Dropout streams: stream_id0 + 1 (first LayerNorm), + 2 (second); attention keeps from mask_q / mask_k (vbg_attn_mask; NULL = none). */
typedef struct vbg_planes_ref { unsigned short* buf; long long plane, ld; } vbg_planes_ref;
typedef struct vbg_bert_layer_fwd_desc {
    int ntok, hidden, inter, heads;
    float eps, drop_p; unsigned long long seed, stream_id0;
    int form_qkv, form_attn, form_ao, form_ffn;
    int tile_qkv, tile_ao, tile_ffn1, tile_ffn2;
    /* packed sequences: the fields of vbg_attn_desc */
    int ntasks, max_len; const int* tasks; const int* seq_len; const int* seq_row0; const int* pad_off; long long ntok_pad;
    const unsigned* mask_q; const unsigned* mask_k; const long long* mask_off; float attn_scale, keep_scale;
    /* inputs */
    const float* x; vbg_planes_ref xa;
    vbg_planes_ref wqkv, wo, wi, wo2;
    const float* bqkv; const float* bo; const float* bi; const float* bo2; const float* g1; const float* b1; const float* g2; const float* b2;
    /* outputs (fp32 [ntok][hidden] unless noted) */
    vbg_planes_ref pqkv;                                     /* [ntok][3 hidden] planes of q, k, v */
    float* ctx; float* lse; float* kbar; vbg_planes_ref pctx, pctxq;      /* attention output, its row statistics [2][heads][ntok_pad], kbar (optional) */
    float* ao; float* x1; float* xhat1; float* rstd1; vbg_planes_ref px1, px1q;
    float* h; vbg_planes_ref pg, pgq;                        /* h [ntok][inter] fp32; gelu(h) as planes only */
    float* fo; float* y; float* xhat2; float* rstd2; vbg_planes_ref py, pyq;
} vbg_bert_layer_fwd_desc;
int vbg_bert_layer_fwd(const vbg_bert_layer_fwd_desc* desc, void* stream);

/* ------------------------------------------------------------------------------------------
 * a4. token -> segment aggregation  (model/BERTgrid_generator.py:148-191)
 * ------------------------------------------------------------------------------------------ */
/* tok_row[i] = row of the i-th token (mask==1 order) in `tok`; runs: start[s], len[s] in token order.
 * mode 0 mean = sequential sum in token order then / n (bit-exact vs the reference), 1 = first. */
int vbg_seg_reduce_fwd(const float* tok, const int* tok_row, const int* run_start, const int* run_len, int nseg,
                       int hidden, int mode, float* out, void* stream);
int vbg_seg_reduce_bwd(const float* dout, const int* tok_row, const int* run_start, const int* run_len, int nseg,
                       int hidden, int mode, float* dtok_accum, void* stream);

/* ------------------------------------------------------------------------------------------
 * a5 / a9. bbox -> owner map, grid scatter, label raster  (model/BERTgrid_generator.py:193-245,
 *          model/semantic_segmentation_head.py:314-341).  Bit-exact.
 * ------------------------------------------------------------------------------------------ */
/* boxes int32 [nbox,4] (x1,y1,x2,y2) for all docs, doc b owns boxes [box_off[b], box_off[b+1]).
 * owner[b,y,x] = GLOBAL index of the last box of doc b whose rectangle
 * rows int(y1/stride):int(y2/stride), cols int(x1/stride):int(x2/stride) (python slice clipping,
 * truncating division) covers the cell, else -1. */
int vbg_owner_map(const int* boxes, const int* box_off, int B, int gh, int gw, int stride, int* owner, void* stream);
/* grid[b,y,x,:] = emb[owner] or 0.  layout 0: NHWC [B,gh,gw,C]; 1: NCHW [B,C,gh,gw] (reference layout) */
int vbg_grid_scatter_fwd(const float* emb, const int* owner, int B, int gh, int gw, int C, int layout, float* grid,
                         void* stream);
/* demb[s,:] (+)= sum over cells owned by s of dgrid (CopySlices semantics); dgrid NHWC */
int vbg_grid_scatter_bwd(const float* dgrid, const int* owner, const int* boxes, const int* box_doc, int nbox, int gh,
                         int gw, int stride, int C, float* demb_accum, void* stream);
/* full-resolution labels from the stride-1 owner map: pos_neg (0 bg / 1 class>0 / 2 class==0), cls */
int vbg_label_raster(const int* owner, const int* seg_class, long long ncell, int* pos_neg, int* cls, void* stream);

/* ------------------------------------------------------------------------------------------
 * a6-a9, a11. convolution trunk helpers (NHWC): BatchNorm (Sync-able), max-pool, FPN resampling
 * ------------------------------------------------------------------------------------------ */
/* The two reductions below spread their fp64 partial sums over vbg_bn_slots() SLOT ROWS of 2*C doubles (same-address
 * atomics serialise; consumers fold the slots): `slots_accum` is [vbg_bn_slots()][2*C], zero on entry.  The folding entry
 * points take `clear_slots`: non-zero = zero the slot rows behind the read, so one persistent workspace serves every layer
 * without fill launches. */
int vbg_bn_slots(void);
/* per-channel sum / sum of squares over rows of x[M,C] (torch.nn.BatchNorm2d training statistics,
 * model/ResNetFPN_ViBERTgrid.py:116-123): slot[s][0..C) += sum, slot[s][C..2C) += sumsq */
int vbg_bn_stats(const float* x, long long M, int C, double* slots_accum, void* stream);
/* from sums over `count` rows held in `nslots` slot rows (nslots = 1: already folded, e.g. after a SyncBN all-reduce;
 * count_dev, if non-NULL, is a device scalar that overrides `count`: the all-reduced row count): mean, invstd;
 * running <- (1-mom)*running + mom*{mean, unbiased var} */
int vbg_bn_finalize(double* stats, int nslots, int clear_slots, double count, const double* count_dev, int C, float eps, float momentum,
                    float* mean, float* invstd, float* running_mean, float* running_var, void* stream);
/* y = relu?( (x-mean)*invstd*gamma + beta (+ res) ) */
int vbg_bn_apply(const float* x, const float* res, long long M, int C, const float* mean, const float* invstd,
                 const float* gamma, const float* beta, int relu, float* y, unsigned* y_amax, void* stream);
/* y_amax (optional): amax slot (VBG_AMAX_WORDS words, zeroed by the caller) that receives the bit pattern of max |y|: the scale of y as an operand of
 * fp16-form products */
/* backward reductions: slot[s][0..C) += sum(g), slot[s][C..2C) += sum(g*xhat), g = dy*(y>0 if relu) */
int vbg_bn_bwd_reduce(const float* dy, const float* y, const float* x, long long M, int C, const float* mean,
                      const float* invstd, int relu, double* slots_accum, void* stream);
/* dx = gamma*invstd*(g - sum_g/count - xhat*sum_gx/count) from FOLDED sums[2*C]; dres = g (optional);
 * dgamma += sum_gx, dbeta += sum_g (optional) */
int vbg_bn_bwd_apply(const float* dy, const float* y, const float* x, long long M, int C, const float* mean,
                     const float* invstd, const float* gamma, const double* sums, double count, const double* count_dev,
                     int relu, float* dx,
                     float* dres, float* dgamma_accum, float* dbeta_accum, unsigned* dx_amax, void* stream);
/* dx_amax (optional): amax slot (VBG_AMAX_WORDS words, zeroed by the caller) that receives the bit pattern of max |dx| -- the scale of the fp16-form
 * products that consume dx */
/* vbg_bn_finalize + vbg_bn_apply, and vbg_bn_param_grad + vbg_bn_bwd_apply, as ONE launch each (round 5; torch.nn.BatchNorm2d training
 * forward / backward of model/ResNetFPN_ViBERTgrid.py:116-123 on one rank): every block folds the `nslots` slot rows of its 64 channels in
 * its prologue (fold order of the separate entry points: the same bits), the block of the first row chunk publishes mean / invstd / the
 * running statistics (forward) or adds the affine gradients (backward: dbeta += sum_g, dgamma += sum_gx).  C % 64 == 0.  The slot rows are
 * NOT cleared: hand in zeroed rows per use. */
int vbg_bn_apply_fold(const float* x, const float* res, long long M, int C, double* slots, int nslots, double count,
                      const double* count_dev, float eps, float momentum, float* mean, float* invstd, float* running_mean,
                      float* running_var, const float* gamma, const float* beta, int relu, float* y, unsigned* y_amax, void* stream);
/* (count_dev, optional: a device scalar that overrides `count` -- SyncBatchNorm hands in the all-reduced [sum, sumsq, count] buffer as
 * slots with nslots = 1 and its last element as count_dev)
 * vbg_bn_fold_count: fold the slot rows into folded[0..2C) and put `count` (this rank's row count) into folded[2C]: the buffer of
 * torch.nn.SyncBatchNorm's forward all-reduce, from one launch */
int vbg_bn_fold_count(double* slots, int nslots, int clear_slots, int C, double* folded, double count, void* stream);
int vbg_bn_bwd_apply_fold(const float* dy, const float* y, const float* x, long long M, int C, const float* mean,
                          const float* invstd, const float* gamma, double* slots, int nslots, double count, int relu, float* dx,
                          float* dres, float* dgamma_accum, float* dbeta_accum, unsigned* dx_amax, void* stream);
/* fold `nslots` slot rows: folded[0..2C) = sum over slots (optional output), and (optional) the BatchNorm affine
   gradients from these LOCAL sums: dbeta += (float)folded[0..C), dgamma += (float)folded[C..2C)  (call before a SyncBN
   all-reduce of `folded`; torch.nn.SyncBatchNorm leaves weight/bias gradients per-rank for DDP to average,
   model/ResNetFPN_ViBERTgrid.py:196-206 `norm_layer`) */
int vbg_bn_param_grad(double* slots, int nslots, int clear_slots, int C, double* folded, float* dgamma_accum, float* dbeta_accum,
                      void* stream);
int vbg_maxpool3x3s2_fwd(const float* x, int B, int H, int W, int C, float* y, int* argmax, void* stream);
/* bwd: every dx element is WRITTEN (gather over the <= 4 windows that contain it; no atomics, no zero fill needed) */
int vbg_maxpool3x3s2_bwd(const float* dy, const int* argmax, int B, int Ho, int Wo, int C, int H, int W,
                         float* dx, void* stream);
/* nn.AvgPool2d(2, 2) of the ResNet-D projection shortcut (model/ResNetFPN_ViBERTgrid.py:222-236): x [B,H,W,C] ->
 * y [B,H/2,W/2,C] (floor); bwd: dx [B,H,W,C] = dy/4 over each 2x2 window, 0 on a dropped trailing row / column */
int vbg_avgpool2_fwd(const float* x, int B, int H, int W, int C, float* y, void* stream);
int vbg_avgpool2_bwd(const float* dy, int B, int H, int W, int C, float* dx, void* stream);
/* y[b,y,x,:] = lo[b,y/2,x/2,:] + skip[b,y,x,:]   (nearest x2 upsample + add) */
int vbg_upsample2_add(const float* lo, const float* skip, int B, int H, int W, int C, float* y, void* stream);
/* lo[b,y,x,:] (+)= sum of the f x f block of hi   (backward of nearest upsampling by f) */
int vbg_sumpool(const float* hi, int B, int H, int W, int C, int f, float* lo, int accumulate, void* stream);
/* layout changes: [B,C,HW] <-> [B,HW,C] */
int vbg_nchw_to_nhwc(const float* x, int B, int C, int HW, float* y, void* stream);
int vbg_nhwc_to_nchw(const float* x, int B, int C, int HW, float* y, void* stream);
/* nearest-upsample a [B,h,w,C] NHWC map by f into NCHW [B,C,h*f,w*f] (eval-mode seg logits) */
int vbg_upsample_nhwc_to_nchw(const float* x, int B, int h, int w, int C, int f, float* y, void* stream);
int vbg_add_inplace(float* a, const float* b, long long n, void* stream);

/* ------------------------------------------------------------------------------------------
 * a10. RoIAlign  (torchvision.ops.RoIAlign(7, 1/4, sampling_ratio=-1, aligned=False);
 *      model/grid_roi_align.py:37-41, 81).  feat NHWC [B,H,W,C]; boxes int32 image coords.
 * ------------------------------------------------------------------------------------------ */
int vbg_roi_align_fwd(const float* feat, int B, int H, int W, int C, const int* boxes, const int* box_doc, int nroi,
                      int out, float scale, float* y /* [nroi,out,out,C] */, void* stream);
int vbg_roi_align_bwd(const float* dy, int B, int H, int W, int C, const int* boxes, const int* box_doc, int nroi,
                      int out, float scale, float* dfeat_accum, void* stream);

/* ------------------------------------------------------------------------------------------
 * a13. losses  (pipeline/custom_loss.py:35-101, 127-201)
 * ------------------------------------------------------------------------------------------ */
/* per-element CE.  Element i of the list is pixel/row e = elem ? elem[i] : i; its label is labels[e];
 * its logits row is e itself (H <= 0) or, for logits kept at 1/2^up_shift resolution of an
 * [*, H, W] label map, the low-resolution pixel under e (nearest upsampling folded into the index).
 * loss[i] = -w[t] * log_softmax(x)[t]. */
int vbg_ce_fwd(const float* logits, long long ld, int ncls, const int* elem, const int* labels, long long n,
               const float* weight, int up_shift, int H, int W, float* loss, void* stream);
/* dlogits[row(e)] += g * w[t] * (softmax(x) - onehot(t)),  g = gmul * (gscale_dev ? *gscale_dev : 1) (atomic) */
int vbg_ce_bwd(const float* logits, long long ld, int ncls, const int* elem, const int* labels, long long n,
               const float* weight, const float* gscale_dev, float gmul, int up_shift, int H, int W,
               float* dlogits_accum, void* stream);
/* order-preserving compaction: out_idx = ascending i with (labels[i] == value) == eq; *out_count_dev = how many */
long long vbg_compact_ws_bytes(long long n);
int vbg_compact(const int* labels, long long n, int value, int eq, int* out_idx, int* out_count_dev, void* ws,
                long long ws_bytes, void* stream);
/* STABLE descending sort (LSD radix): keys_out sorted, idx_out[r] = original position of rank r */
long long vbg_sort_ws_bytes(long long n);
int vbg_sort_desc(const float* keys, long long n, float* keys_out, int* idx_out, void* ws, long long ws_bytes,
                  void* stream);
int vbg_gather_f32(const float* src, const int* idx, long long n, float* out, void* stream);
int vbg_gather_i32(const int* src, const int* idx, long long n, int* out, void* stream);
int vbg_sum_f32(const float* x, long long n, float* out_accum, void* stream);

/* ------------------------------------------------------------------------------------------
 * a12 (classifier_mode full / crf). row subsets and the linear-chain CRF
 * ------------------------------------------------------------------------------------------ */
/* dst[r,:] = src[idx[r],:]  /  dst[idx[r],:] += src[r,:]  : `fuse_embeddings[pred_pos_neg_mask]` and its backward
 * (model/field_type_classification_head.py:371, 389-395) */
int vbg_gather_rows(const float* src, const int* idx, long long n, int C, float* dst, void* stream);
int vbg_scatter_rows_add(const float* src, const int* idx, long long n, int C, float* dst_accum, void* stream);
/* CRF over `ndoc` documents whose segments are the row ranges doc_off[d]..doc_off[d+1] of emissions [N, ntag] (ntag <= 64,
 * trans[i*ntag + j] = score of j -> i, model/crf.py:33-46).  fwd: forward algorithm (:48-79) and gold path score (:81-97):
 * nll[d] = (log Z_d - score_d) / n_d (:147-151); alpha [N, ntag] and logz [ndoc] are kept for bwd, which writes
 * demissions [N, ntag] and accumulates dtrans [ntag, ntag], both scaled by gout[d].  viterbi (:99-145): best tag per row,
 * path score per document; backptr [N, ntag] is workspace. */
int vbg_crf_nll_fwd(const float* emissions, const int* tags, const int* doc_off, int ndoc, const float* trans, int ntag,
                    int start_tag, int stop_tag, float* alpha, float* logz, float* nll, void* stream);
int vbg_crf_nll_bwd(const float* emissions, const int* tags, const int* doc_off, int ndoc, const float* trans, int ntag,
                    int start_tag, int stop_tag, const float* alpha, const float* logz, const float* gout, float* demissions,
                    float* dtrans_accum, void* stream);
int vbg_crf_viterbi(const float* emissions, const int* doc_off, int ndoc, const float* trans, int ntag, int start_tag,
                    int stop_tag, int* backptr, int* path, float* score, void* stream);

/* ------------------------------------------------------------------------------------------
 * a15. optimizers on flat fp32 ranges  (torch.optim.SGD / AdamW; train_SROIE.py:223-235)
 * ------------------------------------------------------------------------------------------ */
int vbg_sgd_step(float* p, const float* g, float* mom, long long n, float lr, float momentum, float wd,
                 int first_step, float grad_scale, void* stream);
int vbg_adamw_step(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2,
                   float eps, float wd, int step, float grad_scale, void* stream);
/* out[0] += sum(g^2) */
int vbg_sumsq(const float* g, long long n, float* out_accum, void* stream);
int vbg_scale_inplace(float* x, long long n, float s, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VBG_H */
