"""Thin torch-tensor -> raw-pointer shims over the C-ABI (vbg/lib.py).  PyTorch is used for device
memory and streams only; every function below launches hand-written HIP kernels on the CURRENT
stream and never synchronises.  No fallbacks: tensors must live on a GPU."""
import ctypes as C

import os

import torch

from .lib import (EPI_GELU_DUAL, EPI_NONE, EPI_RELU, OP_CONV_K, OP_CONV_R, OP_DENSE_K, OP_DENSE_R, OP_WT_R, ConvGeo,
                  AttnDesc, BertLayerFwdDesc, GemmDesc, PlaneGemmDesc, check, lib)

f32 = torch.float32
i32 = torch.int32


def raw_stream(device=None) -> int:
    """hipStream_t of the current stream of `device` (default: the current device) as an integer.  Two C calls: torch.cuda.current_stream()
    resolves the device through several Python layers and an environment lookup, ~9 us a call -- 200 calls of a single-document
    inference, 1500 of a training step"""
    idx = None if device is None else torch.device(device).index
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice() if idx is None else idx)


def _stream():
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


# ----------------------------------------------------------------------------------------------
# Second compute stream.  The text encoder and the part of the CNN in front of the early fusion are independent, forward and
# backward; enqueued on two HIP streams the workgroups of one fill the CUs the other leaves idle (a 198-tile plane product
# covers 0.77 of the chip, a grouped weight-gradient launch 0.84).  `VBG_OVERLAP=0` puts everything back on one stream.
# ----------------------------------------------------------------------------------------------
_OVERLAP = [os.environ.get("VBG_OVERLAP", "1") != "0"]
_SIDE = {}


def overlap_enabled() -> bool:
    return _OVERLAP[0]


def set_overlap(on: bool):
    _OVERLAP[0] = bool(on)


_WGRAD_STREAM = [os.environ.get("VBG_WGRAD_STREAM", "0") != "0"]


def wgrad_stream_enabled() -> bool:
    """the grouped weight-gradient launch of an encoder layer goes on its own stream (it depends on nothing the rest of the
    backward waits for): its 216 tiles and the next layer's data-gradient products share the chip"""
    return _WGRAD_STREAM[0]


def set_wgrad_stream(on: bool):
    _WGRAD_STREAM[0] = bool(on)


_CONV_WGRAD_STREAM = [int(os.environ.get("VBG_CONV_WGRAD_STREAM", "2"))]


_CONV_WGRAD_STREAM_MAXPIX = [int(os.environ.get("VBG_CONV_WGRAD_STREAM_MAXPIX", "0"))]


def conv_wgrad_stream_enabled(level: int = 1, pixels: int = 0) -> bool:
    """weight gradients of the conv + BatchNorm nodes on a stream of their own, beside the input gradient of the same node (level 2:
    those of the plain convolution nodes -- FPN, heads -- as well); pixels: B * H * W of the node's output (VBG_CONV_WGRAD_STREAM_MAXPIX > 0:
    only nodes with at most that many)"""
    if _CONV_WGRAD_STREAM_MAXPIX[0] > 0 and pixels > _CONV_WGRAD_STREAM_MAXPIX[0]:
        return False
    return int(_CONV_WGRAD_STREAM[0]) >= level


_HEADS_STREAM = [os.environ.get("VBG_HEADS_STREAM", "1") != "0"]          # round 6: +2.3 % at cfg2 (A/B x 3 on one box: 31.26 -> 30.56 ms)


def heads_stream_enabled() -> bool:
    """the RoI / field-type branch (RoIAlign, region-map convolutions, late fusion, classifier, its losses) on a stream of its own beside
    the segmentation head -- both start from P_fuse and meet again in the loss sum, forward and backward (same conditions as VBG_OVERLAP)"""
    return _HEADS_STREAM[0]


def set_heads_stream(on: bool):
    _HEADS_STREAM[0] = bool(on)


_STAGE1_BWD_AFTER = [int(os.environ.get("VBG_STAGE1_BWD_AFTER", "6"))]          # round 6: +0.5 % at cfg2 (A/B x 2 over n = -1, 0, 2 ... 10: profiles/r06_stage1_bwd_order.txt)


def stage1_bwd_after() -> int:
    """how many of the encoder's TOP layers run their backward before the backward of the CNN's first stage is enqueued (-1: the
    autograd engine's own order -- every encoder layer first, because the encoder's nodes are younger; see ViBERTgridNet._features)"""
    return _STAGE1_BWD_AFTER[0]


def side_stream(device, name: str = "side") -> "torch.cuda.Stream":
    """the side stream `name` of `device` (created on first use)"""
    idx = torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    s = _SIDE.get((idx, name))
    if s is None:
        s = _SIDE[(idx, name)] = torch.cuda.Stream(device=idx)
    return s


_RESERVE_AMAX = [os.environ.get("VBG_AMAX_RESERVE", "1") != "0"]


def reserve_for(stream, *tensors):
    """every operand a launch on `stream` reads that was allocated under ANOTHER stream: the caching allocator must not hand the block
    to anybody else before that launch has run (record_stream marks the whole block, so a slot view reserves its pool).
    VBG_AMAX_RESERVE=0 leaves the amax slots (int32 views) out -- the round-5 behaviour, kept ONLY as the A/B that names the root
    cause of the intermittent weight-gradient mismatch (tools/stream_race_check.py --amax-pool 16)"""
    for t in tensors:
        if t is not None and (_RESERVE_AMAX[0] or t.dtype != torch.int32):
            t.record_stream(stream)


def side_streams():
    """every side stream created so far: whoever consumes results of the whole backward on another stream (the gradient
    all-reduce, vbg/optim.FlatReducer) waits for these as well"""
    return list(_SIDE.values())


_ASYNC_H2D = [os.environ.get("VBG_ASYNC_H2D", "1") != "0"]


def h2d(host, device):
    """numpy array / CPU tensor -> device tensor through pinned staging memory and an ASYNCHRONOUS copy.  A pageable `.to(device)`
    returns only when the copy has run, i.e. after everything enqueued before it: one such upload in the middle of the forward costs
    the host its whole lead over the device (measured: 30-200 us idle gaps behind every one of them, tools/gap_report.py)"""
    t = torch.from_numpy(host) if not torch.is_tensor(host) else host
    device = torch.device(device)
    if device.type != "cuda" or not _ASYNC_H2D[0]:
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


def _untag(t):
    """a kernel of this library is about to write INTO t through its raw pointer (accumulate launches): torch's version counter does not
    move, so an amax tag a producer left on the tensor object (vbg.functions._amax_tag validates tags by `_version`) would survive a
    change of max |t| -- the tag is dropped here instead; the producer that completes the tensor sets a new one"""
    if t is not None:
        t.__dict__.pop("_vbg_amax", None)


def P(t):
    if t is None:
        return None
    assert t.is_cuda, "libvbg operates on device memory only"
    return C.c_void_p(t.data_ptr())


def _vec_ok(t_ptr: int, ld: int) -> int:
    return int(t_ptr % 16 == 0 and ld % 4 == 0)


def _chk_f32(*ts):
    for t in ts:
        if t is not None:
            assert t.is_cuda and t.dtype == f32, (t.device, t.dtype)


# ----------------------------------------------------------------------------------------------
# GEMM family
# ----------------------------------------------------------------------------------------------
_GEMM_F16 = [os.environ.get("VBG_GEMM_F16", "1") != "0"]          # round 6: the generic kernels' forward products on two fp16 pieces (0: three bf16 pieces)


def set_gemm_f16(on: bool):
    _GEMM_F16[0] = bool(on)


def gemm_raw(M, N, K, A, lda, a_kind, B, ldb, b_kind, Cout, ldc, *, bias=None, epi=EPI_NONE, C2=None, accumulate=False,
             splitk=1, geo=None, segs=None, a_hw=(0, 0), a_relu_scale=None, grp=None, ngroups=0, grp_max=(0, 0), tile=0,
             alpha=1.0, a_ptr_off=0, b_ptr_off=0, c_ptr_off=0, bk=0, stats=None, slab_stride=0, f16=False):
    """A, B, Cout: tensors (their data_ptr + element offsets are used).  stats: BatchNorm slot workspace [slots*2N] fp64 that receives the
    per-column sum / sum of squares of the output (fused into the epilogue; the launch is then never split)."""
    d = GemmDesc()
    d.M, d.N, d.K = int(M), int(N), int(K)
    ap = A.data_ptr() + 4 * a_ptr_off
    bp = B.data_ptr() + 4 * b_ptr_off
    d.A, d.lda, d.a_kind = ap, int(lda), a_kind
    d.B, d.ldb, d.b_kind = bp, int(ldb), b_kind
    d.a_vec = _vec_ok(ap, lda)
    d.b_vec = _vec_ok(bp, ldb)
    if segs:
        d.a_nseg = len(segs)
        vec = 1
        for i, (t, kend, ld, sh) in enumerate(segs):
            d.a_seg_ptr[i] = t.data_ptr()
            d.a_seg_kend[i] = int(kend)
            d.a_seg_ld[i] = int(ld)
            d.a_seg_shift[i] = int(sh)
            vec &= _vec_ok(t.data_ptr(), ld)
        d.a_vec = vec
        d.a_H, d.a_W = int(a_hw[0]), int(a_hw[1])
    if a_relu_scale is not None:
        d.a_prologue, d.a_scale = 1, float(a_relu_scale)
    if geo is not None:
        d.geo = geo
    d.C, d.ldc = Cout.data_ptr() + 4 * c_ptr_off, int(ldc)
    d.C2 = None if C2 is None else C2.data_ptr() + 4 * c_ptr_off
    d.bias = None if bias is None else bias.data_ptr()
    if stats is not None:
        d.stats, d.stats_slots = stats.data_ptr(), bn_slots()
    if splitk == 1 and not accumulate and torch.is_grad_enabled() and stats is None:
        splitk = 0          # training: let the library split few-tile / long-K products; no_grad (inference, eval) stays bit-reproducible
    if splitk == 0 and accumulate:
        splitk = 1
    if accumulate:
        _untag(Cout)
    d.epi, d.alpha, d.accumulate, d.splitk, d.tile = epi, float(alpha), int(bool(accumulate)), int(splitk), int(tile)
    d.bk = int(bk)
    d.slab_stride = int(slab_stride)          # (> 0: split s stores its partial product at Cout + s * slab_stride; slab_reduce adds them)
    if _AMP[0] or _SPLIT3[0]:
        # f16: a FORWARD product (activations x weights, inside fp16's range): two fp16 pieces per operand, three piece products (include/vbg.h
        # vbg_gemm_desc.bf16 = 2, round 6) -- half the matrix-core work of the six-product form; the library runs it for the forward kinds only
        d.bf16 = 1 if _AMP[0] else (2 if (f16 and _GEMM_F16[0] and grp is None) else 3)
        if _DISPATCH[0] is not None and not _AMP[0] and grp is None:
            _seen("gemm:f16x2" if d.bf16 == 2 else "gemm:bf16x3")
        if bk == 16:
            d.bk = 0        # (the 16-deep k-tiles are an fp32-form tuning)
        if grp is not None and not _AMP[0]:
            d.bf16, d.bk = 0, int(bk)          # grouped (attention) products keep their tuned fp32 form: measured equal in the step
    if _FORCE[0] or _FORCE[1]:          # debugging / conditioning experiments: force one tile configuration
        d.tile, d.bk = _FORCE[0] or d.tile, _FORCE[1] or d.bk
    if grp is not None:
        d.grp, d.ngroups = grp.data_ptr(), int(ngroups)
        d.grp_maxM, d.grp_maxN = int(grp_max[0]), int(grp_max[1])
    prof = _GEMM_PROF
    if prof is not None and prof.match(a_kind, b_kind, grp is not None):
        e0, e1 = prof.events()
        check(lib.vbg_gemm_timed(C.byref(d), _stream(), e0, e1), "vbg_gemm_timed")
        prof.add(2.0 * M * N * K, e0, e1, 1 if (_AMP[0] or not _SPLIT3[0]) else 6)
        return
    check(lib.vbg_gemm(C.byref(d), _stream()), "vbg_gemm")


# ----------------------------------------------------------------------------------------------
# plane operands: a tensor split ONCE into three bf16 planes [3][rows][ld] (csrc/gemm_planes.hip)
# ----------------------------------------------------------------------------------------------
class Planes:
    """bf16 planes of a [rows, cols] fp32 matrix (K-contiguous along cols): `buf` int16 [3, rows, ld], ld = cols rounded up to 32"""
    __slots__ = ("buf", "rows", "cols", "ld")

    def __init__(self, buf, rows, cols, ld):
        self.buf, self.rows, self.cols, self.ld = buf, rows, cols, ld

    @property
    def plane(self):
        return self.buf.stride(0)

    def col_block(self, c0, ncols):
        """the planes of columns [c0, c0 + ncols) as an operand of their own (same rows, row stride and plane stride)"""
        assert c0 % 8 == 0 and c0 + ncols <= self.ld
        return Planes(self.buf[:, :, c0:c0 + ncols], self.rows, int(ncols), self.ld)


def _ld32(n):
    return (int(n) + 31) // 32 * 32


def planes_empty(rows, cols, device):
    ld = _ld32(cols)
    return Planes(torch.empty((3, rows, ld), device=device, dtype=torch.int16), int(rows), int(cols), ld)


def split_planes(x, relu=False, out=None, colsum_out=None):
    """x [rows, cols] fp32 (last dim contiguous) -> Planes of x (rows = GEMM rows, cols = reduction index).
    colsum_out [cols] fp32: += the column sums of x in the same pass (bias gradients)."""
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == f32
    rows, cols = x.shape
    o = out if out is not None else planes_empty(rows, cols, x.device)
    check(lib.vbg_split_planes(P(x), x.stride(0), rows, cols, P(o.buf), o.ld, o.plane, int(relu), P(colsum_out), _stream()), "vbg_split_planes")
    return o


def pair_empty(rows, cols, device):
    """fp16-pair planes [2][rows][ld] (hi = fp16(x), lo' = fp16((x - hi) * 2^11)) of a [rows, cols] matrix"""
    ld = _ld32(cols)
    return Planes(torch.empty((2, rows, ld), device=device, dtype=torch.int16), int(rows), int(cols), ld)


def split_planes_pair(x, out=None, amax_slot_=None, colsum_out=None):
    """x [rows, cols] fp32 -> fp16-pair Planes (operands of plane_gemm(form=1)).  amax_slot_: scale x by the power of two of that slot
    first (a gradient operand; pass the same slot as plane_gemm(a_amax=...)); colsum_out [cols]: += column sums of x (bias gradients)"""
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == f32
    rows, cols = x.shape
    o = out if out is not None else pair_empty(rows, cols, x.device)
    check(lib.vbg_split_planes_pair(P(x), x.stride(0), rows, cols, P(o.buf), o.ld, o.plane, P(amax_slot_), P(colsum_out), _stream()), "vbg_split_planes_pair")
    return o


_PAIR = [os.environ.get("VBG_PAIR", "1") != "0"]


_PAIR_FORCE = [os.environ.get("VBG_PAIR_FORCE", "0") != "0"]


def set_pair(on: bool, force: bool = False):
    """forward BERT linears (QKV, FFN1, FFN2) on two fp16 pieces per operand / three piece products (csrc/gemm_planes.hip FORM 1)
    instead of three bf16 pieces / six: operands are LayerNorm / GELU outputs and weights, inside fp16's range.  force: also for
    problems too small to fill the 8-wave tiles the form exists for (parity tests at batch 2 run the arithmetic of batch 8)"""
    _PAIR[0] = bool(on)
    _PAIR_FORCE[0] = bool(force)


def pair_enabled() -> bool:
    return _PAIR[0] and _PLANES[0] and _SPLIT3[0] and not _amp_generic()


_PAIR_BWD = [os.environ.get("VBG_PAIR_BWD", "1") != "0"]


def set_pair_bwd(on: bool):
    """the BERT backward products (data gradients, grouped weight gradients) and the attention-output projection on two fp16 pieces as
    well: gradient operands are split by a pass of their own after their largest magnitude is known (amax slots), scaled by its power
    of two; the activations saved for backward are pair planes (4 instead of 6 bytes per element)"""
    _PAIR_BWD[0] = bool(on)


_BOUND_PLANES = [os.environ.get("VBG_BOUND_PLANES", "1") != "0"]


def bound_planes_enabled() -> bool:
    """gradient pair planes straight from the producing GEMM's epilogue, scaled by a bound instead of a measured maximum (no split pass)"""
    return _BOUND_PLANES[0]


def pair_bwd_enabled() -> bool:
    return pair_enabled() and _PAIR_BWD[0]


_PAIR_SMALL = [os.environ.get("VBG_PAIR_SMALL", "1") != "0"]


def set_pair_small(on: bool):
    """forward-only encoder layers (inference, validation) below the 8-wave tiles on the fp16-pair form's 64 x 64 tile instead of the
    six-product bf16 form (`VBG_PAIR_SMALL=0`: the A/B)"""
    _PAIR_SMALL[0] = bool(on)


def pair_small_enabled() -> bool:
    return _PAIR_SMALL[0]


def pair_tile(M, N, wide=False):
    """tile of a form-1 product [M, N], or 0 when the problem is too small for the form (it then runs the bf16 form)"""
    t = _dense_tile(M, N, wide)
    if t == 64064:
        return 128129 if _PAIR_FORCE[0] else 0
    return t


def split_planes_t_batched(src_flat, dst_planes, tbl_dev, njobs, total_tiles):
    """transposed planes of many matrices of one fp32 buffer in one launch (table layout: include/vbg.h)"""
    check(lib.vbg_split_planes_t_batched(P(src_flat), P(dst_planes), P(tbl_dev), int(njobs), int(total_tiles), dst_planes.stride(0), _stream()),
          "vbg_split_planes_t_batched")


def split_planes_pair_t_batched(src_flat, dst_planes, tbl_dev, njobs, total_tiles):
    """the same as fp16-pair planes (W^T operands of the form-1 data gradients)"""
    check(lib.vbg_split_planes_pair_t_batched(P(src_flat), P(dst_planes), P(tbl_dev), int(njobs), int(total_tiles), dst_planes.stride(0), _stream()),
          "vbg_split_planes_pair_t_batched")


def split_planes_t(x, out=None):
    """x [rows, cols] fp32 -> Planes of x^T ([cols, rows]: the reduction index becomes x's row index)"""
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == f32
    rows, cols = x.shape
    o = out if out is not None else planes_empty(cols, rows, x.device)
    check(lib.vbg_split_planes_t(P(x), x.stride(0), rows, cols, P(o.buf), o.ld, o.plane, _stream()), "vbg_split_planes_t")
    return o


def plane_gemm(a: Planes, b: Planes, out=None, *, bias=None, epi=EPI_NONE, C2=None, accumulate=False, splitk=1, alpha=1.0, tile=0,
               out_planes=None, ldc=None, trans=False, colsum_out=None, form=0, out_pair=None, a_amax=None, c_amax=None,
               q_ref_in=None, q_l1=None, q_mul=1.0, q_ref_out=None):
    """out[M, N] (+)= alpha * a[M, K] b[N, K]^T (+ bias).  out_planes: Planes [M, N] that receive the split of the stored value.
    trans: out[Ma, Nb] (+)= alpha * a[K, Ma]^T b[K, Nb] (the operands' ROWS are the reduction index: weight gradients)."""
    d = PlaneGemmDesc()
    if trans:
        assert a.rows == b.rows, (a.rows, b.rows)
        d.M, d.N, d.K, d.trans = a.cols, b.cols, a.rows, 1
    else:
        assert a.cols == b.cols, (a.cols, b.cols)
        d.M, d.N, d.K = a.rows, b.rows, _ld32(a.cols)
    d.A, d.a_plane, d.lda = a.buf.data_ptr(), a.plane, a.ld
    d.B, d.b_plane, d.ldb = b.buf.data_ptr(), b.plane, b.ld
    if out is not None:
        d.C, d.ldc = out.data_ptr(), int(ldc if ldc is not None else out.stride(-2))
    else:
        d.ldc = (d.N + 3) // 4 * 4
    d.C2 = None if C2 is None else C2.data_ptr()
    d.bias = None if bias is None else bias.data_ptr()
    if out_planes is not None:
        d.Cp, d.c_plane, d.ldp = out_planes.buf.data_ptr(), out_planes.plane, out_planes.ld
    if accumulate:
        _untag(out)
    d.epi, d.alpha, d.accumulate, d.splitk, d.tile = epi, float(alpha), int(bool(accumulate)), int(splitk), int(tile)
    if form:                               # a, b: fp16-pair planes [2][rows][ld]; a_amax: the slot a's planes were scaled by
        assert a.buf.shape[0] == 2 and b.buf.shape[0] == 2
        d.form = 2 if _AMP[0] else 1       # (autocast region: the hi planes only, one product)
        d.a_amax = None if a_amax is None else a_amax.data_ptr()
    else:
        assert a.buf.shape[0] == 3 and b.buf.shape[0] == 3
    if c_amax is not None:                 # amax slot that receives max |stored value|
        d.c_amax = c_amax.data_ptr()
    if out_pair is not None:               # the stored value also as fp16-pair planes
        d.Cq, d.q_plane, d.ldq = out_pair.buf.data_ptr(), out_pair.plane, out_pair.ld
        if q_ref_in is not None:           # ... of a GRADIENT: scaled by the power of two of a bound (include/vbg.h cq_ref_in), not of a measured maximum
            d.cq_ref_in, d.cq_mul = q_ref_in.data_ptr(), float(q_mul)
            d.cq_l1_in = None if q_l1 is None else q_l1.data_ptr()
            d.cq_ref_out = None if q_ref_out is None else q_ref_out.data_ptr()
    if colsum_out is not None:             # += column sums of the stored values (a bias gradient)
        d.colsum = colsum_out.data_ptr()
    if _DISPATCH[0] is not None:
        _seen(("plane_gemm:onep" if _AMP[0] else "plane_gemm:pair") if form else "plane_gemm:bf16x3")
        _seen(f"plane_gemm:tile{int(tile)}")
    if _STREAMK[0] and not form and c_amax is None and colsum_out is None and not trans and splitk == 1 and tile in (128129, 128130):
        ws, cnt, ncu = _sk_workspace(a.buf.device)
        d.sk_ws, d.sk_cnt, d.sk_blocks = ws.data_ptr(), cnt.data_ptr(), ncu
    prof = _GEMM_PROF
    if prof is not None and not trans and prof.match(OP_DENSE_K, OP_DENSE_K, False):
        e0, e1 = prof.events()
        check(lib.vbg_plane_gemm_timed(C.byref(d), _stream(), e0, e1), "vbg_plane_gemm_timed")
        bm, bn = {256128: (256, 128), 64064: (64, 64), 128064: (128, 64)}.get(int(d.tile), (128, 128)) if (form or int(d.tile)) else (0, 0)
        npl = (1 if _AMP[0] else 2) if form else 3
        ing = (-(-d.M // bm)) * (-(-d.N // bn)) * float(d.K) * (bm + bn) * npl * 2 if bm else None
        prof.add(2.0 * d.M * d.N * (a.rows if trans else a.cols), e0, e1, (1 if _AMP[0] else 3) if form else 6, ingest=ing)
        return out
    check(lib.vbg_plane_gemm(C.byref(d), _stream()), "vbg_plane_gemm")
    return out


_STREAMK = [os.environ.get("VBG_STREAMK", "0") != "0"]          # off: measured slower than the plain rounds at the BERT shapes (csrc/gemm_planes.hip)
_SK_WS = {}


def set_streamk(on: bool):
    """stream-K tail of the NT plane products (csrc/gemm_planes.hip plane_gemm_sk_kernel)"""
    _STREAMK[0] = bool(on)


def _sk_workspace(device):
    """(slabs, zeroed tile counters, CU count) of the stream-K tail: one set per (device, stream) -- launches on one stream are
    ordered, launches on different streams must not share slabs"""
    key = (device.index, raw_stream())
    w = _SK_WS.get(key)
    if w is None:
        ncu = torch.cuda.get_device_properties(device).multi_processor_count
        w = (torch.empty((ncu * 2 * 128 * 128,), device=device, dtype=f32), torch.zeros((ncu,), device=device, dtype=i32), ncu)
        _SK_WS[key] = w
    return w


def plane_gemm_grouped(problems, *, trans=True, accumulate=True, tile=0, alpha=1.0, form=0, a_amax=None):
    """several independent plane products of ONE reduction length in one launch: problems = [(a: Planes, b: Planes, out), ...]
    (trans: out[a.cols, b.cols] (+)= a^T b, the weight gradients of a layer).  form=1: fp16-pair planes, a_amax = [slot or None per
    problem] (the scale of each problem's A operand)"""
    assert 1 <= len(problems) <= 4
    if trans and tile == 0:
        # 256 x 128 output tiles move a third fewer operand bytes per product (csrc/gemm_planes.hip: the kernels are bound by what a CU
        # ingests); worth it once they still cover most of the chip (the four weight gradients of a bert-base layer: 216 tiles)
        t256 = sum(((a.cols + 255) // 256) * ((b.cols + 127) // 128) for a, b, _ in problems)
        if t256 >= 0.6 * torch.cuda.get_device_properties(problems[0][2].device).multi_processor_count:
            tile = 256128
    d = PlaneGemmDesc()
    d.ngroups, d.trans, d.accumulate, d.tile, d.alpha, d.splitk = len(problems), int(trans), int(bool(accumulate)), int(tile), float(alpha), 1
    k = None
    for i, (a, b, out) in enumerate(problems):
        g = d.grp[i]
        g.A, g.a_plane, g.lda = a.buf.data_ptr(), a.plane, a.ld
        g.B, g.b_plane, g.ldb = b.buf.data_ptr(), b.plane, b.ld
        g.C, g.ldc = out.data_ptr(), out.stride(-2)
        if form:
            assert a.buf.shape[0] == 2 and b.buf.shape[0] == 2
            g.a_amax = None if (a_amax is None or a_amax[i] is None) else a_amax[i].data_ptr()
        if trans:
            assert a.rows == b.rows
            g.M, g.N, kk = a.cols, b.cols, a.rows
        else:
            assert a.cols == b.cols
            g.M, g.N, kk = a.rows, b.rows, _ld32(a.cols)
        assert k is None or k == kk, "grouped products share the reduction length"
        k = kk
    d.K = k
    d.form = (2 if _AMP[0] else 1) if form else 0       # (autocast region: the hi planes only, one product)
    _seen(("plane_gemm:grouped_onep" if _AMP[0] else "plane_gemm:grouped_pair") if form else "plane_gemm:grouped_bf16x3")
    check(lib.vbg_plane_gemm(C.byref(d), _stream()), "vbg_plane_gemm (grouped)")


_PLANES = [True]
_W_EPOCH = [0]


def set_planes(on: bool):
    """dense linear products from pre-split bf16 planes (csrc/gemm_planes.hip) instead of the in-kernel split of vbg_gemm"""
    _PLANES[0] = bool(on)


def planes_enabled() -> bool:
    return _PLANES[0] and _SPLIT3[0] and not _amp_generic()


_FLASH = [os.environ.get("VBG_FLASH", "1") != "0"]


def set_flash(on: bool):
    """fused attention kernels (csrc/attn.hip) for the plane path with 64-wide heads; off = grouped score GEMMs + row softmax"""
    _FLASH[0] = bool(on)


def flash_ok(hidden: int, inter: int, dh: int, maxlen: int) -> bool:
    """does an encoder layer of these dimensions run the fused attention?  ONE predicate for vbg.functions.BertLayerFn and for the
    generator's all-layer dropout-mask launch (the two hand-written copies had drifted apart by the `inter % 32` term)"""
    return planes_enabled() and hidden % 32 == 0 and inter % 32 == 0 and dh == 64 and maxlen <= 512 and flash_enabled()


def flash_enabled() -> bool:
    return _FLASH[0]


_HOME = [os.environ.get("VBG_HOME", "1") != "0"]


def set_home(on: bool):
    """ViBERTgridNet homes its trainable parameters in flat storage at its first training forward (vbg.optim.home_parameters); off =
    parameters stay where the caller put them unless an optimizer of vbg.optim moves them (the A/B, and the generic per-parameter route)"""
    _HOME[0] = bool(on)


def home_enabled() -> bool:
    return _HOME[0]


def bump_weight_epoch():
    """parameters were changed by something torch's version counters do not see (the fused optimizer kernels): cached weight
    planes are stale"""
    _W_EPOCH[0] += 1


def weight_planes(owner, transposed=False, view=None, also=(), pair=False) -> Planes:
    """planes of a 2-D weight [N, K] (transposed: of w^T, the B operand of the data-gradient product), split once per weight
    version.  The cache lives ON the parameter object `owner` (it dies with it: a recycled device address can never serve another
    model's planes); `view`: the matrix to split when it is not `owner` itself (the stacked q/k/v view that starts at `owner`).
    Stale when the optimizer kernels ran (epoch), torch updated the tensor in place (`_version`) or it moved (data_ptr)."""
    w = owner if view is None else view
    assert w.dim() == 2 and w.stride(1) == 1 and w.stride(0) == w.shape[1]
    flat = getattr(owner, "_vbg_flat", None)
    if flat is not None and flat[0].pflat.data_ptr() + 4 * flat[1] == w.data_ptr():
        # the parameter lives in a flat buffer (vbg/optim.FlatGroup): its planes are a view of the buffer's plane image, which one
        # launch per optimizer step refreshes for all weights
        g, off = flat
        owners = (owner,) + tuple(also)          # (their version counters tell the buffer's images about torch's in-place updates)
        if pair:
            pl = g.planes_t_of(off, w.shape[0], w.shape[1], owners, pair=True) if transposed else g.pair_of(off, w.shape[0], w.shape[1], owners)
        else:
            pl = g.planes_t_of(off, w.shape[0], w.shape[1], owners) if transposed else g.planes_of(off, w.shape[0], w.shape[1], owners)
        if pl is not None:
            return pl
    cache = owner.__dict__.setdefault("_vbg_wplanes", {})
    key = (tuple(w.shape), bool(transposed), bool(pair))
    tag = (_W_EPOCH[0], owner._version, w.data_ptr()) + tuple(t._version for t in also)      # `also`: the other tensors a stacked view covers
    hit = cache.get(key)
    if hit is not None and hit[0] == tag:
        return hit[1]
    with torch.no_grad():
        buf = hit[1] if hit is not None else None
        if pair and transposed:
            pl = split_planes_pair(w.detach().t().contiguous(), out=buf)       # (no flat buffer: a plain transpose, tests only)
        else:
            pl = split_planes_pair(w.detach(), out=buf) if pair else (split_planes_t(w.detach(), out=buf) if transposed else split_planes(w.detach(), out=buf))
    cache[key] = (tag, pl)
    return pl


def _dense_tile(M, N, wide=False):
    """tile of a forward / data-gradient plane product [M, N]: the 8-wave 128 x 128 tile (2 waves per SIMD) once it fills the
    chip, 256 x 128 for the widest outputs (fewer, fuller rounds), 64 x 64 for small problems (tests, single documents)"""
    t128 = ((M + 127) // 128) * ((N + 127) // 128)
    if t128 < 100:
        return 64064
    return 256128 if (wide and M >= 2048) else 128129


def _wgrad_tile(N, K):
    """tile of a weight-gradient plane product dW[N, K]: no split-K (float atomics run at the L2's atomic rate, ~17 us per pass
    over a 3072 x 768 gradient); the 8-wave 128 x 128 tile from 100 tiles on, 64 x 64 below"""
    return 128129 if ((N + 127) // 128) * ((K + 127) // 128) >= 100 else 64064


class GemmProfiler:
    """Optional timing of ONE GEMM variant (bench.py roofline).  Each matching launch goes through `vbg_gemm_timed`: the two
    events receive the dispatch packet's own begin / end timestamps (hipExtLaunchKernel start / stop events) on the stream the
    kernel is launched on -- the quantity rocprofv3's kernel trace reports -- without the barrier packets a hipEventRecord pair
    would put around every kernel.  flops = 2*M*N*K of that launch."""

    def __init__(self, a_kind, b_kind, grouped=False):
        self.key = (a_kind, b_kind, grouped)
        self.records = []
        self.pool = []

    def match(self, a_kind, b_kind, grouped):
        return (a_kind, b_kind, grouped) == self.key

    def events(self):
        out = []
        for _ in range(2):
            h = C.c_void_p()
            check(lib.vbg_timer_create(C.byref(h)), "vbg_timer_create")
            out.append(h)
        return out

    def add(self, flops, e0, e1, products=6, ingest=None):
        """products: MFMA piece products the launch executes per algorithmic product (6: three bf16 pieces per operand, 3: two fp16
        pieces, 1: amp / the fp32 matrix pipe).  ingest: bytes the launch's workgroups move from L2 into LDS (plane products: tiles x
        k-tiles x stage bytes), or None"""
        self.records.append((flops, e0, e1, products, ingest))

    def summary(self):
        """-> (launches, total algorithmic flops, total ms, total EXECUTED matrix-core flops)  (call after a device sync); releases
        the events"""
        ms = 0.0
        self.ingest = [0, 0.0, 0.0]              # launches with a known operand ingest, their bytes, their ms
        for _, e0, e1, _, ing in self.records:
            v = C.c_float()
            check(lib.vbg_timer_elapsed_ms(e0, e1, C.byref(v)), "vbg_timer_elapsed_ms")
            ms += v.value
            if ing is not None:
                self.ingest[0] += 1; self.ingest[1] += ing; self.ingest[2] += v.value
        out = (len(self.records), sum(r[0] for r in self.records), ms, sum(r[0] * r[3] for r in self.records))
        for _, e0, e1, _, _ in self.records:
            lib.vbg_timer_destroy(e0)
            lib.vbg_timer_destroy(e1)
        self.records = []
        return out


_GEMM_PROF = None


def set_gemm_profiler(p):
    global _GEMM_PROF
    _GEMM_PROF = p


_FORCE = [0, 0]

# amp (reference: `with torch.cuda.amp.autocast(enabled=scaler is not None)` around the model call,
# pipeline/train_val_utils.py:264): every GEMM / convolution product multiplies on the bf16 matrix cores.  Tensors stay fp32 in
# memory (the kernel rounds operands to bf16 on their way into LDS and accumulates in fp32), so nothing else changes shape or
# dtype and no loss scaling is needed for range (a GradScaler passed by the caller keeps working on the fp32 gradients).
# ViBERTgridNet.forward latches torch.is_autocast_enabled() here; the backward of that forward sees the same setting.
_AMP = [False]
# Round 4: inside an autocast region the FAST kernels run in a one-product form instead of handing the work to the generic kernels:
# the fp16-pair plane products (BERT linears, forward / data gradient / weight gradient) and the pre-split-filter row-reuse
# convolutions (forward, input gradient, weight gradient) multiply the hi pieces only -- the operand rounded to fp16, which IS the
# operand of the reference's fp16 autocast (pipeline/train_val_utils.py:264), gradients scaled into range by their amax slots as in
# the pair form -- with fp32 accumulation; everything else (1x1 convolutions, heads, stem) stays on the bf16 form of csrc/gemm.hip.
# VBG_AMP_FAST=0: every product of an autocast region on the generic kernels (rounds 1-3).
_AMP_FAST = [os.environ.get("VBG_AMP_FAST", "1") != "0"]


def set_amp_fast(on: bool):
    _AMP_FAST[0] = bool(on)


def _amp_generic() -> bool:
    """autocast region AND the fast kernels' one-product forms switched off: the generic kernels take every product"""
    return _AMP[0] and not _AMP_FAST[0]


def amp_one_product() -> bool:
    """inside an autocast region with the one-product forms of the fast kernels on"""
    return _AMP[0] and _AMP_FAST[0]


_SPLIT3 = [True]        # fp32-grade products as six bf16 piece products (exact three-way operand split), see csrc/gemm.hip


def set_precision(form: str):
    """arithmetic form of the non-amp products: "split" (default; fp32-grade, six bf16 piece products) or "fp32" (fp32 matrix
    pipe everywhere; bit-reproducible against the split form only to fp32 rounding)"""
    assert form in ("split", "fp32"), form
    _SPLIT3[0] = form == "split"


def precision() -> str:
    return "split" if _SPLIT3[0] else "fp32"


def set_amp(on: bool):
    _AMP[0] = bool(on)


def amp_enabled() -> bool:
    return _AMP[0]


class amp_scope:
    def __init__(self, on=True):
        self.on = on

    def __enter__(self):
        self.prev = _AMP[0]
        _AMP[0] = bool(self.on)

    def __exit__(self, *exc):
        _AMP[0] = self.prev


def _pick_splitk(M, N, Kred, bk=32):
    """Split the reduction of a weight-gradient GEMM so the launch has ~2048 blocks (two rounds of the 1024 resident 64x64 blocks),
    keeping >= 20 k-tiles per split (>= 16 beyond 64 splits: the atomic epilogue grows with them).  Measured on MI355X with every
    shape warmed up (round 1): e.g. 768x768 K4128: 16 splits 76 TF/s -> 6 splits 90; 3072x768: 5 -> 4 splits 110 -> 116;
    256->256 3x3 conv on 32x32 maps: 21 -> 12 splits 91 -> 97 TF/s."""
    tiles64 = ((M + 63) // 64) * ((N + 63) // 64)
    nkt = (Kred + bk - 1) // bk
    want = (2048 + tiles64 // 2) // max(tiles64, 1)
    return int(max(1, min(want, max(min(nkt // 20, 64), min(nkt // 16, 256)))))


def linear_fwd(x, w, bias=None, epi=EPI_NONE, out=None, out2=None):
    """y[M,N] = x[M,K] @ w[N,K]^T (+bias) ; rows of x may be strided (x.stride(0))."""
    _chk_f32(x, w, bias)
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K and x.stride(1) == 1 and w.stride(1) == 1
    if out is None:
        out = torch.empty((M, N), device=x.device, dtype=f32)
    if epi == EPI_GELU_DUAL and out2 is None:
        out2 = torch.empty_like(out)
    sk = slab_split(M, N, K) if (epi in (EPI_NONE, EPI_RELU) and N % 4 == 0 and out.stride(0) % 4 == 0 and _GEMM_PROF is None) else 1
    if sk > 1:
        # a handful of output tiles over a very long reduction (the field-type head's first layer on one document: 32 tiles of 416
        # k-tiles, 253 us): the reduction is cut into `sk` slabs that a second small launch adds in order -- deterministic
        slabs = torch.empty((sk, M, N), device=x.device, dtype=f32)
        gemm_raw(M, N, K, x, x.stride(0), OP_DENSE_K, w, w.stride(0), OP_DENSE_K, slabs, N, splitk=sk, slab_stride=M * N, f16=True)
        check(lib.vbg_slab_reduce(P(slabs), sk, M * N, M, N, N, P(bias), int(epi == EPI_RELU), P(out), out.stride(0), _stream()), "vbg_slab_reduce")
        _seen("gemm:slab_split")
        return out
    gemm_raw(M, N, K, x, x.stride(0), OP_DENSE_K, w, w.stride(0), OP_DENSE_K, out, out.stride(0), bias=bias, epi=epi, C2=out2, f16=True)
    return (out, out2) if epi == EPI_GELU_DUAL else out


_SLAB_SPLIT = [os.environ.get("VBG_SLAB_SPLIT", "1") != "0"]


def slab_split(M, N, K) -> int:
    """slabs a forward linear layer's reduction is cut into (1: none): at most 64 output tiles of 64 x 64 and k >= 4096, enough slabs to
    put ~256 workgroups on the chip, each at least 1024 deep, at most 8"""
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    if not _SLAB_SPLIT[0] or tiles > 64 or K < 4096 or M == 0:
        return 1
    sk = min(8, 256 // tiles, K // 1024)
    return sk if sk >= 2 else 1


def linear_dgrad(dy, w, out=None, accumulate=False):
    """dx[M,K] (+)= dy[M,N] @ w[N,K]"""
    _chk_f32(dy, w)
    M, N = dy.shape
    K = w.shape[1]
    assert w.shape[0] == N and dy.stride(1) == 1 and w.stride(1) == 1
    if out is None:
        assert not accumulate
        out = torch.empty((M, K), device=dy.device, dtype=f32)
    gemm_raw(M, K, N, dy, dy.stride(0), OP_DENSE_K, w, w.stride(0), OP_DENSE_R, out, out.stride(0), accumulate=accumulate, splitk=_BWD_SPLIT)
    return out


def linear_wgrad(dy, x, out, accumulate=True):
    """dw[N,K] (+)= dy[M,N]^T @ x[M,K]   (split over M with atomics when dw has few tiles)"""
    _chk_f32(dy, x, out)
    M, N = dy.shape
    K = x.shape[1]
    assert x.shape[0] == M and out.shape == (N, K) and dy.stride(1) == 1 and x.stride(1) == 1
    sk = _pick_splitk(N, K, M)
    if not accumulate and sk > 1:
        out.zero_()
    gemm_raw(N, K, M, dy, dy.stride(0), OP_DENSE_R, x, x.stride(0), OP_DENSE_R, out, out.stride(0),
             accumulate=(accumulate or sk > 1), splitk=sk)
    return out


def colsum(x, out=None, accumulate=False):
    _chk_f32(x)
    M, N = x.shape
    assert x.stride(1) == 1
    if out is None:
        out = torch.empty((N,), device=x.device, dtype=f32)
        accumulate = False
    if _COLSUM_F64[0]:
        key = (x.device, raw_stream(x.device))
        ws = _COLSUM_WS.get(key)
        if ws is None or ws.numel() < N:
            ws = _COLSUM_WS[key] = torch.empty((max(N, 4096),), device=x.device, dtype=torch.float64)
        check(lib.vbg_colsum_f64(P(x), x.stride(0), M, N, P(out), int(accumulate), P(ws), _stream()), "vbg_colsum_f64")
        return out
    check(lib.vbg_colsum(P(x), x.stride(0), M, N, P(out), int(accumulate), _stream()), "vbg_colsum")
    return out


# bias gradients that do not ride on a split pass are summed in fp64 (csrc/rowops.hip colsum_f64_*: the 1x1 segmentation classifiers'
# bias gradients cancel to 1e-3 of their running partial sums); VBG_COLSUM_F64=0 restores the fp32 atomics
_COLSUM_F64 = [os.environ.get("VBG_COLSUM_F64", "1") != "0"]
_COLSUM_WS = {}


def conv_geo(Hs, Ws, Cs, Hr, Wr, kh, kw, stride, pad, dgrad=0):
    g = ConvGeo()
    g.Hs, g.Ws, g.Cs, g.Hr, g.Wr, g.kh, g.kw, g.stride, g.pad, g.dgrad = Hs, Ws, Cs, Hr, Wr, kh, kw, stride, pad, dgrad
    return g


def conv_out_hw(H, W, k, stride, pad):
    return (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1


def fuse_stats_ok(M, N, K):
    """fuse the BatchNorm statistics into the producing GEMM unless that GEMM is one the library would rather split (few tiles, long
    reduction: the layer4 convolutions), mirroring the rule in csrc/gemm.hip"""
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    return N % 4 == 0 and not (tiles <= 384 and (K + 31) // 32 >= 48)


_CONV3 = [os.environ.get("VBG_CONV3", "1") != "0"]


def set_conv3(on: bool):
    """row-reuse kernel (csrc/conv3.hip) for the wide 3x3 / s1 / p1 convolutions; off = the generic implicit GEMM of csrc/gemm.hip"""
    _CONV3[0] = bool(on)


_CONV3_MIN_TILES = [int(os.environ.get("VBG_CONV3_MIN_TILES", "480"))]
# the forward in its two-piece fp16 form also wins on 64-pixel tiles once they cover the chip (256 channels at 32 x 32 pixels: 59 vs 74 us)
_CONV3_MIN_TILES_FWD = [int(os.environ.get("VBG_CONV3_MIN_TILES_FWD", "256"))]


_CONV3_ROI = [os.environ.get("VBG_CONV3_ROI", "1") != "0"]       # [N, 7, 7, C] region maps on the row-reuse kernels (two images per 128-slot tile)
_CONV3_N64 = [os.environ.get("VBG_CONV3_N64", "1") != "0"]       # 64-filter tiles (the 64-channel stage of the trunk)


def conv3_ok(B, H, W, Cs, N, kh, kw, stride, pad, fwd=False) -> bool:
    """shapes the row-reuse kernel takes: whole image rows per pixel tile, full 128-wide column tiles and at least 480 64-pixel tiles
    (= 240 of the 128-pixel tiles the kernel then uses).  Below that the kernel would run 64-pixel tiles, one 4-wave workgroup per CU:
    measured level with the generic 64 x 64 tiles (70 vs 72 us forward at 256 channels, 32 x 32 pixels) and behind them once the filter
    has to be turned for the input gradient (84 vs 74 us) -- the late stages stay on the generic kernel; default split form only.
    Also: 7 x 7 region maps (two images per tile, see csrc/conv3.hip) and filter counts that are odd multiples of 64 (64-wide tiles)"""
    if not (_CONV3[0] and _SPLIT3[0] and not _amp_generic() and kh == 3 and kw == 3 and stride == 1 and pad == 1 and Cs % 16 == 0):
        return False
    min_tiles = _CONV3_MIN_TILES_FWD[0] if fwd and _CONV3_F16[0] else _conv3_min_tiles_bwd()
    if H == 7 and W == 7:
        return _CONV3_ROI[0] and N % 128 == 0 and B * 49 * Cs < (1 << 29) and ((B + 1) // 2) * 2 * (N // 128) >= min_tiles
    if not (W in (16, 32, 64, 128, 256, 512, 1024) and (H * W) % 64 == 0 and H * W * Cs < (1 << 29)):
        return False
    if N % 128 == 0:
        return (B * H * W // 64) * (N // 128) >= min_tiles or conv3_split(B, H, W, Cs, N) > 1
    if N % 64 == 0 and _CONV3_N64[0]:
        return (H * W) % 128 == 0 and (B * H * W // 128) * (N // 64) >= 240
    return False


_CONV3_F16 = [os.environ.get("VBG_CONV3_F16", "1") != "0"]
_CONV3_MIN_TILES_BWD = [int(os.environ.get("VBG_CONV3_MIN_TILES_BWD", "0"))]


def _conv3_min_tiles_bwd():
    """64-pixel tiles an input gradient needs to take the row-reuse kernel (VBG_CONV3_MIN_TILES_BWD overrides)"""
    if _CONV3_MIN_TILES_BWD[0] > 0:
        return _CONV3_MIN_TILES_BWD[0]
    return _CONV3_MIN_TILES[0]          # (the forward's 256 was A/B-ed in the step for the fp16-form input gradient: 190.97 vs 190.74 docs/s, no gain)


def conv3_f16_enabled() -> bool:
    """the wide 3x3 convolutions' FORWARD runs the two-piece fp16 form (and consumes an activation's amax slot when the producer left one)"""
    return _CONV3_F16[0] and _CONV3[0] and _SPLIT3[0]


def set_conv3_f16(on: bool):
    """forward convolutions of csrc/conv3.hip on two fp16 pieces per operand (three piece products) instead of three bf16 pieces (six)"""
    _CONV3_F16[0] = bool(on)


_AMAX_POOL = {}
_AMAX_POOL_SLOTS = [int(os.environ.get("VBG_AMAX_POOL_SLOTS", "256"))]      # (tests shrink it to make pools turn over inside one backward)


AMAX_WORDS, AMAX_STRIDE = 64, 32          # include/vbg.h VBG_AMAX_WORDS / VBG_AMAX_STRIDE


def amax_slot(device):
    """a fresh ZERO amax slot (64 int32 words 128 bytes apart whose max is the bit pattern of a tensor's largest magnitude: vbg_amax and
    the amax outputs of bn_apply / bn_bwd_apply max INTO it).  Slots come from a zero-filled pool (one 2 MB fill launch per 256
    slots); an exhausted pool is replaced, never rewound, so a slot saved for backward stays valid.  One pool per STREAM: the fill
    of a fresh pool is ordered in front of every use of its slots by the stream itself (with one pool for all streams, a pool created
    by the encoder's side stream handed its next slots to the CNN's stream, whose kernels could run before the fill)"""
    key = (device.type, device.index, raw_stream(device) if device.type == "cuda" else 0)
    ent = _AMAX_POOL.get(key)
    if ent is None or ent[1] >= ent[0].shape[0]:
        ent = _AMAX_POOL[key] = [torch.zeros((_AMAX_POOL_SLOTS[0], AMAX_WORDS * AMAX_STRIDE), device=device, dtype=torch.int32), 0]
    i = ent[1]
    ent[1] = i + 1
    return ent[0][i]


def amax(x, slot=None):
    """bit pattern of max |x| in an amax slot (no host sync): the scale of fp16-form products whose operand x is a gradient"""
    slot = amax_slot(x.device) if slot is None else slot
    check(lib.vbg_amax(P(x), x.numel(), P(slot), _stream()), "vbg_amax")
    return slot


_CONV3_SPLITK = [os.environ.get("VBG_CONV3_SPLITK", "1") != "0"]
_CONV3_TICKETS = {}


def conv3_split(B, H, W, Cs, N) -> int:
    """workgroups per tile the row-reuse kernel wants for this shape (1: no split; csrc/conv3.hip vbg_conv3x3_split)"""
    return int(lib.vbg_conv3x3_split(B, H, W, Cs, N)) if _CONV3_SPLITK[0] else 1


def _conv3_tickets(device, tiles):
    """the arrival counters of the split form: zero between launches; one set per (device, stream) -- two split convolutions on
    different streams must not share counters"""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    t = _CONV3_TICKETS.get(key)
    if t is None or t.numel() < tiles:
        t = _CONV3_TICKETS[key] = torch.zeros(max(1024, tiles), device=device, dtype=torch.int32)
    return t


def conv3x3(x, w_ohwi, bias=None, out=None, stats=None, accumulate=False, f16x2=False, x_amax=None, nsplit=None, w_planes=None, n_out=None, bn=0):
    """y (+)= conv3x3(x NHWC, w [N,3,3,Cs]), stride 1, pad 1 (csrc/conv3.hip).  f16x2: the two-piece fp16 form -- for operands inside
    fp16's range (activations, filters); a gradient operand x needs x_amax (device word with the bits of max |x|): the kernel then
    scales x by the power of two that centres it in fp16's range and the result back (exact).  nsplit: workgroups per tile (None: the
    library's choice for the shape)"""
    B, H, W, Cs = x.shape
    N = w_ohwi.shape[0] if n_out is None else int(n_out)
    if out is None:
        assert not accumulate
        out = torch.empty((B, H, W, N), device=x.device, dtype=f32)
    elif accumulate:
        _untag(out)
    assert x_amax is None or f16x2
    assert w_planes is None or f16x2
    nz = conv3_split(B, H, W, Cs, N) if nsplit is None else int(nsplit)
    slab = tickets = None
    if nz > 1:
        tiles = (B * H * W // 128) * (N // (bn or 128))
        slab = torch.empty((tiles * nz, 128 * (bn or 128)), device=x.device, dtype=f32)
        tickets = _conv3_tickets(x.device, tiles)
    if _DISPATCH[0] is not None:
        _seen("conv3:fwd")
        if nz > 1:
            _seen("conv3:split")
        if H == 7 and W == 7:
            _seen("conv3:roi")
        if f16x2:
            _seen("conv3:f16x2")
        if bn == 64:
            _seen("conv3:bn64")
    ev = None
    if _CONV3_PROF[0] is not None:           # bench.py's second roofline object: events around the launch, on the launch stream
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    if w_planes is not None:                 # the filter as pre-split fp16-pair planes (conv3_planes): no filter work in the kernel
        _seen("conv3:pw")
        onep = _AMP[0] and _AMP_FAST[0]       # autocast region: the hi pieces only, one product
        if onep:
            _seen("conv3:onep")
        check((lib.vbg_conv3x3_pw_amp if onep else lib.vbg_conv3x3_pw)(P(x), P(w_planes), P(bias), P(out), P(stats), bn_slots() if stats is not None else 0,
                                                                     B, H, W, Cs, N, int(accumulate), P(x_amax), P(slab), P(tickets), nz, int(bn), _stream()),
              "vbg_conv3x3_pw")
    else:
        check(lib.vbg_conv3x3(P(x), P(w_ohwi), P(bias), P(out), P(stats), bn_slots() if stats is not None else 0, B, H, W, Cs, N,
                              int(accumulate), int(bool(f16x2)), P(x_amax), P(slab), P(tickets), nz, _stream()), "vbg_conv3x3")
    if ev is not None:
        ev[1].record()
        _CONV3_PROF[0].append((2.0 * B * H * W * Cs * N * 9, (1 if (w_planes is not None and _AMP[0] and _AMP_FAST[0]) else 3) if f16x2 else 6, ev[0], ev[1]))
    return out


_CONV3_PROF = [None]
_DISPATCH = [None]


def dispatch_log(on: bool):
    """on: start counting which kernel families the next calls take ({"plane_gemm:pair": n, "conv3:fwd": n, "conv3:split": n,
    "conv3:roi": n, "conv3:wgrad": n, ...}) and return the dict; off: stop.  The batch-8 parity test asserts with it that the paths
    bench.py times are the ones it held to the reference (tests/test_gpu_full_scale.py)."""
    _DISPATCH[0] = {} if on else None
    return _DISPATCH[0]


def _seen(key):
    d = _DISPATCH[0]
    if d is not None:
        d[key] = d.get(key, 0) + 1


def set_conv3_profiler(records):
    """records: a list that receives (algorithmic flops, piece products per product, start event, stop event) per row-reuse
    forward / input-gradient launch -- or None (off)"""
    _CONV3_PROF[0] = records


# ---- largest column L1 norm of weight matrices (the bound factor of a data gradient dY W, include/vbg.h cq_l1_in) -------------------------
_WL1 = {"entries": [], "table": None, "host": None, "out": None, "maxc": 0}


def weight_col_l1max(owner, view=None):
    """one device word with the float bits of max_c sum_r |w[r][c]| of the 2-D weight `owner` (or of `view`, a matrix starting at it),
    recomputed once per weight version for ALL registered matrices by one launch (+ one fill)"""
    from .lib import L1Entry
    import weakref
    w = owner if view is None else view
    assert w.dim() == 2 and w.stride(1) == 1
    st = _WL1
    slot = owner.__dict__.get("_vbg_l1")
    tag = (_W_EPOCH[0], owner._version, w.data_ptr())
    if slot is not None and slot["tag"] == tag:
        return st["out"][slot["idx"]:slot["idx"] + 1]
    if slot is None or slot["ptr"] != w.data_ptr():
        # (a parameter that MOVED -- flattened into a flat buffer after a warm-up step, .to(), a fresh .data -- registers again; its old
        #  entry leaves the table: the kernel must never read rows x cols floats from an address that may have been freed)
        st["entries"] = [e for e in st["entries"] if e["ref"]() is not owner]
        slot = {"idx": -1, "tag": None, "ptr": w.data_ptr(), "shape": tuple(w.shape), "ld": w.stride(0), "ref": weakref.ref(owner)}
        owner.__dict__["_vbg_l1"] = slot
        st["entries"].append(slot)
        st["table"] = None
    # a model went away, or another registered matrix moved since it registered: those leave the table as well (as _c3pw_refresh does);
    # a moved one registers again on its next use
    keep = []
    for e in st["entries"]:
        o = e["ref"]()
        if o is not None and (e is slot or o.data_ptr() == e["ptr"]):
            keep.append(e)
        elif o is not None:
            o.__dict__.pop("_vbg_l1", None)
    if len(keep) != len(st["entries"]):
        st["entries"] = keep
        st["table"] = None
    if st["table"] is None or st["out"] is None or st["table"].device != w.device:
        n = len(st["entries"])
        host = (L1Entry * n)()
        for i, e in enumerate(st["entries"]):
            e["idx"] = i
            host[i].w, host[i].ld, host[i].rows, host[i].cols = e["ptr"], e["ld"], e["shape"][0], e["shape"][1]
        st["host"] = host
        st["table"] = h2d(torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8), w.device)
        st["out"] = torch.zeros((max(n, 64),), device=w.device, dtype=torch.int32)
        st["maxc"] = max(e["shape"][1] for e in st["entries"])
    else:
        st["out"].zero_()
    check(lib.vbg_col_l1_max(P(st["table"]), len(st["entries"]), st["maxc"], P(st["out"]), _stream()), "vbg_col_l1_max")
    for e in st["entries"]:
        o = e["ref"]()
        if o is not None:
            e["tag"] = (_W_EPOCH[0], o._version, e["ptr"])
    return st["out"][slot["idx"]:slot["idx"] + 1]


# ---- pre-split filter images for the PW form of the row-reuse kernels (csrc/conv3.hip conv3_wprep_kernel) ------------------------------
_CONV3_PW = [os.environ.get("VBG_CONV3_PW", "1") != "0"]
_C3PW = {"entries": [], "table": None, "host": None, "n": 0}


def set_conv3_pw(on: bool):
    """filters of the fp16-form 3x3 convolutions as pre-split planes streamed by LDS-DMA (on) or split inside every workgroup (off)"""
    _CONV3_PW[0] = bool(on)


class _C3Entry:
    __slots__ = ("ref", "flip", "out", "tag", "shape", "ptr", "bn")


def _c3pw_tag(owner):
    return (_W_EPOCH[0], owner._version, owner.data_ptr())


def _c3pw_table(entries, device):
    """(host table, device table, n) describing the plane images of `entries` for vbg_conv3x3_wprep"""
    from .lib import Conv3WprepEntry
    n = len(entries)
    host = (Conv3WprepEntry * n)()
    for i, e in enumerate(entries):
        Cout, Cin = e.shape
        rows = Cin if e.flip else Cout
        host[i].w, host[i].out, host[i].Cout, host[i].Cin, host[i].flip = e.ptr, e.out.data_ptr(), Cout, Cin, int(e.flip)
        host[i].bn = e.bn if e.bn else (64 if (rows % 128 != 0 and rows % 64 == 0) else 128)
    raw = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8)
    return host, h2d(raw, device), n


def _c3pw_refresh(device):
    """every registered image that is stale (the optimizer kernels ran, load_state_dict, ...) in ONE launch; the device table is kept
    while the set of registered filters does not change"""
    st = _C3PW
    live = []
    for e in st["entries"]:
        owner = e.ref()
        if owner is not None and owner.data_ptr() == e.ptr and owner.device == device:
            live.append(e)
        elif owner is not None:
            owner.__dict__.pop("_vbg_c3pw", None)          # moved: registers again on its next use
    if len(live) != len(st["entries"]) or st["table"] is None or st["table"].device != device:
        st["entries"] = live
        st["host"], st["table"], st["n"] = _c3pw_table(live, device) if live else (None, None, 0)
    if not live:
        return
    check(lib.vbg_conv3x3_wprep(P(st["table"]), C.byref(st["host"]), st["n"], _stream()), "vbg_conv3x3_wprep")
    for e in live:
        e.tag = _c3pw_tag(e.ref())


def conv3_planes(owner, w_ohwi, flip, bn=0):
    """fp16-pair plane image of the 3x3 filter `owner` (an OIHW parameter in channels_last memory, `w_ohwi` its free [O,3,3,I] view) for
    vbg_conv3x3_pw: flip = False the forward filter, True the filter of the input gradient.  Written once per weight version for ALL
    registered filters by one launch (first use: one launch for the new image).  None: no PW form for this filter."""
    if owner is None or not _CONV3_PW[0] or not w_ohwi.is_cuda or w_ohwi.data_ptr() != owner.data_ptr():
        return None
    Cout, kh, kw, Cin = w_ohwi.shape
    if kh != 3 or kw != 3 or Cin % 8 != 0 or (Cout if flip else Cin) % 16 != 0:
        return None
    slot = owner.__dict__.setdefault("_vbg_c3pw", {})
    key = (bool(flip), int(bn))
    e = slot.get(key)
    if e is None or e.ptr != owner.data_ptr():
        import weakref
        e = _C3Entry()
        e.ref, e.flip, e.shape, e.ptr, e.bn = weakref.ref(owner), bool(flip), (int(Cout), int(Cin)), owner.data_ptr(), int(bn)
        nbytes = int(lib.vbg_conv3x3_wprep_bytes(int(Cout), int(Cin), int(bool(flip)), int(bn)))
        e.out = torch.empty((nbytes,), device=w_ohwi.device, dtype=torch.uint8)
        slot[key] = e
        st = _C3PW
        st["entries"] = [x for x in st["entries"] if not (x.ref() is owner and x.flip == e.flip and x.bn == e.bn)] + [e]
        st["table"] = None                                  # the full table is rebuilt at the next refresh
        host, tab, n = _c3pw_table([e], w_ohwi.device)     # the new image alone
        check(lib.vbg_conv3x3_wprep(P(tab), C.byref(host), n, _stream()), "vbg_conv3x3_wprep")
        e.tag = _c3pw_tag(owner)                            # (the table tensor is freed stream-ordered: the launch has it)
        return e.out
    if e.tag != _c3pw_tag(owner):
        _c3pw_refresh(w_ohwi.device)
    return e.out


def conv3x3_wflip(w_ohwi):
    """[Cout,3,3,Cin] -> [Cin,3,3,Cout] turned by 180 degrees: the filter of the input gradient"""
    Cout, _, _, Cin = w_ohwi.shape
    out = torch.empty((Cin, 3, 3, Cout), device=w_ohwi.device, dtype=f32)
    check(lib.vbg_conv3x3_wflip(P(w_ohwi), Cout, Cin, P(out), _stream()), "vbg_conv3x3_wflip")
    return out


def conv3w_ok(B, H, W, Cs, Cout, kh, kw, stride, pad) -> bool:
    """shapes the row-reuse weight-gradient kernel takes (csrc/conv3.hip): default split form, strips of at least 8 k-tiles"""
    roi = H == 7 and W == 7 and _CONV3_ROI[0]              # region maps: four two-row chunks per image
    if not (_CONV3[0] and _SPLIT3[0] and not _amp_generic() and kh == 3 and kw == 3 and stride == 1 and pad == 1 and (W % 16 == 0 or roi)
            and Cs % 32 == 0 and (Cout % 128 == 0 or (_CONV3W_N64[0] and Cout % 64 == 0 and Cs % 64 == 0))
            and (3 * W + 18) * Cs < (1 << 28)):
        return False
    strips = int(lib.vbg_conv3x3_wgrad_strips(B, H, W, Cs, Cout))
    return strips >= 1 and (B * 4 if roi else B * H * W // 16) // strips >= _CONV3W_MIN[0]


_CONV3W_MIN = [int(os.environ.get("VBG_CONV3W_MIN", "8"))]
# odd multiples of 64 filters (the 64 -> 64 convolutions of the first trunk stage) on the row-reuse weight-gradient kernel's [64 x 9 x 64]
# blocks in the fp16 form instead of the generic fp32-MFMA product: 116 -> 40 us + a 9.5 us strip reduction per layer at cfg2 (round 5;
# until the strips were reduced in parallel -- conv3_wgrad_reduce_par_kernel -- the 256 slabs of a 9216-float4 dW took 64 us to add and
# the generic kernel was as fast).  VBG_CONV3W_N64=0: the generic product
_CONV3W_N64 = [os.environ.get("VBG_CONV3W_N64", "1") != "0"]


def conv3x3_wgrad(dy, x, dw_ohwi, slabs=True, f16x2=False, dy_amax=None, x_amax=None):
    """dw += conv3x3 weight gradient (stride 1, pad 1); slabs: deterministic strip reduction through scratch, else float atomics.
    f16x2: two fp16 pieces per operand / three piece products, both operands scaled by their largest magnitudes (device words
    dy_amax / x_amax, taken with vbg_amax passes when the producers did not write them)"""
    B, H, W, Cs = x.shape
    Cout = dy.shape[3]
    slab = None
    if slabs:
        strips = int(lib.vbg_conv3x3_wgrad_strips(B, H, W, Cs, Cout))
        slab = torch.empty((strips, dw_ohwi.numel()), device=x.device, dtype=f32)
    if f16x2:
        dy_amax = amax(dy) if dy_amax is None else dy_amax
        x_amax = amax(x) if x_amax is None else x_amax
    _seen("conv3:wgrad")
    form = (2 if (_AMP[0] and _AMP_FAST[0]) else 1) if f16x2 else 0
    if form == 2:
        _seen("conv3:wgrad_onep")
    check(lib.vbg_conv3x3_wgrad(P(dy), P(x), P(dw_ohwi), P(slab), B, H, W, Cs, Cout, form, P(dy_amax), P(x_amax), _stream()),
          "vbg_conv3x3_wgrad")
    return dw_ohwi


def conv2d_fwd(x, w_ohwi, stride, pad, bias=None, out=None, stats=None, w_owner=None, x_amax=None):
    """x NHWC [B,H,W,Cin] contiguous; w_ohwi [Cout,kh,kw,Cin] contiguous -> y NHWC [B,Ho,Wo,Cout]."""
    _chk_f32(x, w_ohwi, bias)
    B, H, W, Cin = x.shape
    Cout, kh, kw, _ = w_ohwi.shape
    assert x.is_contiguous() and w_ohwi.is_contiguous() and w_ohwi.shape[3] == Cin and Cin % 16 == 0
    Ho, Wo = conv_out_hw(H, W, kh, stride, pad)
    if out is None:
        out = torch.empty((B, Ho, Wo, Cout), device=x.device, dtype=f32)
    M, K = B * Ho * Wo, kh * kw * Cin
    if conv3_ok(B, H, W, Cin, Cout, kh, kw, stride, pad, fwd=True):
        late = conv3_late_choice(B, H, W, Cin, Cout) if _CONV3_F16[0] else None
        if late is not None:
            wp = conv3_planes(w_owner, w_ohwi, False, bn=late[0])
            if wp is not None:
                return conv3x3(x, w_ohwi, bias, out, stats, f16x2=True, w_planes=wp, x_amax=x_amax, nsplit=late[1], bn=late[0])
        wp = conv3_planes(w_owner, w_ohwi, False) if (_CONV3_F16[0] and conv3_pw_ok(B, H, W, Cin, Cout)) else None
        # x_amax: the producer's amax slot of the activation, when it published one (bn_apply in a training step).  The fp16 form then
        # scales x by the power of two that centres its largest magnitude in fp16's range, like a gradient operand: an activation of
        # 65520 or more no longer becomes inf, and small activations keep more bits -- exact either way
        return conv3x3(x, w_ohwi, bias, out, stats, f16x2=_CONV3_F16[0], w_planes=wp, x_amax=x_amax if _CONV3_F16[0] else None)
    if kh == 1 and kw == 1 and stride == 1 and pad == 0:
        gemm_raw(M, Cout, K, x, Cin, OP_DENSE_K, w_ohwi, K, OP_DENSE_K, out, Cout, bias=bias, stats=stats, f16=True)
    else:
        gemm_raw(M, Cout, K, x, Cin, OP_CONV_K, w_ohwi, K, OP_DENSE_K, out, Cout, bias=bias,
                 geo=conv_geo(H, W, Cin, Ho, Wo, kh, kw, stride, pad, 0), stats=stats, f16=True)
    return out


_BWD_SPLIT = 0          # backward products run with grad mode off; they are training-only, so the library may always split them


_CONV3_F16_BWD = [os.environ.get("VBG_CONV3_F16_BWD", "1") != "0"]


def set_conv3_f16_bwd(on: bool):
    """input gradients of the wide 3x3 convolutions in the two-piece fp16 form, dy scaled by the power of two derived from its largest
    magnitude (exact; csrc/conv3.hip); off: three bf16 pieces / six piece products"""
    _CONV3_F16_BWD[0] = bool(on)


def conv3_f16_bwd_enabled() -> bool:
    return _CONV3[0] and _SPLIT3[0] and not _amp_generic() and _CONV3_F16[0] and _CONV3_F16_BWD[0]


def conv3_f16_bwd_ok(B, H, W, Cout, Cin, kh, kw, stride, pad) -> bool:
    """will conv2d_dgrad of this convolution run the fp16 form (and want the largest magnitude of dy)?"""
    return _CONV3_F16[0] and _CONV3_F16_BWD[0] and conv3_ok(B, H, W, Cout, Cin, kh, kw, stride, pad)


_CONV3_BN64 = [os.environ.get("VBG_CONV3_BN64", "1") != "0"]


def conv3_late_choice(B, H, W, Cs, N):
    """(filters per tile, workgroups per tile) for the PW kernels on the LATE trunk stages -- 16-239 tiles of 128 x 128 --, or None.
    64-filter tiles double the tile count: 256 channels at 32 x 32 then fill the chip with ONE workgroup per tile (no slabs, no tickets:
    37 vs 48 us with three workgroups per 128-filter tile), 512 channels at 16 x 16 take four per tile on half the slab bytes (42 vs 49 us;
    tools/conv3_bn_sweep.py)"""
    if not (_CONV3_BN64[0] and _CONV3_PW[0] and _CONV3_SPLITK[0]) or (H == 7 and W == 7) or N % 128 or (H * W) % 128 or W >= 128 or Cs % 16:
        return None
    t128 = (B * H * W // 128) * (N // 128)
    if t128 >= 240 or t128 < 4:          # (round 6: from 4 tiles on -- a single document's last stage)
        return None
    t64 = 2 * t128
    nz = 1 if t64 >= 240 else min(4, -(-480 // t64))
    while nz > 1:
        cs = nz // 3 if nz % 3 == 0 else nz
        if Cs % cs == 0 and (Cs // cs) % 16 == 0:
            break
        nz -= 1
    return 64, nz


def conv3_pw_ok(B, H, W, Cs, N) -> bool:
    """does vbg_conv3x3 run 128-pixel tiles on this shape (what the PW kernels exist for)?  Mirrors csrc/conv3.hip conv3x3_impl."""
    if H == 7 and W == 7:
        return True
    if (H * W) % 128 != 0:
        return False
    if conv3_split(B, H, W, Cs, N) > 1 or W >= 128:
        return True
    bn = 64 if (N % 128 != 0 and N % 64 == 0) else 128
    return (B * H * W // 128) * ((N + bn - 1) // bn) >= 240


def conv2d_dgrad(dy, w_ohwi, x_shape, stride, pad, out=None, accumulate=False, dy_amax=None, w_owner=None):
    """dx NHWC [B,H,W,Cin] (+)= conv_transpose(dy NHWC [B,Ho,Wo,Cout], w).  dy_amax: device word with the bits of max |dy| when the
    producer of dy wrote one (bn_bwd_apply); otherwise the fp16 form takes it with one vbg_amax pass over dy"""
    _chk_f32(dy, w_ohwi)
    B, H, W, Cin = x_shape
    Cout, kh, kw, _ = w_ohwi.shape
    Bq, Ho, Wo, Cq = dy.shape
    assert dy.is_contiguous() and w_ohwi.is_contiguous() and Cq == Cout and Bq == B
    if out is None:
        assert not accumulate
        out = torch.empty(x_shape, device=dy.device, dtype=f32)
    M = B * H * W
    if conv3_ok(B, H, W, Cout, Cin, kh, kw, stride, pad):
        # the input gradient of a 3x3 / s1 / p1 convolution is the same convolution of dy with the turned, channel-swapped filter
        if _CONV3_F16[0] and _CONV3_F16_BWD[0]:
            late = conv3_late_choice(B, H, W, Cout, Cin)
            if late is not None:
                wp = conv3_planes(w_owner, w_ohwi, True, bn=late[0])
                if wp is not None:
                    return conv3x3(dy, w_ohwi, None, out, None, accumulate, f16x2=True, x_amax=dy_amax if dy_amax is not None else amax(dy),
                                   w_planes=wp, n_out=Cin, nsplit=late[1], bn=late[0])
            wp = conv3_planes(w_owner, w_ohwi, True) if conv3_pw_ok(B, H, W, Cout, Cin) else None
            if wp is not None:       # the turned filter exists as plane image only: no fp32 copy of it is written
                return conv3x3(dy, w_ohwi, None, out, None, accumulate, f16x2=True, x_amax=dy_amax if dy_amax is not None else amax(dy),
                               w_planes=wp, n_out=Cin)
            return conv3x3(dy, conv3x3_wflip(w_ohwi), None, out, None, accumulate, f16x2=True, x_amax=dy_amax if dy_amax is not None else amax(dy))
        return conv3x3(dy, conv3x3_wflip(w_ohwi), None, out, None, accumulate)
    if kh == 1 and kw == 1 and stride == 1 and pad == 0:
        gemm_raw(M, Cin, Cout, dy, Cout, OP_DENSE_K, w_ohwi, Cin, OP_DENSE_R, out, Cin, accumulate=accumulate, splitk=_BWD_SPLIT)
    else:
        assert Cout % 16 == 0 and Cin % 4 == 0
        gemm_raw(M, Cin, kh * kw * Cout, dy, Cout, OP_CONV_K, w_ohwi, Cin, OP_WT_R, out, Cin, accumulate=accumulate,
                 geo=conv_geo(Ho, Wo, Cout, H, W, kh, kw, stride, pad, 1), splitk=_BWD_SPLIT)
    return out


def conv3_f16_wgrad_ok(B, H, W, Cin, Cout, kh, kw, stride, pad) -> bool:
    """will conv2d_wgrad of this convolution run the fp16 form (and want the largest magnitudes of dy and x)?"""
    return _CONV3_F16[0] and _CONV3_F16_BWD[0] and conv3w_ok(B, H, W, Cin, Cout, kh, kw, stride, pad)


def conv2d_wgrad(dy, x, dw_ohwi, stride, pad, accumulate=True, dy_amax=None, x_amax=None):
    """dw [Cout,kh,kw,Cin] (+)= sum over pixels dy^T * im2col(x)."""
    _chk_f32(dy, x, dw_ohwi)
    B, H, W, Cin = x.shape
    Cout, kh, kw, _ = dw_ohwi.shape
    _, Ho, Wo, _ = dy.shape
    assert dy.is_contiguous() and x.is_contiguous() and dw_ohwi.is_contiguous()
    Mpix, Kc = B * Ho * Wo, kh * kw * Cin
    if conv3w_ok(B, H, W, Cin, Cout, kh, kw, stride, pad):
        if not accumulate:
            dw_ohwi.zero_()
        if _CONV3_F16[0] and _CONV3_F16_BWD[0]:
            return conv3x3_wgrad(dy, x, dw_ohwi, f16x2=True, dy_amax=dy_amax, x_amax=x_amax)
        return conv3x3_wgrad(dy, x, dw_ohwi)
    sk = _pick_splitk(Cout, Kc, Mpix)
    if not accumulate and sk > 1:
        dw_ohwi.zero_()
    acc = accumulate or sk > 1
    if kh == 1 and kw == 1 and stride == 1 and pad == 0:
        gemm_raw(Cout, Kc, Mpix, dy, Cout, OP_DENSE_R, x, Cin, OP_DENSE_R, dw_ohwi, Kc, accumulate=acc, splitk=sk)
    else:
        assert Cin % 16 == 0
        gemm_raw(Cout, Kc, Mpix, dy, Cout, OP_DENSE_R, x, Cin, OP_CONV_R, dw_ohwi, Kc, accumulate=acc, splitk=sk,
                 geo=conv_geo(H, W, Cin, Ho, Wo, kh, kw, stride, pad, 0))
    return dw_ohwi


def im2col(x, kh, kw, stride, pad, Kpad):
    _chk_f32(x)
    B, H, W, Cc = x.shape
    Ho, Wo = conv_out_hw(H, W, kh, stride, pad)
    out = torch.empty((B * Ho * Wo, Kpad), device=x.device, dtype=f32)
    check(lib.vbg_im2col(P(x), B, H, W, Cc, kh, kw, stride, pad, Kpad, P(out), _stream()), "vbg_im2col")
    return out


# ----------------------------------------------------------------------------------------------
# row kernels
# ----------------------------------------------------------------------------------------------
def embed_ln_fwd(ids, pos_ids, word, pos, type0, gamma, beta, eps, p, seed, sid):
    ntok, hidden = ids.numel(), word.shape[1]
    out = torch.empty((ntok, hidden), device=word.device, dtype=f32)
    xhat = torch.empty_like(out)
    rstd = torch.empty((ntok,), device=word.device, dtype=f32)
    check(lib.vbg_embed_ln_fwd(P(ids), P(pos_ids), ntok, hidden, P(word), P(pos), P(type0), P(gamma), P(beta), eps, p, seed, sid,
                               P(out), P(xhat), P(rstd), _stream()), "vbg_embed_ln_fwd")
    return out, xhat, rstd


def embed_ln_bwd(dout, xhat, rstd, ids, pos_ids, gamma, p, seed, sid, dword, dpos, dtype0, dgamma, dbeta):
    ntok, hidden = xhat.shape
    check(lib.vbg_embed_ln_bwd(P(dout), P(xhat), P(rstd), P(ids), P(pos_ids), ntok, hidden, P(gamma), p, seed, sid, P(dword),
                               P(dpos), P(dtype0), P(dgamma), P(dbeta), _stream()), "vbg_embed_ln_bwd")


def dropout_add_ln_fwd(x, res, gamma, beta, eps, p, seed, sid, out_planes=None, out_pair=None):
    """out_planes: Planes [rows, hidden] that receive the split of y in the same pass (ld == hidden: no padding columns); out_pair:
    fp16-pair Planes of y as well (with out_planes)"""
    rows, hidden = x.shape
    y = torch.empty_like(x)
    xhat = torch.empty_like(x)
    rstd = torch.empty((rows,), device=x.device, dtype=f32)
    if out_planes is not None or out_pair is not None:
        assert (out_planes is None or (out_planes.ld == hidden and out_planes.rows == rows)) and (out_pair is None or out_pair.ld == hidden)
        check(lib.vbg_dropout_add_ln_fwd_planes(P(x), P(res), rows, hidden, P(gamma), P(beta), eps, p, seed, sid, P(y), P(xhat), P(rstd),
                                                P(out_planes.buf) if out_planes is not None else None, out_planes.ld if out_planes is not None else 0,
                                                out_planes.plane if out_planes is not None else 0,
                                                P(out_pair.buf) if out_pair is not None else None, out_pair.ld if out_pair is not None else 0,
                                                out_pair.plane if out_pair is not None else 0, _stream()), "vbg_dropout_add_ln_fwd_planes")
        return y, xhat, rstd
    check(lib.vbg_dropout_add_ln_fwd(P(x), P(res), rows, hidden, P(gamma), P(beta), eps, p, seed, sid, P(y), P(xhat), P(rstd),
                                     _stream()), "vbg_dropout_add_ln_fwd")
    return y, xhat, rstd


_LN_WS = {}


def _ln_workspace(device, hidden, rows, na=2):
    """partials workspace of the LayerNorm backward ([blocks of the kernel][na][hidden] fp32, uninitialised: every block stores its row,
    the fold launch reads them all); one per device and stream, grown on demand"""
    key = (device, hidden, na, raw_stream(device))
    need = int(lib.vbg_ln_bwd_ws_rows(rows)) * na * hidden
    ws = _LN_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = _LN_WS[key] = torch.empty((need,), device=device, dtype=f32)
    return ws


def dropout_add_ln_bwd_planes(dy, xhat, rstd, gamma, p, seed, sid, dgamma, dbeta, dbias):
    """LayerNorm backward with dx as planes (-> Planes, dres) and dbias += column sums of dx in the same pass"""
    rows, hidden = xhat.shape
    dev = xhat.device
    ws = _ln_workspace(dev, hidden, rows, 3)
    pdx = planes_empty(rows, hidden, dev)
    assert pdx.ld == hidden
    dres = torch.empty_like(xhat)
    check(lib.vbg_dropout_add_ln_bwd_planes(P(dy), P(xhat), P(rstd), rows, hidden, P(gamma), p, seed, sid, P(pdx.buf), pdx.ld, pdx.plane, P(dres),
                                            P(dgamma), P(dbeta), P(dbias), P(ws), _stream()), "vbg_dropout_add_ln_bwd_planes")
    return pdx, dres


def dropout_add_ln_bwd_pair(dy, xhat, rstd, gamma, p, seed, sid, dgamma, dbeta, dbias, dy_amax, dx_amax, dx_bound):
    """LayerNorm backward with dx as fp16-pair planes scaled by a BOUND (-> Planes, dres): dy_amax = amax slot of dy (in), dx_amax = zeroed
    slot for max |dx| (out), dx_bound = zeroed slot that receives the bound (out: pass it as a_amax of the products that read the planes);
    dbias += column sums of dx"""
    rows, hidden = xhat.shape
    dev = xhat.device
    ws = _ln_workspace(dev, hidden, rows, 3)
    qdx = pair_empty(rows, hidden, dev)
    assert qdx.ld == hidden
    dres = torch.empty_like(xhat)
    check(lib.vbg_dropout_add_ln_bwd_pair(P(dy), P(xhat), P(rstd), rows, hidden, P(gamma), p, seed, sid, P(qdx.buf), qdx.ld, qdx.plane, P(dres),
                                          P(dgamma), P(dbeta), P(dbias), P(ws), P(dy_amax), P(dx_amax), P(dx_bound), _stream()),
          "vbg_dropout_add_ln_bwd_pair")
    return qdx, dres


def dropout_add_ln_bwd(dy, xhat, rstd, gamma, p, seed, sid, dgamma, dbeta, dx_amax=None):
    """dx_amax: zeroed amax slot that receives max |dx| (the scale of dx as an fp16-pair operand)"""
    rows, hidden = xhat.shape
    dx = torch.empty_like(xhat)
    dres = torch.empty_like(xhat)
    ws = _ln_workspace(xhat.device, hidden, rows) if rows >= 512 else None
    check(lib.vbg_dropout_add_ln_bwd(P(dy), P(xhat), P(rstd), rows, hidden, P(gamma), p, seed, sid, P(dx), P(dres), P(dgamma),
                                     P(dbeta), P(ws), P(dx_amax), _stream()), "vbg_dropout_add_ln_bwd")
    return dx, dres


def softmax_fwd(s, off, lens, ldp, ngroups, heads, maxlen, scale, p, seed, sid):
    check(lib.vbg_softmax_fwd(P(s), P(off), P(lens), P(ldp), ngroups, heads, maxlen, scale, p, seed, sid, _stream()), "vbg_softmax_fwd")


def softmax_bwd(pbuf, dp, off, lens, ldp, ngroups, heads, maxlen, scale, p):
    check(lib.vbg_softmax_bwd(P(pbuf), P(dp), P(off), P(lens), P(ldp), ngroups, heads, maxlen, scale, p, _stream()), "vbg_softmax_bwd")


# ----------------------------------------------------------------------------------------------
# fused attention (csrc/attn.hip): no [L, L] block is written; meta = vbg.functions.AttnMeta with the flash tables
# ----------------------------------------------------------------------------------------------
def attn_keep_scale(p):
    """1 / (1 - p') with p' = the 16-bit quantised drop rate the mask kernel realises (thr16 / 65536)"""
    thr = int(lib.vbg_attn_drop_thr16(float(p)))
    return 65536.0 / (65536.0 - thr)


def attn_mask(meta, p, seed, sid):
    """dropout keeps of one layer and step in both orientations -> (mask_q, mask_k) uint32 words"""
    dev = meta.lens.device
    mq = torch.empty((meta.mask_words,), device=dev, dtype=i32)
    mk = torch.empty((meta.mask_words,), device=dev, dtype=i32)
    check(lib.vbg_attn_mask(P(meta.lens), P(meta.mask_off), meta.nseq, meta.heads, meta.maxlen, float(p), seed, sid, P(mq), P(mk), _stream()),
          "vbg_attn_mask")
    return mq, mk


_MASK_POOL = [os.environ.get("VBG_MASK_POOL", "1") != "0"]


def mask_pool_enabled() -> bool:
    """attention-dropout keeps of all encoder layers of a step from ONE launch (on) or one launch per layer (off)"""
    return _MASK_POOL[0]


def attn_mask_layers(meta, p, seed, sid0, sid_stride, nlayers):
    """the dropout keeps of `nlayers` encoder layers of one step in ONE launch -> [(mask_q, mask_k)] per layer (views of two buffers);
    layer l draws from stream id sid0 + l * sid_stride: the same bits as attn_mask(meta, p, seed, sid0 + l * sid_stride)"""
    dev = meta.lens.device
    words = (meta.mask_words + 3) // 4 * 4
    mq = torch.empty((nlayers, words), device=dev, dtype=i32)
    mk = torch.empty((nlayers, words), device=dev, dtype=i32)
    check(lib.vbg_attn_mask_layers(P(meta.lens), P(meta.mask_off), meta.nseq, meta.heads, meta.maxlen, float(p), seed, sid0, sid_stride, nlayers,
                                   words, P(mq), P(mk), _stream()), "vbg_attn_mask_layers")
    return [(mq[l, :meta.mask_words], mk[l, :meta.mask_words]) for l in range(nlayers)]


_ATTN_PAIR = [os.environ.get("VBG_ATTN_PAIR", "1") != "0"]


def set_attn_pair(on: bool):
    """fused attention on two fp16 pieces per operand / three piece products (csrc/attn.hip FORM 1; inside an autocast region the hi pieces
    alone, FORM 2) wherever the encoder layer runs its fp16-pair path; off: three bf16 pieces / six piece products everywhere"""
    _ATTN_PAIR[0] = bool(on)


def attn_pair_enabled() -> bool:
    return _ATTN_PAIR[0]


def attn(meta, mode, qkv: Planes, dO, out, lse, delta, masks, scale, p, kbar=None, out_planes=None, o=None, out_amax=None, out_pair=None,
         do_amax=None):
    """one fused attention pass (mode: lib.ATTN_FWD / ATTN_DQ / ATTN_DKV) over all (sequence, head) pairs of the packed batch.  The
    arithmetic form follows the operands: three bf16 planes -> six piece products; fp16-pair planes -> three (inside an autocast region:
    their hi planes alone, one product).  do_amax: the amax slot dO's pair planes were scaled with."""
    d = AttnDesc()
    pairs = qkv.buf.shape[0] == 2
    assert dO is None or (dO.buf.shape[0] == 2) == pairs, "q / k / v and dO planes of different forms"
    d.form = (2 if amp_one_product() else 1) if pairs else 0
    if do_amax is not None:
        d.do_amax = do_amax.data_ptr()
    _seen(("attn:onep" if d.form == 2 else "attn:pair") if pairs else "attn:bf16x3")
    d.mode, d.heads, d.ntasks, d.max_len = int(mode), meta.heads, meta.ntasks, meta.maxlen
    d.tasks, d.seq_len, d.seq_row0, d.pad_off, d.ntok_pad = P(meta.tasks), P(meta.lens), P(meta.seq_row0), P(meta.pad_off), meta.ntok_pad
    d.qkv, d.qkv_plane, d.qkv_ld = qkv.buf.data_ptr(), qkv.plane, qkv.ld
    if dO is not None:
        d.dO, d.do_plane, d.do_ld = dO.buf.data_ptr(), dO.plane, dO.ld
    d.out, d.ldo = out.data_ptr(), out.stride(0)
    d.lse = lse.data_ptr()
    d.delta = None if delta is None else delta.data_ptr()
    if kbar is not None:
        d.kbar, d.ldk = kbar.data_ptr(), kbar.stride(0)
    if o is not None:                     # DQ: the forward's O (same row stride as kbar)
        assert kbar is not None and o.stride(0) == kbar.stride(0)
        d.o = o.data_ptr()
    if out_planes is not None:            # FWD: the planes of O ride along (the A operand of the output projection)
        d.out_planes, d.op_plane, d.op_ld = out_planes.buf.data_ptr(), out_planes.plane, out_planes.ld
    if out_pair is not None:              # FWD: ... and as fp16-pair planes (saved for the output projection's weight gradient)
        d.out_pair, d.oq_plane, d.oq_ld = out_pair.buf.data_ptr(), out_pair.plane, out_pair.ld
    if masks is not None:
        d.mask_q, d.mask_k, d.mask_off = masks[0].data_ptr(), masks[1].data_ptr(), meta.mask_off.data_ptr()
        d.keep_scale = attn_keep_scale(p)
    else:
        d.keep_scale = 1.0
    d.scale = float(scale)
    if out_amax is not None:              # DQ / DKV: amax slot that receives max |d(qkv)|
        d.out_amax = out_amax.data_ptr()
    check(lib.vbg_attn(C.byref(d), _stream()), "vbg_attn")
    return out


# ----------------------------------------------------------------------------------------------
# One encoder layer forward as ONE library call (csrc/encoder.hip, include/vbg.h vbg_bert_layer_fwd): the same seven launches with the
# same descriptors as plane_gemm / attn / dropout_add_ln_fwd above would build one by one -- bit-identical results -- for about a third of
# the host time per layer (no per-launch descriptor building in Python, one ctypes call).  `VBG_LAYER_ENTRY=0`: the per-launch path.
# ----------------------------------------------------------------------------------------------
_LAYER_ENTRY = [os.environ.get("VBG_LAYER_ENTRY", "1") != "0"]


def set_layer_entry(on: bool):
    _LAYER_ENTRY[0] = bool(on)


def layer_entry_ok() -> bool:
    """the measurement hooks time single launches and the stream-K tail needs its workspace: both stay on the per-launch path"""
    return _LAYER_ENTRY[0] and _GEMM_PROF is None and not _STREAMK[0]


def stacked_qkv(wq, wk, wv, bq, bk, bv, pair):
    """planes of [wq; wk; wv] and the stacked bias for parameters that are NOT stored back to back (no flat buffers: inference, frozen
    encoders) -- the Q/K/V projection then runs as one product as it does in training.  Cached on wq; stale when the optimizer kernels
    ran (epoch), torch updated one of the six tensors in place (`_version`) or one moved (data_ptr).  No-grad callers only."""
    cache = wq.__dict__.setdefault("_vbg_wplanes", {})
    key = ("qkv", bool(pair))
    tag = (_W_EPOCH[0],) + tuple(v for t in (wq, wk, wv, bq, bk, bv) for v in (t._version, t.data_ptr()))
    hit = cache.get(key)
    if hit is not None and hit[0] == tag:
        return hit[1], hit[2]
    n, k = wq.shape
    with torch.no_grad():
        pl = hit[1] if (hit is not None and hit[1].buf.device == wq.device) else (pair_empty(3 * n, k, wq.device) if pair else planes_empty(3 * n, k, wq.device))
        for j, w in enumerate((wq, wk, wv)):
            assert tuple(w.shape) == (n, k) and w.is_contiguous()
            sub = Planes(pl.buf[:, j * n:(j + 1) * n], n, k, pl.ld)
            if pair:
                split_planes_pair(w.detach(), out=sub)
            else:
                split_planes(w.detach(), out=sub)
        bias = torch.cat([bq.detach().reshape(-1), bk.detach().reshape(-1), bv.detach().reshape(-1)]).contiguous()
    cache[key] = (tag, pl, bias)
    return pl, bias


def _pref(r, pl):
    if pl is not None:
        r.buf, r.plane, r.ld = pl.buf.data_ptr(), pl.buf.stride(0), pl.ld


def bert_layer_fwd(meta, *, eps, p, seed, sid, x, xa, pair_qkv, wqkv, bqkv, tile_qkv, pqkv, attn_pair, ctxv, lse, kbar, pctx, pctxq, masks, scale,
                   wo, bo, ao_pair, tile_ao, ao, g1, b1, x1, xh1, rs1, px1, px1q, wi, bi, wo2, bo2, pair_ffn, tile_ffn1, tile_ffn2, h, pg, pgq, fo,
                   g2, b2, y, xh2, rs2, py, pyq):
    """Planes arguments may be None where include/vbg.h marks them optional; the flags say which form each product runs (pair: fp16-pair
    planes, three piece products -- one on the hi planes inside an autocast region -- else three bf16 planes, six)."""
    d = BertLayerFwdDesc()
    ntok, hid = x.shape
    inter = h.shape[1]
    gform = 2 if _AMP[0] else 1
    d.ntok, d.hidden, d.inter, d.heads = ntok, hid, inter, meta.heads
    d.eps, d.drop_p, d.seed, d.stream_id0 = eps, p, seed, sid
    d.form_qkv = gform if pair_qkv else 0
    d.form_attn = (2 if amp_one_product() else 1) if attn_pair else 0
    d.form_ao = gform if ao_pair else 0
    d.form_ffn = gform if pair_ffn else 0
    d.tile_qkv, d.tile_ao, d.tile_ffn1, d.tile_ffn2 = tile_qkv, tile_ao, tile_ffn1, tile_ffn2
    d.ntasks, d.max_len, d.ntok_pad = meta.ntasks, meta.maxlen, meta.ntok_pad
    d.tasks, d.seq_len, d.seq_row0, d.pad_off = meta.tasks.data_ptr(), meta.lens.data_ptr(), meta.seq_row0.data_ptr(), meta.pad_off.data_ptr()
    if masks is not None:
        d.mask_q, d.mask_k, d.mask_off = masks[0].data_ptr(), masks[1].data_ptr(), meta.mask_off.data_ptr()
        d.keep_scale = attn_keep_scale(p)
    else:
        d.keep_scale = 1.0
    d.attn_scale = scale
    d.x = x.data_ptr()
    _pref(d.xa, xa); _pref(d.wqkv, wqkv); _pref(d.wo, wo); _pref(d.wi, wi); _pref(d.wo2, wo2)
    d.bqkv, d.bo, d.bi, d.bo2 = bqkv.data_ptr(), bo.data_ptr(), bi.data_ptr(), bo2.data_ptr()
    d.g1, d.b1, d.g2, d.b2 = g1.data_ptr(), b1.data_ptr(), g2.data_ptr(), b2.data_ptr()
    _pref(d.pqkv, pqkv)
    d.ctx, d.lse = ctxv.data_ptr(), lse.data_ptr()
    if kbar is not None:
        d.kbar = kbar.data_ptr()
    _pref(d.pctx, pctx); _pref(d.pctxq, pctxq)
    d.ao, d.x1, d.xhat1, d.rstd1 = ao.data_ptr(), x1.data_ptr(), xh1.data_ptr(), rs1.data_ptr()
    _pref(d.px1, px1); _pref(d.px1q, px1q)
    d.h = h.data_ptr()
    _pref(d.pg, pg); _pref(d.pgq, pgq)
    d.fo, d.y, d.xhat2, d.rstd2 = fo.data_ptr(), y.data_ptr(), xh2.data_ptr(), rs2.data_ptr()
    _pref(d.py, py); _pref(d.pyq, pyq)
    if _DISPATCH[0] is not None:            # (the same tags the per-launch path leaves: tests assert which forms ran)
        for f, t in ((d.form_qkv, tile_qkv), (d.form_ao, tile_ao), (d.form_ffn, tile_ffn1), (d.form_ffn, tile_ffn2)):
            _seen(("plane_gemm:onep" if f == 2 else "plane_gemm:pair") if f else "plane_gemm:bf16x3")
            _seen(f"plane_gemm:tile{int(t)}")
        _seen(("attn:onep" if d.form_attn == 2 else "attn:pair") if attn_pair else "attn:bf16x3")
        _seen("bert_layer_fwd:entry")
    check(lib.vbg_bert_layer_fwd(C.byref(d), _stream()), "vbg_bert_layer_fwd")


def row_softmax(x):
    rows, cols = x.shape
    y = torch.empty_like(x)
    check(lib.vbg_row_softmax(P(x.contiguous()), rows, cols, P(y), _stream()), "vbg_row_softmax")
    return y


def gelu_bwd_(h, dg):
    _untag(dg)
    check(lib.vbg_gelu_bwd(P(h), P(dg), dg.numel(), _stream()), "vbg_gelu_bwd")
    return dg


def relu_bwd_(y, dy):
    _untag(dy)
    check(lib.vbg_relu_bwd(P(y), P(dy), dy.numel(), _stream()), "vbg_relu_bwd")
    return dy


def add_(a, b):
    assert a.numel() == b.numel() and a.is_contiguous() and b.is_contiguous()
    _untag(a)
    check(lib.vbg_add_inplace(P(a), P(b), a.numel(), _stream()), "vbg_add_inplace")
    return a


def scale_(a, s):
    check(lib.vbg_scale_inplace(P(a), a.numel(), float(s), _stream()), "vbg_scale_inplace")
    return a


# ----------------------------------------------------------------------------------------------
# BERTgrid
# ----------------------------------------------------------------------------------------------
def seg_reduce_fwd(tok, tok_row, run_start, run_len, mode):
    nseg, hidden = run_start.numel(), tok.shape[1]
    out = torch.empty((nseg, hidden), device=tok.device, dtype=f32)
    check(lib.vbg_seg_reduce_fwd(P(tok), P(tok_row), P(run_start), P(run_len), nseg, hidden, mode, P(out), _stream()), "vbg_seg_reduce_fwd")
    return out


def seg_reduce_bwd(dout, tok_row, run_start, run_len, mode, dtok):
    nseg, hidden = dout.shape
    check(lib.vbg_seg_reduce_bwd(P(dout), P(tok_row), P(run_start), P(run_len), nseg, hidden, mode, P(dtok), _stream()), "vbg_seg_reduce_bwd")


def owner_map(boxes, box_off, B, gh, gw, stride):
    own = torch.empty((B, gh, gw), device=box_off.device, dtype=i32)
    check(lib.vbg_owner_map(P(boxes) if boxes.numel() else None, P(box_off), B, gh, gw, stride, P(own), _stream()), "vbg_owner_map")
    return own


def grid_scatter_fwd(emb, owner, C_, layout=0):
    B, gh, gw = owner.shape
    shape = (B, gh, gw, C_) if layout == 0 else (B, C_, gh, gw)
    grid = torch.empty(shape, device=owner.device, dtype=f32)
    check(lib.vbg_grid_scatter_fwd(P(emb) if emb.numel() else None, P(owner), B, gh, gw, C_, layout, P(grid), _stream()), "vbg_grid_scatter_fwd")
    return grid


def grid_scatter_bwd(dgrid, owner, boxes, box_doc, stride, demb):
    B, gh, gw = owner.shape
    nbox, C_ = demb.shape
    check(lib.vbg_grid_scatter_bwd(P(dgrid), P(owner), P(boxes) if nbox else None, P(box_doc) if nbox else None, nbox, gh, gw, stride,
                                   C_, P(demb), _stream()), "vbg_grid_scatter_bwd")


def label_raster(owner, seg_class):
    pn = torch.empty_like(owner)
    cl = torch.empty_like(owner)
    check(lib.vbg_label_raster(P(owner), P(seg_class) if seg_class.numel() else None, owner.numel(), P(pn), P(cl), _stream()), "vbg_label_raster")
    return pn, cl


# ----------------------------------------------------------------------------------------------
# conv helpers
# ----------------------------------------------------------------------------------------------
_BN_SLOTS = [0]


def bn_slots():
    if not _BN_SLOTS[0]:
        _BN_SLOTS[0] = int(lib.vbg_bn_slots())
    return _BN_SLOTS[0]


_BN_WS = {}


def _bn_workspace(device, C_):
    """persistent zero slot rows [slots*2C] fp64 (one per device, stream and width): every reduction that writes into it is folded
    -- and the slots cleared again -- by the very next launch (bn_finalize / bn_fold / bn_param_grad), so no per-layer fills"""
    key = (device, C_, raw_stream(device))
    ws = _BN_WS.get(key)
    if ws is None:
        ws = _BN_WS[key] = torch.zeros((bn_slots() * 2 * C_,), device=device, dtype=torch.float64)
    return ws


_BN_FOLD = [os.environ.get("VBG_BN_FOLD", "1") != "0"]
_BN_ZPOOL = {}


def set_bn_fold(on: bool):
    """BatchNorm finalize / affine-gradient folds inside the apply kernels' prologues (vbg_bn_apply_fold / vbg_bn_bwd_apply_fold: one
    launch instead of two per layer and direction) or as launches of their own (off)"""
    _BN_FOLD[0] = bool(on)


def bn_fold_ok(C_, sync=False) -> bool:
    return _BN_FOLD[0] and not sync and C_ % 64 == 0


def bn_zero_slots(device, C_):
    """ZEROED slot rows [slots * 2C] fp64 for one reduction whose consumer does not clear them (the folding apply kernels: many blocks read
    the rows): slices of a zero-filled pool, one 8 MB fill per ~70 layers; an exhausted pool is replaced, never rewound"""
    key = (device, raw_stream(device))
    n = bn_slots() * 2 * C_
    ent = _BN_ZPOOL.get(key)
    if ent is None or ent[1] + n > ent[0].numel():
        ent = _BN_ZPOOL[key] = [torch.zeros((max(1 << 20, n),), device=device, dtype=torch.float64), 0]
    o = ent[1]
    ent[1] = o + n
    return ent[0][o:o + n]


def bn_fold_count(slots, C_, count):
    """[2C + 1] fp64 = the folded slot rows (cleared behind the read) and this rank's row count: the SyncBatchNorm all-reduce buffer"""
    out = torch.empty((2 * C_ + 1,), device=slots.device, dtype=torch.float64)
    check(lib.vbg_bn_fold_count(P(slots), bn_slots(), 1, C_, P(out), float(count), _stream()), "vbg_bn_fold_count")
    return out


def bn_apply_fold(x2d, res2d, slots, count, eps, momentum, running_mean, running_var, gamma, beta, relu, y_amax=None, nslots=None, count_dev=None):
    """finalize + apply in one launch -> (y, mean, invstd); slots: zeroed rows the statistics were accumulated into (bn_zero_slots), or
    (nslots = 1, count_dev) the all-reduced sums of a SyncBatchNorm and their row count on the device"""
    M, C_ = x2d.shape
    out = torch.empty_like(x2d)
    mean = torch.empty((C_,), device=x2d.device, dtype=f32)
    invstd = torch.empty_like(mean)
    check(lib.vbg_bn_apply_fold(P(x2d), P(res2d), M, C_, P(slots), bn_slots() if nslots is None else int(nslots), float(count), P(count_dev), eps, momentum,
                                P(mean), P(invstd), P(running_mean), P(running_var), P(gamma), P(beta), int(relu), P(out), P(y_amax), _stream()),
          "vbg_bn_apply_fold")
    if running_var is not None:
        _BN_EPOCH[0] += 1
    return out, mean, invstd


def bn_bwd_apply_fold(dy, y, x, mean, invstd, gamma, slots, count, relu, want_dres, dgamma, dbeta, dx_amax=None):
    """affine-gradient fold + backward apply in one launch -> (dx, dres); dgamma / dbeta are added into"""
    M, C_ = x.shape
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    check(lib.vbg_bn_bwd_apply_fold(P(dy), P(y), P(x), M, C_, P(mean), P(invstd), P(gamma), P(slots), bn_slots(), float(count), int(relu), P(dx),
                                    P(dres), P(dgamma), P(dbeta), P(dx_amax), _stream()), "vbg_bn_bwd_apply_fold")
    return dx, dres


def bn_stats(x2d, stats=None):
    """per-channel (sum, sum of squares) partials in the persistent slot workspace (MUST be consumed by bn_finalize / bn_fold next), or in
    the zeroed rows `stats`"""
    M, C_ = x2d.shape
    if stats is None:
        stats = _bn_workspace(x2d.device, C_)
    check(lib.vbg_bn_stats(P(x2d), M, C_, P(stats), _stream()), "vbg_bn_stats")
    return stats


def bn_fold(slots, C_, out=None):
    """sum the slot rows -> [2C] fp64 (and clear them)"""
    if out is None:
        out = torch.empty((2 * C_,), device=slots.device, dtype=torch.float64)
    check(lib.vbg_bn_param_grad(P(slots), bn_slots(), 1, C_, P(out), None, None, _stream()), "vbg_bn_param_grad")
    return out


def bn_finalize(stats, C_, nslots, count, eps, momentum, running_mean, running_var, count_dev=None):
    """nslots > 1: `stats` is the slot workspace (cleared behind the read); nslots == 1: already folded sums"""
    mean = torch.empty((C_,), device=stats.device, dtype=f32)
    invstd = torch.empty_like(mean)
    check(lib.vbg_bn_finalize(P(stats), int(nslots), int(nslots > 1), float(count), P(count_dev), C_, eps, momentum, P(mean), P(invstd),
                              P(running_mean), P(running_var), _stream()), "vbg_bn_finalize")
    if running_var is not None:
        _BN_EPOCH[0] += 1              # the kernel wrote the running statistics behind torch's version counters
    return mean, invstd


_BN_EPOCH = [0]


def bn_epoch() -> int:
    """bumped whenever a kernel of this library updated BatchNorm running statistics in place (caches derived from them are stale)"""
    return _BN_EPOCH[0]


def bn_apply(x2d, res2d, mean, invstd, gamma, beta, relu, out=None, y_amax=None):
    """y_amax: zeroed int32 device word that receives the bit pattern of max |y|"""
    M, C_ = x2d.shape
    if out is None:
        out = torch.empty_like(x2d)
    check(lib.vbg_bn_apply(P(x2d), P(res2d), M, C_, P(mean), P(invstd), P(gamma), P(beta), int(relu), P(out), P(y_amax), _stream()), "vbg_bn_apply")
    return out


def bn_bwd_reduce(dy, y, x, mean, invstd, relu, sums=None):
    """(sum g, sum g*xhat) partials in the persistent slot workspace (MUST be consumed by bn_param_grad next) or in the zeroed rows `sums`"""
    M, C_ = x.shape
    if sums is None:
        sums = _bn_workspace(x.device, C_)
    check(lib.vbg_bn_bwd_reduce(P(dy), P(y), P(x), M, C_, P(mean), P(invstd), int(relu), P(sums), _stream()), "vbg_bn_bwd_reduce")
    return sums


def bn_bwd_apply(dy, y, x, mean, invstd, gamma, sums, count, relu, want_dres, dgamma, dbeta, count_dev=None, dx_amax=None):
    """dx_amax: zeroed int32 device word that receives the bit pattern of max |dx| (the scale of the fp16-form products that consume dx)"""
    M, C_ = x.shape
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    check(lib.vbg_bn_bwd_apply(P(dy), P(y), P(x), M, C_, P(mean), P(invstd), P(gamma), P(sums), float(count), P(count_dev), int(relu), P(dx), P(dres),
                               P(dgamma), P(dbeta), P(dx_amax), _stream()), "vbg_bn_bwd_apply")
    return dx, dres


def bn_param_grad(slots, C_, dgamma, dbeta):
    """fold the slot rows (-> [2C] fp64, returned; slots cleared) and accumulate the affine gradients from them"""
    folded = torch.empty((2 * C_,), device=slots.device, dtype=torch.float64)
    check(lib.vbg_bn_param_grad(P(slots), bn_slots(), 1, C_, P(folded), P(dgamma), P(dbeta), _stream()), "vbg_bn_param_grad")
    return folded


def maxpool_fwd(x):
    B, H, W, C_ = x.shape
    Ho, Wo = conv_out_hw(H, W, 3, 2, 1)
    y = torch.empty((B, Ho, Wo, C_), device=x.device, dtype=f32)
    am = torch.empty((B, Ho, Wo, C_), device=x.device, dtype=i32)
    check(lib.vbg_maxpool3x3s2_fwd(P(x), B, H, W, C_, P(y), P(am), _stream()), "vbg_maxpool_fwd")
    return y, am


def maxpool_bwd(dy, am, H, W):
    B, Ho, Wo, C_ = dy.shape
    dx = torch.empty((B, H, W, C_), device=dy.device, dtype=f32)
    check(lib.vbg_maxpool3x3s2_bwd(P(dy), P(am), B, Ho, Wo, C_, H, W, P(dx), _stream()), "vbg_maxpool_bwd")
    return dx


def avgpool2_fwd(x):
    B, H, W, C_ = x.shape
    y = torch.empty((B, H // 2, W // 2, C_), device=x.device, dtype=f32)
    check(lib.vbg_avgpool2_fwd(P(x), B, H, W, C_, P(y), _stream()), "vbg_avgpool2_fwd")
    return y


def avgpool2_bwd(dy, H, W):
    B, _, _, C_ = dy.shape
    dx = torch.empty((B, H, W, C_), device=dy.device, dtype=f32)
    check(lib.vbg_avgpool2_bwd(P(dy), B, H, W, C_, P(dx), _stream()), "vbg_avgpool2_bwd")
    return dx


def upsample2_add(lo, skip):
    B, H, W, C_ = skip.shape
    y = torch.empty_like(skip)
    check(lib.vbg_upsample2_add(P(lo), P(skip), B, H, W, C_, P(y), _stream()), "vbg_upsample2_add")
    return y


def sumpool(hi, f, out=None, accumulate=False):
    B, H, W, C_ = hi.shape
    if out is None:
        out = torch.empty((B, H // f, W // f, C_), device=hi.device, dtype=f32)
        accumulate = False
    elif accumulate:
        _untag(out)
    check(lib.vbg_sumpool(P(hi), B, H, W, C_, f, P(out), int(accumulate), _stream()), "vbg_sumpool")
    return out


def nchw_to_nhwc(x):
    B, C_, H, W = x.shape
    y = torch.empty((B, H, W, C_), device=x.device, dtype=f32)
    check(lib.vbg_nchw_to_nhwc(P(x), B, C_, H * W, P(y), _stream()), "vbg_nchw_to_nhwc")
    return y


def nhwc_to_nchw(x):
    B, H, W, C_ = x.shape
    y = torch.empty((B, C_, H, W), device=x.device, dtype=f32)
    check(lib.vbg_nhwc_to_nchw(P(x), B, C_, H * W, P(y), _stream()), "vbg_nhwc_to_nchw")
    return y


def upsample_nhwc_to_nchw(x, f):
    B, h, w, C_ = x.shape
    y = torch.empty((B, C_, h * f, w * f), device=x.device, dtype=f32)
    check(lib.vbg_upsample_nhwc_to_nchw(P(x), B, h, w, C_, f, P(y), _stream()), "vbg_upsample_nhwc_to_nchw")
    return y


def normalize_resize(img, oh, ow, mean, std, batch, b):
    _, h, w = img.shape
    _, H, W, _ = batch.shape
    m = (C.c_float * 3)(*[float(v) for v in mean])
    s = (C.c_float * 3)(*[float(v) for v in std])
    check(lib.vbg_normalize_resize(P(img), h, w, oh, ow, m, s, P(batch), b, H, W, _stream()), "vbg_normalize_resize")


def rescale_boxes(coor_i64, rh, rw):
    S = coor_i64.shape[0]
    out = torch.empty((S, 4), device=coor_i64.device, dtype=i32)
    check(lib.vbg_rescale_boxes(P(coor_i64) if S else None, S, float(rh), float(rw), P(out) if S else None, _stream()), "vbg_rescale_boxes")
    return out


# ----------------------------------------------------------------------------------------------
# RoIAlign
# ----------------------------------------------------------------------------------------------
def roi_align_fwd(feat, boxes, box_doc, out_size, scale):
    B, H, W, C_ = feat.shape
    n = boxes.shape[0]
    y = torch.empty((n, out_size, out_size, C_), device=feat.device, dtype=f32)
    check(lib.vbg_roi_align_fwd(P(feat), B, H, W, C_, P(boxes) if n else None, P(box_doc) if n else None, n, out_size, scale, P(y), _stream()), "vbg_roi_align_fwd")
    return y


def roi_align_bwd(dy, feat_shape, boxes, box_doc, out_size, scale, dfeat):
    B, H, W, C_ = feat_shape
    n = boxes.shape[0]
    check(lib.vbg_roi_align_bwd(P(dy), B, H, W, C_, P(boxes) if n else None, P(box_doc) if n else None, n, out_size, scale, P(dfeat), _stream()), "vbg_roi_align_bwd")


# ----------------------------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------------------------
def ce_fwd(logits2d, elem, labels, n, weight=None, up_shift=0, H=0, W=0):
    loss = torch.empty((n,), device=logits2d.device, dtype=f32)
    check(lib.vbg_ce_fwd(P(logits2d), logits2d.stride(0), logits2d.shape[1], P(elem), P(labels), n, P(weight), up_shift, H, W, P(loss), _stream()), "vbg_ce_fwd")
    return loss


def ce_bwd(logits2d, elem, labels, n, weight, gscale_dev, gmul, up_shift, H, W, dlogits):
    check(lib.vbg_ce_bwd(P(logits2d), logits2d.stride(0), logits2d.shape[1], P(elem), P(labels), n, P(weight), P(gscale_dev), float(gmul), up_shift,
                         H, W, P(dlogits), _stream()), "vbg_ce_bwd")


def compact(labels, value, eq):
    """-> (idx int32[n] (first count valid), count int32[1] device)"""
    n = labels.numel()
    idx = torch.empty((max(n, 1),), device=labels.device, dtype=i32)
    cnt = torch.empty((1,), device=labels.device, dtype=i32)
    wsb = lib.vbg_compact_ws_bytes(n)
    ws = torch.empty((wsb,), device=labels.device, dtype=torch.uint8)
    check(lib.vbg_compact(P(labels), n, value, int(eq), P(idx), P(cnt), P(ws), wsb, _stream()), "vbg_compact")
    return idx, cnt


def sort_desc(keys):
    n = keys.numel()
    ko = torch.empty_like(keys)
    io = torch.empty((n,), device=keys.device, dtype=i32)
    if n == 0:
        return ko, io
    wsb = lib.vbg_sort_ws_bytes(n)
    ws = torch.empty((wsb,), device=keys.device, dtype=torch.uint8)
    check(lib.vbg_sort_desc(P(keys), n, P(ko), P(io), P(ws), wsb, _stream()), "vbg_sort_desc")
    return ko, io


def gather_f32(src, idx):
    n = idx.numel()
    out = torch.empty((n,), device=src.device, dtype=f32)
    check(lib.vbg_gather_f32(P(src), P(idx), n, P(out), _stream()), "vbg_gather_f32")
    return out


def gather_i32(src, idx):
    n = idx.numel()
    out = torch.empty((n,), device=src.device, dtype=i32)
    check(lib.vbg_gather_i32(P(src), P(idx), n, P(out), _stream()), "vbg_gather_i32")
    return out


def gather_rows(src, idx):
    n, C_ = idx.numel(), src.shape[1]
    out = torch.empty((n, C_), device=src.device, dtype=f32)
    check(lib.vbg_gather_rows(P(src), P(idx), n, C_, P(out), _stream()), "vbg_gather_rows")
    return out


def scatter_rows_add(src, idx, dst):
    check(lib.vbg_scatter_rows_add(P(src), P(idx), idx.numel(), src.shape[1], P(dst), _stream()), "vbg_scatter_rows_add")
    return dst


def crf_nll_fwd(em, tags, doc_off, trans, start, stop):
    """-> (nll [ndoc], alpha [N, ntag], logz [ndoc])"""
    N, ntag = em.shape
    ndoc = doc_off.numel() - 1
    alpha = torch.empty((N, ntag), device=em.device, dtype=f32)
    logz = torch.empty((ndoc,), device=em.device, dtype=f32)
    nll = torch.empty((ndoc,), device=em.device, dtype=f32)
    check(lib.vbg_crf_nll_fwd(P(em), P(tags), P(doc_off), ndoc, P(trans), ntag, start, stop, P(alpha), P(logz), P(nll), _stream()),
          "vbg_crf_nll_fwd")
    return nll, alpha, logz


def crf_nll_bwd(em, tags, doc_off, trans, start, stop, alpha, logz, gout, dtrans):
    N, ntag = em.shape
    dem = torch.empty_like(em)
    check(lib.vbg_crf_nll_bwd(P(em), P(tags), P(doc_off), doc_off.numel() - 1, P(trans), ntag, start, stop, P(alpha), P(logz), P(gout),
                              P(dem), P(dtrans), _stream()), "vbg_crf_nll_bwd")
    return dem


def crf_viterbi(em, doc_off, trans, start, stop):
    """-> (path int32 [N], score [ndoc])"""
    N, ntag = em.shape
    ndoc = doc_off.numel() - 1
    bp = torch.empty((max(N, 1), ntag), device=em.device, dtype=i32)
    path = torch.empty((N,), device=em.device, dtype=i32)
    score = torch.empty((ndoc,), device=em.device, dtype=f32)
    check(lib.vbg_crf_viterbi(P(em), P(doc_off), ndoc, P(trans), ntag, start, stop, P(bp), P(path), P(score), _stream()), "vbg_crf_viterbi")
    return path, score


def sum_f32(x, out):
    check(lib.vbg_sum_f32(P(x), x.numel(), P(out), _stream()), "vbg_sum_f32")
    return out


def sumsq(x, out):
    check(lib.vbg_sumsq(P(x), x.numel(), P(out), _stream()), "vbg_sumsq")
    return out


# ----------------------------------------------------------------------------------------------
# optimizers
# ----------------------------------------------------------------------------------------------
def sgd_step(p, g, mom, lr, momentum, wd, first, grad_scale=1.0):
    check(lib.vbg_sgd_step(P(p), P(g), P(mom), p.numel(), lr, momentum, wd, int(first), grad_scale, _stream()), "vbg_sgd_step")


def adamw_step(p, g, m, v, lr, b1, b2, eps, wd, step, grad_scale=1.0):
    check(lib.vbg_adamw_step(P(p), P(g), P(m), P(v), p.numel(), lr, b1, b2, eps, wd, step, grad_scale, _stream()), "vbg_adamw_step")
