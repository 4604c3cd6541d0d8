"""autograd.Function wrappers: every forward AND backward below is a sequence of libvbg kernel launches
(vbg/ops.py).  torch.autograd only sequences the nodes and accumulates parameter gradients.

Layout conventions: activations NHWC fp32 contiguous; conv weights are the reference's OIHW parameters
held in channels_last memory, i.e. physically [Cout, kh, kw, Cin] (`ohwi()` is a free view)."""
import torch
import torch.distributed as dist

from . import ops
from .lib import ATTN_DKV, ATTN_DQ, ATTN_FWD, EPI_GELU_DUAL, EPI_MUL_GELU_GRAD, EPI_NONE, EPI_RELU, OP_DENSE_K, OP_DENSE_R

f32 = torch.float32


def ohwi(w):
    """OIHW parameter -> contiguous [Cout,kh,kw,Cin] tensor (a view when the parameter is channels_last)."""
    v = w.permute(0, 2, 3, 1)
    return v if v.is_contiguous() else v.contiguous()


def oihw_grad_like(dw_ohwi):
    return dw_ohwi.permute(0, 3, 1, 2)


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# ----------------------------------------------------------------------------------------------
# Gradient sink: when a parameter is marked `_vbg_sunk` (vbg/optim.FlatGroup: its `.grad` is a persistent view into a flat
# gradient buffer that is zeroed once per step), the weight-gradient GEMMs, bias column sums and normalisation-layer
# reductions accumulate STRAIGHT into it and autograd gets None for that input: no temporary gradient, no extra
# read-modify-write pass, no allocator churn.  `GRAD_READY` (set by FlatReducer) is told
# when a sunk gradient is complete so the bucket all-reduce can start, exactly like a post-accumulate hook would.
# ----------------------------------------------------------------------------------------------
GRAD_READY = [None]        # callable(param) or None
# set by ViBERTgridNet for the duration of a forward that put its encoder on the side stream (the graph then holds a JoinSideFn node, whose
# end-of-backward callback joins every side stream): backward nodes built meanwhile may use side streams of their own
SIDE_OK = [False]


def wgrad_dest(w):
    if getattr(w, "_vbg_sunk", False):
        g = w.grad
        if g is None:
            # `.grad` was set to None (optimizer.zero_grad() of torch.optim) and no StepRootFn armed this backward (a Function used on
            # its own): the flat view comes back zeroed
            g = w._vbg_flat[0].attach(w)
        return g
    return None


def sinks(w) -> bool:
    """will the BACKWARD of this step find a flat gradient view for w?  (asked in forward, where the reference's loop has not called
    optimizer.zero_grad() yet -- `.grad` may be None there and attached again, by StepRootFn, when the backward starts)"""
    return getattr(w, "_vbg_sunk", False)


def wgrad_done(w):
    if GRAD_READY[0] is not None:
        GRAD_READY[0](w)


def _bias_grad(b_param, dy2d):
    """column sums of dy: added into the sunk flat gradient (autograd gets None) or returned as a fresh tensor"""
    dst = wgrad_dest(b_param)
    if dst is not None:
        ops.colsum(dy2d, out=dst, accumulate=True)
        wgrad_done(b_param)
        return None
    return ops.colsum(dy2d)


def _affine_dest(gamma_param, beta_param):
    """(dgamma, dbeta, sunk): accumulation targets for the normalisation-layer kernels (they += into them)"""
    dg, db = wgrad_dest(gamma_param), wgrad_dest(beta_param)
    if dg is not None and db is not None:
        return dg, db, True
    return torch.zeros_like(gamma_param), torch.zeros_like(beta_param), False


def _affine_done(gamma_param, beta_param, dg, db, sunk):
    if sunk:
        wgrad_done(gamma_param)
        wgrad_done(beta_param)
        return None, None
    return dg, db


def _amax_tag(x):
    """the amax slot the producer of x published (ConvBnFn tags its output), or None.  The tag carries the tensor's version counter:
    an in-place write between producer and consumer (x.add_(...), a user hook) may have raised max |x| above what the slot says, and
    a too-small amax lets the scaled fp16 pieces of the weight-gradient operand overflow to inf -- a stale tag is ignored and the
    consumer measures the tensor itself (one vbg_amax pass).  """
    tag = getattr(x, "_vbg_amax", None)
    if tag is None:
        return None
    slot, version = tag
    return slot if x._version == version else None


class SyncCtx:
    """Process group for SyncBatchNorm statistics (train_SROIE.py:202-203 `convert_sync_batchnorm`).  `before`: optional hook run on
    the compute stream in front of every SyncBatchNorm collective (vbg/optim.FlatReducer(serialize_syncbn=True) makes the stream wait
    for the gradient buckets in flight on the staging stream, so that never two communicators have a collective in flight)."""
    group = None
    before = None
    seq = 0            # statistics collectives issued since the reducer was built; `last` = (seq, numel) -- for hang reports
    last = None
    force = False      # a process group of ONE rank runs the collectives as well (FlatReducer(force_enable=True): the one-GPU RCCL check)
    direct = None      # vbg.rccl.DirectComm: the statistics as ncclAllReduce calls on the compute stream (no torch.distributed in between)

    @classmethod
    def active(cls):
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size(cls.group) > 1 or cls.force)

    @classmethod
    def all_reduce(cls, t):
        if cls.before is not None:
            cls.before()
        cls.seq += 1
        cls.last = (cls.seq, t.numel())
        if cls.direct is not None and t.is_cuda:
            cls.direct.all_reduce_(t)
        else:
            dist.all_reduce(t, group=cls.group)


# ----------------------------------------------------------------------------------------------
def _linear_wgrad(w_param, dy, x):
    """dW = dy^T x : into the flat gradient buffer when the parameter is sunk (returns None), else a fresh tensor"""
    dst = wgrad_dest(w_param)
    if dst is not None:
        ops.linear_wgrad(dy, x, dst, accumulate=True)
        wgrad_done(w_param)
        return None
    dw = torch.empty_like(w_param)
    ops.linear_wgrad(dy, x, dw, accumulate=False)
    return dw


def _split_with_bias_grad(b_param, dy2d):
    """(planes of dy, bias gradient): the column sums ride on the split pass when the bias gradient is sunk into the flat buffer"""
    dst = wgrad_dest(b_param)
    if dst is not None:
        pl = ops.split_planes(dy2d, colsum_out=dst)
        wgrad_done(b_param)
        return pl, None
    return ops.split_planes(dy2d), ops.colsum(dy2d)


def _plane_wgrad(w_param, dy, x):
    """dW = dy^T x from transposed planes (both operands K-contiguous along the token index); into the flat gradient buffer when
    the parameter is sunk (returns None), else a fresh tensor"""
    N, K = dy.shape[1], x.shape[1]
    pa, pb = ops.split_planes_t(dy), ops.split_planes_t(x)
    dst = wgrad_dest(w_param)
    if dst is not None:
        ops.plane_gemm(pa, pb, dst, accumulate=True, tile=ops._wgrad_tile(N, K))
        wgrad_done(w_param)
        return None
    dw = torch.empty_like(w_param)
    ops.plane_gemm(pa, pb, dw, tile=ops._wgrad_tile(N, K))
    return dw


class LinearFn(torch.autograd.Function):
    """y = x @ w^T + b (optional fused ReLU)."""

    @staticmethod
    def forward(ctx, x, w, b, relu):
        x = _c(x)
        y = ops.linear_fwd(x, w, b, EPI_RELU if relu else EPI_NONE)
        ctx.relu = relu
        ctx.has_bias = b is not None
        ctx.w_ref, ctx.b_ref = w, b
        ctx.save_for_backward(x, w, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dy = _c(dy)
        if ctx.relu:
            dy = ops.relu_bwd_(y, dy.clone())
        dx = ops.linear_dgrad(dy, w) if ctx.needs_input_grad[0] else None
        dw = _linear_wgrad(ctx.w_ref, dy, x)
        db = _bias_grad(ctx.b_ref, dy) if ctx.has_bias else None
        return dx, dw, db, None


class SegLinearFn(torch.autograd.Function):
    """y = cat(x_i upsampled by 2^shift_i, channel dim) @ w^T + b with the concat never materialised
    (K segments read in place).  Covers early fusion (ResNetFPN_ViBERTgrid.py:317-318), P_fuse (:502-506)
    and late fusion (field_type_classification_head.py:185-188)."""

    @staticmethod
    def forward(ctx, w, b, shifts, hw, *xs):
        """w: [N, K] linear weight or [N, K, 1, 1] (channels_last) 1x1 conv weight"""
        ctx.w_ref, ctx.b_ref = w, b
        w2d = w.reshape(w.shape[0], -1)
        xs = [_c(x) for x in xs]
        chans = [x.shape[-1] for x in xs]
        K = sum(chans)
        assert w2d.shape[1] == K and w2d.is_contiguous()
        N = w2d.shape[0]
        full = xs[shifts.index(0)]
        M = full.numel() // full.shape[-1]
        out = torch.empty((M, N), device=full.device, dtype=f32)
        kend, segs = 0, []
        for x, c, s in zip(xs, chans, shifts):
            kend += c
            segs.append((x, kend, c, s))
        ops.gemm_raw(M, N, K, xs[0], chans[0], OP_DENSE_K, w2d, K, OP_DENSE_K, out, N, bias=b, segs=segs,
                     a_hw=hw if any(shifts) else (0, 0), f16=True)
        ctx.shifts, ctx.hw, ctx.chans, ctx.has_bias = shifts, hw, chans, b is not None
        ctx.full_shape = full.shape
        ctx.save_for_backward(w2d, *xs)
        return out.view(*full.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        w2d, *xs = ctx.saved_tensors
        N, K = w2d.shape
        dy = _c(dy)
        dy2 = dy.view(-1, N)
        dst = wgrad_dest(ctx.w_ref)
        sunk = dst is not None
        dw = dst.reshape(N, K) if sunk else torch.empty_like(w2d)
        assert dw.is_contiguous() and (not sunk or dw.data_ptr() == dst.data_ptr())
        pooled = {0: dy}
        dxs = []
        koff = 0
        for i, (x, c, s) in enumerate(zip(xs, ctx.chans, ctx.shifts)):
            if s not in pooled:
                B, H, W = ctx.full_shape[0], ctx.hw[0], ctx.hw[1]
                pooled[s] = ops.sumpool(dy.view(B, H, W, N), 1 << s)
            g2 = pooled[s].view(-1, N)
            Ms = g2.shape[0]
            dx = None
            if ctx.needs_input_grad[4 + i]:
                dx = torch.empty_like(x)
                ops.gemm_raw(Ms, c, N, g2, N, OP_DENSE_K, w2d, K, OP_DENSE_R, dx, c, b_ptr_off=koff)
            dxs.append(dx)
            sk = ops._pick_splitk(N, c, Ms)
            if sk > 1 and not sunk:
                dw[:, koff:koff + c].zero_()
            ops.gemm_raw(N, c, Ms, g2, N, OP_DENSE_R, x.view(-1, c), c, OP_DENSE_R, dw, K, c_ptr_off=koff, accumulate=sunk or sk > 1,
                         splitk=sk)
            koff += c
        db = _bias_grad(ctx.b_ref, dy2) if ctx.has_bias else None
        if sunk:
            wgrad_done(ctx.w_ref)
            dw_ret = None
        else:
            dw_ret = dw.view(ctx.w_ref.shape)
        return (dw_ret, db, None, None, *dxs)


# ----------------------------------------------------------------------------------------------
def _conv_any(x, w4, stride, pad, bias=None, want_stats=False, w_owner=None, x_amax=None, zero_slots=False):
    """x NHWC; w4 = OHWI weight.  Cin=3 stem goes through im2col (K=147, row stride 148).
    want_stats: -> third result = BatchNorm slot workspace holding the output's column sums / sums of squares (fused into the GEMM
    epilogue), or None when the product is one the library may split (the caller then runs ops.bn_stats)."""
    B, H, W, Cin = x.shape
    Cout, kh, kw, _ = w4.shape
    K = kh * kw * Cin
    Ho, Wo = ops.conv_out_hw(H, W, kh, stride, pad)
    # (zero_slots: the consumer folds the slot rows without clearing them -- rows from the zero pool instead of the persistent workspace)
    stats = ((ops.bn_zero_slots(x.device, Cout) if zero_slots else ops._bn_workspace(x.device, Cout))
             if (want_stats and ops.fuse_stats_ok(B * Ho * Wo, Cout, K)) else None)
    if Cin % 16 != 0:
        Kp = (K + 3) // 4 * 4
        col = ops.im2col(x, kh, kw, stride, pad, Kp)
        out = torch.empty((B, Ho, Wo, Cout), device=x.device, dtype=f32)
        ops.gemm_raw(B * Ho * Wo, Cout, K, col, Kp, OP_DENSE_K, w4, K, OP_DENSE_K, out, Cout, bias=bias, stats=stats)
        return out, col, stats
    return ops.conv2d_fwd(x, w4, stride, pad, bias, stats=stats, w_owner=w_owner, x_amax=x_amax), None, stats


def _conv_wgrad_any(dy, x, col, w4_shape, stride, pad, w_param=None, dy_amax=None, x_amax=None):
    """-> OIHW-shaped gradient for autograd, or None when accumulated straight into the sunk flat gradient"""
    dst = wgrad_dest(w_param) if w_param is not None else None
    if dst is not None:
        dw = ohwi(dst)
        assert dw.data_ptr() == dst.data_ptr()
        if col is not None:
            Cout, kh, kw, Cin = w4_shape
            K = kh * kw * Cin
            sk = ops._pick_splitk(Cout, K, col.shape[0])
            ops.gemm_raw(Cout, K, col.shape[0], dy, Cout, OP_DENSE_R, col, col.shape[1], OP_DENSE_R, dw, K, accumulate=True, splitk=sk)
        else:
            ops.conv2d_wgrad(dy, x, dw, stride, pad, accumulate=True, dy_amax=dy_amax, x_amax=x_amax)
        wgrad_done(w_param)
        return None
    dw = torch.empty(w4_shape, device=dy.device, dtype=f32)
    if col is not None:
        Cout, kh, kw, Cin = w4_shape
        K = kh * kw * Cin
        Mpix = col.shape[0]
        sk = ops._pick_splitk(Cout, K, Mpix)
        if sk > 1:
            dw.zero_()
        ops.gemm_raw(Cout, K, Mpix, dy, Cout, OP_DENSE_R, col, col.shape[1], OP_DENSE_R, dw, K, accumulate=sk > 1, splitk=sk)
    else:
        ops.conv2d_wgrad(dy, x, dw, stride, pad, accumulate=False, dy_amax=dy_amax, x_amax=x_amax)
    return oihw_grad_like(dw)


class ConvFn(torch.autograd.Function):
    """plain convolution (FPN lateral / merge / 1x1 heads), optional bias."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad):
        ctx.x_amax = _amax_tag(x)                         # the producer's word with the bits of max |x| (see ConvBnFn), if any
        x = _c(x)
        w4 = ohwi(w)
        y, col, _ = _conv_any(x, w4, stride, pad, b, w_owner=w, x_amax=ctx.x_amax)
        ctx.stride, ctx.pad, ctx.has_bias = stride, pad, b is not None
        ctx.side_ok = SIDE_OK[0]
        ctx.w_ref, ctx.b_ref = w, b
        ctx.save_for_backward(x, w4, col)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w4, col = ctx.saved_tensors
        dy = _c(dy)
        B_, H_, W_, Cin_ = x.shape
        Cout_, kh, kw, _ = w4.shape
        # fp16-form products of the wide 3x3 convolutions: ONE pass for the largest magnitude of dy serves both gradients
        f16_d = ctx.needs_input_grad[0] and ops.conv3_f16_bwd_ok(B_, H_, W_, Cout_, Cin_, kh, kw, ctx.stride, ctx.pad)
        f16_w = col is None and ops.conv3_f16_wgrad_ok(B_, H_, W_, Cin_, Cout_, kh, kw, ctx.stride, ctx.pad)
        dy_amax = ops.amax(dy) if (f16_d or f16_w) else None
        if ctx.side_ok and ctx.needs_input_grad[0] and sinks(ctx.w_ref) and ops.conv_wgrad_stream_enabled(2, dy.numel() // dy.shape[-1]):
            # (as in ConvBnFn: the weight gradient beside the input gradient, on the conv weight-gradient stream)
            cur, ws = torch.cuda.current_stream(dy.device), ops.side_stream(dy.device, "cwgrad")
            ws.wait_stream(cur)
            with torch.cuda.stream(ws):
                dw = _conv_wgrad_any(dy, x, col, tuple(w4.shape), ctx.stride, ctx.pad, ctx.w_ref, dy_amax=dy_amax, x_amax=ctx.x_amax)
                ops.reserve_for(ws, dy, x, col, dy_amax, ctx.x_amax)
            assert dw is None
        else:
            dw = _conv_wgrad_any(dy, x, col, tuple(w4.shape), ctx.stride, ctx.pad, ctx.w_ref, dy_amax=dy_amax, x_amax=ctx.x_amax)
        dx = ops.conv2d_dgrad(dy, w4, tuple(x.shape), ctx.stride, ctx.pad, dy_amax=dy_amax, w_owner=ctx.w_ref) if ctx.needs_input_grad[0] else None
        db = _bias_grad(ctx.b_ref, dy.view(-1, dy.shape[-1])) if ctx.has_bias else None
        return dx, dw, db, None, None


def _frozen_invstd(running_var, eps):
    """1 / sqrt(running_var + eps) of a BatchNorm that uses its frozen statistics (eval mode), computed once per version of the buffer:
    two tiny torch launches per layer and call otherwise -- 76 of the ~210 launches of a single-document inference.  The cache lives on
    the buffer object and is keyed by its version counter (load_state_dict), the library's own update counter (the training kernels
    write the running statistics behind torch's back), its address and eps."""
    tag = (running_var._version, ops.bn_epoch(), running_var.data_ptr(), float(eps))
    hit = running_var.__dict__.get("_vbg_invstd")
    if hit is not None and hit[0] == tag:
        return hit[1]
    with torch.no_grad():
        inv = torch.rsqrt(running_var + eps)
    running_var.__dict__["_vbg_invstd"] = (tag, inv)
    return inv


class ConvBnFn(torch.autograd.Function):
    """conv (bias-free) -> BatchNorm2d (batch statistics in training, SyncBN-able) -> (+ residual) -> ReLU."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, running_mean, running_var, res, stride, pad, relu, training, momentum, eps, sync):
        ctx.x_amax = _amax_tag(x)                         # the producer's word with the bits of max |x|, if any
        x = _c(x)
        sync = bool(sync) and SyncCtx.active()
        w4 = ohwi(w)
        # one rank, training statistics: finalize rides in the apply kernel's prologue, the affine-gradient fold in the backward apply's
        fold = training and ops.bn_fold_ok(w.shape[0], sync)
        z, col, stats = _conv_any(x, w4, stride, pad, want_stats=training, w_owner=w, x_amax=ctx.x_amax, zero_slots=fold)
        C = z.shape[-1]
        z2 = z.view(-1, C)
        M = z2.shape[0]
        r2 = None if res is None else _c(res).view(-1, C)
        # the largest magnitude of y rides on the kernel that writes it: the scale of y as an operand of the next convolution's
        # fp16-form weight gradient (the tag travels on the tensor object; a consumer that does not find one takes a vbg_amax pass)
        # (grad mode is off inside forward(): whether a backward will come is what ctx.needs_input_grad says)
        # ... and the scale of y as the ACTIVATION operand of the next convolution's fp16-form forward (an activation of 65520 or more no
        # longer becomes inf): published whenever that form is on, inference and no_grad included
        y_amax = ops.amax_slot(x.device) if (ops.conv3_f16_enabled() or (any(ctx.needs_input_grad) and ops.conv3_f16_bwd_enabled())) else None
        if fold:
            if stats is None:
                stats = ops.bn_stats(z2, ops.bn_zero_slots(x.device, C))
            count, count_dev = float(M), None
            y, mean, invstd = ops.bn_apply_fold(z2, r2, stats, count, eps, momentum, running_mean, running_var, gamma, beta, relu, y_amax=y_amax)
            y = y.view(z.shape)
        else:
            if training:
                if stats is None:
                    stats = ops.bn_stats(z2)
                count, count_dev = float(M), None
                if sync:
                    glob = ops.bn_fold_count(stats, C, M)        # [sum, sumsq, rows of this rank] from one launch
                    SyncCtx.all_reduce(glob)
                    count_dev = glob[2 * C:]                   # global row count stays on the device (no sync)
                    if ops.bn_fold_ok(C):                      # finalize from the all-reduced sums in the apply kernel's prologue
                        y, mean, invstd = ops.bn_apply_fold(z2, r2, glob, count, eps, momentum, running_mean, running_var, gamma, beta, relu,
                                                            y_amax=y_amax, nslots=1, count_dev=count_dev)
                        y = y.view(z.shape)
                    else:
                        mean, invstd = ops.bn_finalize(glob, C, 1, count, eps, momentum, running_mean, running_var, count_dev)
                else:
                    mean, invstd = ops.bn_finalize(stats, C, ops.bn_slots(), count, eps, momentum, running_mean, running_var)
            else:
                mean, invstd, count, count_dev = running_mean, _frozen_invstd(running_var, eps), float(M), None
            if not (training and sync and ops.bn_fold_ok(C)):
                y = ops.bn_apply(z2, r2, mean, invstd, gamma, beta, relu, y_amax=y_amax).view(z.shape)
        if y_amax is not None:
            y._vbg_amax = (y_amax, y._version)
        ctx.fold = fold
        ctx.side_ok = SIDE_OK[0]
        ctx.cfg = (stride, pad, relu, training, count, res is not None, sync)
        ctx.w_ref, ctx.affine = w, (gamma, beta)
        ctx.save_for_backward(x, w4, col, z, y if relu else None, mean, invstd, gamma, count_dev)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w4, col, z, y, mean, invstd, gamma, count_dev = ctx.saved_tensors
        stride, pad, relu, training, count, has_res, sync = ctx.cfg
        C = z.shape[-1]
        dy2 = _c(dy).view(-1, C)
        z2 = z.view(-1, C)
        y2 = None if y is None else y.view(-1, C)
        B_, H_, W_, Cin_ = x.shape
        want_amax = (ctx.needs_input_grad[0] and ops.conv3_f16_bwd_ok(B_, H_, W_, C, Cin_, w4.shape[1], w4.shape[2], stride, pad)) or \
            (col is None and ops.conv3_f16_wgrad_ok(B_, H_, W_, Cin_, C, w4.shape[1], w4.shape[2], stride, pad))
        dz_amax = ops.amax_slot(dy2.device) if want_amax else None
        dgamma, dbeta, sunk = _affine_dest(*ctx.affine)           # from the LOCAL sums: the gradient exchange averages them
        if ctx.fold and ops.bn_fold_ok(C, sync):
            # (training statistics on one rank: the fold of the slot rows and the affine gradients ride in the apply kernel's prologue)
            slots = ops.bn_bwd_reduce(dy2, y2, z2, mean, invstd, relu, ops.bn_zero_slots(dy2.device, C))
            dz2, dres2 = ops.bn_bwd_apply_fold(dy2, y2, z2, mean, invstd, gamma, slots, count, relu, has_res, dgamma, dbeta, dx_amax=dz_amax)
            dgamma, dbeta = _affine_done(*ctx.affine, dgamma, dbeta, sunk)
            return ConvBnFn._finish_backward(ctx, dz2, dres2, dz_amax, dgamma, dbeta)
        slots = ops.bn_bwd_reduce(dy2, y2, z2, mean, invstd, relu)
        sums = ops.bn_param_grad(slots, C, dgamma, dbeta)
        dgamma, dbeta = _affine_done(*ctx.affine, dgamma, dbeta, sunk)
        if not training:
            # frozen statistics (module in eval mode inside a training step): mean / var are constants, so the two batch-coupling
            # terms of the input gradient vanish -> dz = gamma * invstd * dy (same kernel, zero sums); dgamma / dbeta as above
            sums = torch.zeros_like(sums)
        elif sync:
            SyncCtx.all_reduce(sums)
        dz2, dres2 = ops.bn_bwd_apply(dy2, y2, z2, mean, invstd, gamma, sums, count, relu, has_res, None, None, count_dev, dx_amax=dz_amax)
        return ConvBnFn._finish_backward(ctx, dz2, dres2, dz_amax, dgamma, dbeta)

    @staticmethod
    def _finish_backward(ctx, dz2, dres2, dz_amax, dgamma, dbeta):
        """the convolution's input / weight gradients from dz (the gradient in front of the BatchNorm), shared by both BatchNorm routes"""
        x, w4, col, z, y, mean, invstd, gamma, count_dev = ctx.saved_tensors
        stride, pad, relu, training, count, has_res, sync = ctx.cfg
        dz = dz2.view(z.shape)
        dres = dres2.view(z.shape) if has_res else None
        if ctx.side_ok and ctx.needs_input_grad[0] and sinks(ctx.w_ref) and ops.conv_wgrad_stream_enabled(1, dz2.shape[0]):
            # the weight gradient goes straight into the flat gradient view and nothing else in this backward reads it: it runs on a
            # stream of its own beside the input gradient, which the layer below waits for -- two launches that each leave most of the
            # chip idle (the one-round launches of the late trunk stages) share it.  The operands stay reserved for that stream when this
            # node releases them; JoinSideFn's end-of-backward callback / FlatReducer's staging stream wait for it.
            cur, ws = torch.cuda.current_stream(dz.device), ops.side_stream(dz.device, "cwgrad")
            ws.wait_stream(cur)
            with torch.cuda.stream(ws):
                dw = _conv_wgrad_any(dz, x, col, tuple(w4.shape), stride, pad, ctx.w_ref, dy_amax=dz_amax, x_amax=ctx.x_amax)
                # (the amax slots too: they are views of the CALLER stream's slot pool, and a pool whose last view dies -- this node's
                # release of ctx.x_amax / dz_amax can be that moment -- goes back to the caching allocator of the caller's stream, which may
                # hand the 2 MB block to the next slot pool and zero it while this launch is still queued: the round-5 outlier)
                ops.reserve_for(ws, dz2, x, col, dz_amax, ctx.x_amax)
            assert dw is None
            dx = ops.conv2d_dgrad(dz, w4, tuple(x.shape), stride, pad, dy_amax=dz_amax, w_owner=ctx.w_ref)
            return dx, None, dgamma, dbeta, None, None, dres, None, None, None, None, None, None, None
        dx = ops.conv2d_dgrad(dz, w4, tuple(x.shape), stride, pad, dy_amax=dz_amax, w_owner=ctx.w_ref) if ctx.needs_input_grad[0] else None
        dw = _conv_wgrad_any(dz, x, col, tuple(w4.shape), stride, pad, ctx.w_ref, dy_amax=dz_amax, x_amax=ctx.x_amax)
        return dx, dw, dgamma, dbeta, None, None, dres, None, None, None, None, None, None, None


class MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        y, am = ops.maxpool_fwd(x)
        ctx.hw = (x.shape[1], x.shape[2])
        ctx.save_for_backward(am)
        return y

    @staticmethod
    def backward(ctx, dy):
        (am,) = ctx.saved_tensors
        return ops.maxpool_bwd(_c(dy), am, *ctx.hw)


class AvgPool2Fn(torch.autograd.Function):
    """nn.AvgPool2d(2, 2) on NHWC (ResNet-D projection shortcut, model/ResNetFPN_ViBERTgrid.py:222-236)."""

    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        ctx.hw = (x.shape[1], x.shape[2])
        return ops.avgpool2_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        return ops.avgpool2_bwd(_c(dy), *ctx.hw)


class UpAddFn(torch.autograd.Function):
    """nearest x2 upsample of `lo` + `skip` (FPN top-down, ResNetFPN_ViBERTgrid.py:490-500)."""

    @staticmethod
    def forward(ctx, lo, skip):
        return ops.upsample2_add(_c(lo), _c(skip))

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        return ops.sumpool(dy, 2), dy


class NhwcToNchwFlatFn(torch.autograd.Function):
    """[N,h,w,C] -> [N, C*h*w] in the reference's NCHW flatten order (nn.Flatten at
    field_type_classification_head.py:72) so `linear.weight` keeps its checkpoint layout."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = x.shape
        return ops.nhwc_to_nchw(_c(x)).view(x.shape[0], -1)

    @staticmethod
    def backward(ctx, dy):
        N, h, w, C = ctx.shape
        return ops.nchw_to_nhwc(_c(dy).view(N, C, h, w))


class RoiAlignFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, boxes, box_doc, out_size, scale):
        feat = _c(feat)
        ctx.cfg = (tuple(feat.shape), out_size, scale)
        ctx.save_for_backward(boxes, box_doc)
        return ops.roi_align_fwd(feat, boxes, box_doc, out_size, scale)

    @staticmethod
    def backward(ctx, dy):
        boxes, box_doc = ctx.saved_tensors
        shape, out_size, scale = ctx.cfg
        dfeat = torch.zeros(shape, device=dy.device, dtype=f32)
        ops.roi_align_bwd(_c(dy), shape, boxes, box_doc, out_size, scale, dfeat)
        return dfeat, None, None, None, None


class SegReduceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tok, tok_row, run_start, run_len, mode):
        tok = _c(tok)
        ctx.mode, ctx.shape = mode, tok.shape
        ctx.save_for_backward(tok_row, run_start, run_len)
        return ops.seg_reduce_fwd(tok, tok_row, run_start, run_len, mode)

    @staticmethod
    def backward(ctx, dy):
        tok_row, run_start, run_len = ctx.saved_tensors
        dtok = torch.zeros(ctx.shape, device=dy.device, dtype=f32)
        ops.seg_reduce_bwd(_c(dy), tok_row, run_start, run_len, ctx.mode, dtok)
        return dtok, None, None, None, None


class GridScatterFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, owner, boxes, box_doc, stride, layout):
        emb = _c(emb)
        ctx.cfg = (stride, layout, emb.shape)
        ctx.save_for_backward(owner, boxes, box_doc)
        return ops.grid_scatter_fwd(emb, owner, emb.shape[1], layout)

    @staticmethod
    def backward(ctx, dy):
        owner, boxes, box_doc = ctx.saved_tensors
        stride, layout, shape = ctx.cfg
        dy = _c(dy)
        if layout == 1:
            dy = ops.nchw_to_nhwc(dy)
        demb = torch.zeros(shape, device=dy.device, dtype=f32)
        ops.grid_scatter_bwd(dy, owner, boxes, box_doc, stride, demb)
        return demb, None, None, None, None, None


# ----------------------------------------------------------------------------------------------
# BERT
# ----------------------------------------------------------------------------------------------
class BertEmbedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, word, pos, typ, gamma, beta, ids, pos_ids, eps, p, seed, sid):
        type0 = typ[0].contiguous()
        out, xhat, rstd = ops.embed_ln_fwd(ids, pos_ids, word, pos, type0, gamma, beta, eps, p, seed, sid)
        ctx.cfg = (p, seed, sid, word.shape, pos.shape, typ.shape)
        ctx.w_refs = (word, pos, typ, gamma, beta)
        ctx.save_for_backward(xhat, rstd, ids, pos_ids, gamma)
        return out

    @staticmethod
    def backward(ctx, dout):
        xhat, rstd, ids, pos_ids, gamma = ctx.saved_tensors
        p, seed, sid, wshape, pshape, tshape = ctx.cfg
        dev = dout.device
        rword, rpos, rtyp, rgamma, rbeta = ctx.w_refs
        sw, sp, st = wgrad_dest(rword), wgrad_dest(rpos), wgrad_dest(rtyp)
        dword = sw if sw is not None else torch.zeros(wshape, device=dev, dtype=f32)      # 94 MB table: scatter-add in place
        dpos = sp if sp is not None else torch.zeros(pshape, device=dev, dtype=f32)
        dtyp = st if st is not None else torch.zeros(tshape, device=dev, dtype=f32)
        dg, db, sunk = _affine_dest(rgamma, rbeta)
        ops.embed_ln_bwd(_c(dout), xhat, rstd, ids, pos_ids, gamma, p, seed, sid, dword, dpos, dtyp[0], dg, db)
        if sw is not None:
            wgrad_done(rword)
            dword = None
        if sp is not None:
            wgrad_done(rpos)
            dpos = None
        if st is not None:
            wgrad_done(rtyp)
            dtyp = None
        dg, db = _affine_done(rgamma, rbeta, dg, db, sunk)
        return dword, dpos, dtyp, dg, db, None, None, None, None, None, None


def _back_to_back(a, b, c):
    """three equal-size fp32 tensors stored consecutively (vbg/optim.FlatGroup lays the Q/K/V projections out this way)"""
    n = 4 * a.numel()
    sa, sb, sc = (t.untyped_storage().data_ptr() for t in (a, b, c))
    return (sa == sb == sc and a.numel() == b.numel() == c.numel() and a.is_contiguous() and b.is_contiguous() and c.is_contiguous()
            and b.data_ptr() == a.data_ptr() + n and c.data_ptr() == b.data_ptr() + n)


def _stack3(a):
    """view of `a` and the two tensors stored right behind it as one [3*rows, ...] tensor"""
    shape = (3 * a.shape[0],) + tuple(a.shape[1:])
    return torch.as_strided(a.detach(), shape, a.stride())


class AttnMeta:
    """Host-built description of the packed variable-length sequences of one batch (see
    model/BERTgrid_generator.py in this package): device tables for the grouped attention GEMMs."""
    __slots__ = ("ntok", "nseq", "heads", "dh", "maxlen", "ld", "s_elems", "soff", "lens", "ldp", "t_qk", "t_pv", "t_dp", "t_dv",
                 "t_dq", "t_dk", "ngroups", "mask_off", "seq_row0", "pad_off", "tok_pad", "tasks", "ntok_pad", "mask_words", "ntasks", "stat_pool",
                 "mask_pool")


class BertLayerFn(torch.autograd.Function):
    """One transformer encoder layer over packed tokens [ntok, hidden] (HF BertLayer: self-attention,
    out-proj + dropout + residual LN, FFN(GELU erf) + dropout + residual LN)."""

    @staticmethod
    def forward(ctx, x, xpl, wq, bq, wk, bk, wv, bv, wo, bo, g1, b1, wi, bi, wo2, bo2, g2, b2, meta, eps, p, seed, layer):
        """xpl: the bf16 planes of x ([3, ntok, hidden] int16) when the producer already wrote them, else None.  Returns (y, planes of y
        or None): the LayerNorm that ends the layer splits its output for the next layer's first product."""
        x = _c(x)
        ntok, hid = x.shape
        H, dh = meta.heads, meta.dh
        dev = x.device
        ctx.side_ok = SIDE_OK[0]            # (this forward runs on the encoder's side stream: the graph holds the node that joins the streams)
        fused_qkv = _back_to_back(wq, wk, wv) and _back_to_back(bq, bk, bv)
        planes = ops.planes_enabled() and hid % 32 == 0 and wi.shape[0] % 32 == 0
        flash = ops.flash_ok(hid, wi.shape[0], dh, meta.maxlen)
        px = pctx = px1 = pg = pqkv = qkv = P = lse = masks = kbar = None
        if not flash:
            qkv = torch.empty((ntok, 3 * hid), device=dev, dtype=f32)
        # forward products whose operands are LayerNorm / GELU outputs and weights run on two fp16 pieces per operand (three piece
        # products, csrc/gemm_planes.hip FORM 1) once the problem fills the 8-wave tiles; the pair planes of x arrive as an attribute of
        # the previous layer's bf16 planes (written by its closing LayerNorm)
        # (round 6: nothing to differentiate -- inference, validation -- and a problem below the 8-wave tiles: the form runs on the 64 x 64 tile,
        #  half the matrix-core work of the six-product form that single documents used to take)
        nograd = not any(ctx.needs_input_grad)
        pair = planes and ops.pair_enabled() and (ops.pair_tile(ntok, hid) != 0 or (nograd and ops.pair_small_enabled()))
        ptile = lambda n, wide=False: ops.pair_tile(ntok, n, wide) or 64004          # (four LDS stages: the weights of a lone document come from HBM)
        carrier_is_pair = xpl is not None and xpl.shape[0] == 2          # (the previous layer ran the all-pair path: xpl ARE the pair planes)
        xq = (ops.Planes(xpl, ntok, hid, xpl.shape[2]) if carrier_is_pair else getattr(xpl, "_vbg_pair", None)) if (pair and xpl is not None) else None
        # all-pair path: the backward products run on two fp16 pieces as well (needs every weight / bias gradient of the layer sunk into
        # the flat buffers, the stacked Q/K/V layout and the fused attention), and so does the attention-output projection; the
        # activations saved for backward are pair planes, no bf16 planes of x1 / gelu(h) / y / ctx are written at all
        pair_bwd = (pair and ops.pair_bwd_enabled() and fused_qkv and dh == 64 and meta.maxlen <= 512 and ops.flash_enabled()
                    and any(ctx.needs_input_grad)
                    and all(sinks(t) for t in (wq, wk, wv, bq, bk, bv, wo, bo, wi, bi, wo2, bo2)))
        assert not carrier_is_pair or pair, "pair planes handed to a layer that does not run the pair form"
        if (pair_bwd or (pair and nograd)) and xq is None:
            xq = ops.split_planes_pair(x)
        # q, k, v leave the projection as planes only (the fused attention kernels' operands): fp16-pair planes when the projection runs
        # the pair form and the backward that follows is the all-pair one (round 5: the attention then runs three fp16 piece products per
        # product, csrc/attn.hip FORM 1), three bf16 planes otherwise
        # parameters NOT stored back to back and nothing to differentiate (inference, validation): the stacked planes of the three
        # projections are cached on the weights (round 6) -- one Q/K/V product as in training instead of three
        one_qkv = fused_qkv or (planes and nograd)
        attn_pair = (flash and pair and one_qkv and xq is not None and ops.attn_pair_enabled() and ops.bound_planes_enabled()
                     and (pair_bwd or not any(ctx.needs_input_grad)))
        # round 6: the layer's seven launches leave from ONE library call (csrc/encoder.hip) -- same descriptors, a third of the host time
        fast = planes and flash and one_qkv and ops.layer_entry_ok()
        if flash:
            pqkv = ops.pair_empty(ntok, 3 * hid, dev) if attn_pair else ops.planes_empty(ntok, 3 * hid, dev)
        if planes:                 # operands split into bf16 planes once (csrc/gemm_planes.hip), weights once per optimizer step
            if not pair_bwd and not (one_qkv and xq is not None and not any(ctx.needs_input_grad)):
                # (the bf16 planes of x: the QKV product's operand without the pair form, and its weight gradient's)
                px = ops.Planes(xpl, ntok, hid, xpl.shape[2]) if (xpl is not None and not carrier_is_pair) else ops.split_planes(x)
            if one_qkv:
                if fused_qkv:
                    wqkv_pl, bqkv_t = ops.weight_planes(wq, view=_stack3(wq), also=(wk, wv), pair=xq is not None), _stack3(bq)
                else:
                    wqkv_pl, bqkv_t = ops.stacked_qkv(wq, wk, wv, bq, bk, bv, pair=xq is not None)
                tile_qkv = ptile(3 * hid) if xq is not None else ops._dense_tile(ntok, 3 * hid)
            if fast:
                pass
            elif one_qkv and xq is not None:
                ops.plane_gemm(xq, wqkv_pl, qkv, bias=bqkv_t,
                               out_planes=None if attn_pair else pqkv, out_pair=pqkv if attn_pair else None, tile=tile_qkv, form=1)
            elif one_qkv:
                ops.plane_gemm(px, wqkv_pl, qkv, bias=bqkv_t, out_planes=pqkv, tile=tile_qkv)
            else:              # parameters not laid out back to back (no flat buffers): one product per projection, same kernel
                for j, (w, b) in enumerate(((wq, bq), (wk, bk), (wv, bv))):
                    ops.plane_gemm(px, ops.weight_planes(w), None if flash else qkv[:, j * hid:(j + 1) * hid], bias=b,
                                   out_planes=pqkv.col_block(j * hid, hid) if flash else None, tile=ops._dense_tile(ntok, hid))
        elif fused_qkv:            # one [ntok,hid] x [3*hid,hid]^T GEMM over the stacked projections
            ops.linear_fwd(x, _stack3(wq), _stack3(bq), out=qkv)
        else:
            for j, (w, b) in enumerate(((wq, bq), (wk, bk), (wv, bv))):
                ops.gemm_raw(ntok, hid, hid, x, hid, OP_DENSE_K, w, hid, OP_DENSE_K, qkv, 3 * hid, bias=b, c_ptr_off=j * hid)
        sid = layer * 8
        ctxv = torch.empty((ntok, hid), device=dev, dtype=f32)
        if flash:
            # fused attention (csrc/attn.hip): scores, softmax, dropout and P V in one pass, nothing of size L x L is stored
            # row statistics (m, 1 / l) and, in backward, delta: zero in the padding rows; one zeroed pool per step for all layers
            pool = getattr(meta, "stat_pool", None)
            st = pool[layer] if pool is not None and layer < pool.shape[0] else torch.zeros((3, H, meta.ntok_pad), device=dev, dtype=f32)
            lse = st[:2]
            ctx.delta_buf = st[2]
            # (the keeps of every layer of the step were drawn by ONE launch in front of the encoder when the generator could: mask_pool)
            mpool = getattr(meta, "mask_pool", None)
            masks = (mpool[layer] if (mpool is not None and layer < len(mpool)) else ops.attn_mask(meta, p, seed, sid + 0)) if p > 0 else None
            kbar = torch.empty((ntok, hid), device=dev, dtype=f32) if any(ctx.needs_input_grad) else None
            # (all-pair backward: O also as fp16-pair planes -- the B operand of the output projection's weight gradient, saved instead of
            #  the split pass the backward used to run over the fp32 O)
            pctxq = ops.pair_empty(ntok, hid, dev) if (pair_bwd and any(ctx.needs_input_grad)) else None
            # (autocast region: the output projection multiplies the hi plane of those, no bf16 planes of O are written)
            amp_ao = pctxq is not None and ops.amp_one_product()
            pctx = None if amp_ao else ops.planes_empty(ntok, hid, dev)
            if not fast:
                ops.attn(meta, ATTN_FWD, pqkv, None, ctxv, lse, None, masks, 1.0 / (dh ** 0.5), p, kbar=kbar, out_planes=pctx, out_pair=pctxq)
        else:
            # scores -> probabilities (in place), grouped over (sequence, head)
            P = torch.empty((meta.s_elems,), device=dev, dtype=f32)
            ops.gemm_raw(0, 0, 0, qkv, 3 * hid, OP_DENSE_K, qkv, 3 * hid, OP_DENSE_K, P, meta.ld, grp=meta.t_qk, ngroups=meta.ngroups,
                         grp_max=(meta.maxlen, meta.maxlen), b_ptr_off=hid, bk=16 if dh <= 128 else 0, tile=64064 if dh <= 128 else 0)
            ops.softmax_fwd(P, meta.soff, meta.lens, meta.ldp, meta.ngroups, H, meta.maxlen, 1.0 / (dh ** 0.5), p, seed, sid + 0)
            ops.gemm_raw(0, 0, 0, P, meta.ld, OP_DENSE_K, qkv, 3 * hid, OP_DENSE_R, ctxv, hid, grp=meta.t_pv, ngroups=meta.ngroups,
                         grp_max=(meta.maxlen, dh), b_ptr_off=2 * hid, a_relu_scale=1.0 / (1.0 - p))
        py = None
        if planes:
            # (the attention-output projection stays on the six-product form, its operand's bf16 planes come from the attention kernel:
            #  measured at full scale, moving it to the pair form as well raised the gradient error of the ill-conditioned trunk
            #  convolutions from 6.3e-4 to 1.0e-3 of the reference -- the forward feeds everything; the backward products do not)
            ao = torch.empty((ntok, hid), device=dev, dtype=f32)
            ao_pair = bool(flash and amp_ao)
            if ao_pair:
                wo_pl, tile_ao = ops.weight_planes(wo, pair=True), ops.pair_tile(ntok, hid) or 128129
            else:
                if pctx is None:
                    pctx = ops.split_planes(ctxv)
                wo_pl, tile_ao = ops.weight_planes(wo), ops._dense_tile(ntok, hid)
            # (the bf16 planes of x1 / gelu(h) / y are operands of the BACKWARD's six-product form: not written when that backward runs the
            #  pair form, nor when there is no backward at all and the forward reads the pair planes)
            keep3 = not pair_bwd and not (pair and nograd)
            px1 = ops.planes_empty(ntok, hid, dev) if keep3 else None
            px1q = ops.pair_empty(ntok, hid, dev) if pair else None
            inter = wi.shape[0]
            h = torch.empty((ntok, inter), device=dev, dtype=f32)
            # gelu(h) leaves the FFN1 epilogue as planes only (the A operand of FFN2 and, untransposed, of its weight gradient)
            pg = ops.planes_empty(ntok, inter, dev) if keep3 else None
            pgq = ops.pair_empty(ntok, inter, dev) if pair else None
            fo = torch.empty((ntok, hid), device=dev, dtype=f32)
            wi_pl, wo2_pl = ops.weight_planes(wi, pair=pair), ops.weight_planes(wo2, pair=pair)
            if pair:
                tile_f1, tile_f2 = ptile(inter, True), ptile(hid)
            else:
                tile_f1, tile_f2 = ops._dense_tile(ntok, inter, True), ops._dense_tile(ntok, hid)
            if fast:
                x1, xh1, y, xh2 = (torch.empty_like(x) for _ in range(4))
                rs1, rs2 = (torch.empty((ntok,), device=dev, dtype=f32) for _ in range(2))
                py = ops.planes_empty(ntok, hid, dev) if keep3 else None
                pyq = ops.pair_empty(ntok, hid, dev) if pair else None
                ops.bert_layer_fwd(meta, eps=eps, p=p, seed=seed, sid=sid, x=x, xa=xq if xq is not None else px, pair_qkv=xq is not None,
                                   wqkv=wqkv_pl, bqkv=bqkv_t, tile_qkv=tile_qkv, pqkv=pqkv, attn_pair=attn_pair, ctxv=ctxv, lse=lse, kbar=kbar,
                                   pctx=pctx, pctxq=pctxq, masks=masks, scale=1.0 / (dh ** 0.5), wo=wo_pl, bo=bo, ao_pair=ao_pair, tile_ao=tile_ao,
                                   ao=ao, g1=g1, b1=b1, x1=x1, xh1=xh1, rs1=rs1, px1=px1, px1q=px1q, wi=wi_pl, bi=bi, wo2=wo2_pl, bo2=bo2,
                                   pair_ffn=pair, tile_ffn1=tile_f1, tile_ffn2=tile_f2, h=h, pg=pg, pgq=pgq, fo=fo, g2=g2, b2=b2, y=y, xh2=xh2,
                                   rs2=rs2, py=py, pyq=pyq)
            else:
                ops.plane_gemm(pctxq if ao_pair else pctx, wo_pl, ao, bias=bo, tile=tile_ao, form=1 if ao_pair else 0)
                x1, xh1, rs1 = ops.dropout_add_ln_fwd(ao, x, g1, b1, eps, p, seed, sid + 1, out_planes=px1, out_pair=px1q)
                if pair:
                    ops.plane_gemm(px1q, wi_pl, h, bias=bi, epi=EPI_GELU_DUAL, out_planes=pg, out_pair=pgq, tile=tile_f1, form=1)
                    ops.plane_gemm(pgq, wo2_pl, fo, bias=bo2, tile=tile_f2, form=1)
                else:
                    ops.plane_gemm(px1, wi_pl, h, bias=bi, epi=EPI_GELU_DUAL, out_planes=pg, tile=tile_f1)
                    ops.plane_gemm(pg, wo2_pl, fo, bias=bo2, tile=tile_f2)
            if pair and not pair_bwd:
                pgq = px1q = None
            g = None
        else:
            ao = ops.linear_fwd(ctxv, wo, bo)
            x1, xh1, rs1 = ops.dropout_add_ln_fwd(ao, x, g1, b1, eps, p, seed, sid + 1)
            h, g = ops.linear_fwd(x1, wi, bi, EPI_GELU_DUAL)
            fo = ops.linear_fwd(g, wo2, bo2)
        if not fast:
            pyq = None
            if planes:
                py = ops.planes_empty(ntok, hid, dev) if keep3 else None
                pyq = ops.pair_empty(ntok, hid, dev) if pair else None
            y, xh2, rs2 = ops.dropout_add_ln_fwd(fo, x1, g2, b2, eps, p, seed, sid + 2, out_planes=py, out_pair=pyq)
        ctx.meta, ctx.cfg = meta, (eps, p, seed, sid)
        ctx.planes, ctx.flash, ctx.pair_bwd = planes, flash, pair_bwd
        ctx.w_refs = (wq, wk, wv, wo, wi, wo2)
        ctx.b_refs = (bq, bk, bv, bo, bi, bo2, g1, b1, g2, b2)
        if not any(ctx.needs_input_grad):
            pass                                   # (nothing is differentiated -- inference, validation: nothing to keep)
        elif planes:
            # backward needs the activations only as GEMM operands: their planes stand in for x / ctx / x1 / gelu(h)
            ctx.pl_shape = (ntok, hid, wi.shape[0])
            if pair_bwd:
                ctx.masks = masks
                ctx.save_for_backward(wq, wk, wv, wo, g1, wi, wo2, g2, pqkv.buf, ctxv, xh1, rs1, h, xh2, rs2, xq.buf, px1q.buf, pgq.buf, lse, kbar,
                                      None if pctxq is None else pctxq.buf)
            elif flash:
                ctx.masks = masks
                ctx.save_for_backward(wq, wk, wv, wo, g1, wi, wo2, g2, pqkv.buf, ctxv, xh1, rs1, h, xh2, rs2, px.buf, pctx.buf, px1.buf, pg.buf, lse, kbar)
            else:
                ctx.save_for_backward(wq, wk, wv, wo, g1, wi, wo2, g2, qkv, P, xh1, rs1, h, xh2, rs2, px.buf, pctx.buf, px1.buf, pg.buf)
        else:
            ctx.save_for_backward(x, wq, wk, wv, wo, g1, wi, wo2, g2, qkv, P, ctxv, xh1, rs1, x1, h, g, xh2, rs2)
        if py is None and pyq is not None:            # all-pair path: the pair planes themselves travel to the next layer
            ctx.mark_non_differentiable(pyq.buf)
            ctx.set_materialize_grads(False)
            return y, pyq.buf
        if py is None:
            return y, None
        ctx.mark_non_differentiable(py.buf)
        ctx.set_materialize_grads(False)       # (no zero tensor for the planes output's gradient slot: 19 MB fill per layer)
        if pyq is not None:
            py.buf._vbg_pair = pyq             # (travels on the tensor object to the next layer's first product)
        return y, py.buf

    @staticmethod
    def _backward_pair(ctx, dy):
        """backward of the all-pair path: every product on two fp16 pieces.  A gradient operand is split by a pass of its own once its
        largest magnitude is known (LayerNorm backward -> fp32 -> vbg_amax -> scaled split with the bias column sums; the GELU-gradient
        product reports the maximum of what it stores), scaled by the power of two of that maximum; products scale back (exact)."""
        (wq, wk, wv, wo, g1, wi, wo2, g2, bqkv, ctxv, xh1, rs1, h, xh2, rs2, bx, bx1, bg, lse, kbar, bctxq) = ctx.saved_tensors
        meta = ctx.meta
        eps, p, seed, sid = ctx.cfg
        ntok, hid, inter = ctx.pl_shape
        H, dh = meta.heads, meta.dh
        dev = h.device
        mk = lambda buf, cols: ops.Planes(buf, ntok, cols, buf.shape[2])
        qx, qx1, qg = mk(bx, hid), mk(bx1, hid), mk(bg, inter)
        # (the attention output as the B operand of the output projection's weight gradient: pair planes written by the forward kernel)
        qctx = mk(bctxq, hid) if bctxq is not None else ops.split_planes_pair(ctxv)
        rbq, rbk, rbv, rbo, rbi, rbo2, rg1, rb1, rg2, rb2 = ctx.b_refs
        rq, rk, rv, ro, ri, ro2 = ctx.w_refs
        tile = lambda n, wide=False: ops.pair_tile(ntok, n, wide)
        # ---- LayerNorm 2 backward -> dfo.  Round 4: straight as pair planes scaled by a rigorous bound (max |dy| x max |gamma| x max rstd x
        #      (2 + sqrt H) / keep; csrc/rowops.hip PL = 2) with the bias gradient of the FFN output projection riding along -- no fp32 dfo,
        #      no split pass.  s_dfo = true max |dfo| (the next bound's input), s_dfo_ref = the scale the planes were written with.
        dg2, db2, sunk2 = _affine_dest(rg2, rb2)
        s_dfo = ops.amax_slot(dev)
        bound = ops.bound_planes_enabled() and sunk2 and hid % 256 == 0
        dyc = _c(dy)
        if bound:
            s_dy = _amax_tag(dy)
            if s_dy is None:
                s_dy = ops.amax(dyc)                  # (the top layer: its output gradient is a sum autograd formed)
            s_dfo_ref = ops.amax_slot(dev)
            qdfo, dx1 = ops.dropout_add_ln_bwd_pair(dyc, xh2, rs2, g2, p, seed, sid + 2, dg2, db2, wgrad_dest(rbo2), s_dy, s_dfo, s_dfo_ref)
        else:
            s_dfo_ref = s_dfo
            dfo, dx1 = ops.dropout_add_ln_bwd(dyc, xh2, rs2, g2, p, seed, sid + 2, dg2, db2, dx_amax=s_dfo)
            qdfo = ops.split_planes_pair(dfo, amax_slot_=s_dfo, colsum_out=wgrad_dest(rbo2))
            del dfo
        dg2, db2 = _affine_done(rg2, rb2, dg2, db2, sunk2)
        wgrad_done(rbo2)
        # ---- dL/dh = (dfo Wo2) o gelu'(h) leaves the product's epilogue as pair planes (round 4): scaled by the power of two of a rigorous
        #      BOUND -- max |dfo| (measured: s_dfo) * the largest column L1 norm of Wo2 (once per weight version) * max gelu' (1.129) -- instead
        #      of a measured maximum, so it needs no fp32 round trip and no split pass; s_dh receives the bound (the consumers' scale)
        s_dh = ops.amax_slot(dev)
        if ops.bound_planes_enabled():
            qdh = ops.pair_empty(ntok, inter, dev)
            ops.plane_gemm(qdfo, ops.weight_planes(ro2, True, view=wo2, pair=True), None, epi=EPI_MUL_GELU_GRAD, C2=h, tile=tile(inter, True), form=1,
                           a_amax=s_dfo_ref, out_pair=qdh, q_ref_in=s_dfo, q_l1=ops.weight_col_l1max(ro2, view=wo2), q_mul=1.13 * 1.01, q_ref_out=s_dh,
                           colsum_out=wgrad_dest(rbi))
        else:
            dh_ = torch.empty((ntok, inter), device=dev, dtype=f32)
            ops.plane_gemm(qdfo, ops.weight_planes(ro2, True, view=wo2, pair=True), dh_, epi=EPI_MUL_GELU_GRAD, C2=h, tile=tile(inter, True), form=1,
                           a_amax=s_dfo_ref, c_amax=s_dh)
            qdh = ops.split_planes_pair(dh_, amax_slot_=s_dh, colsum_out=wgrad_dest(rbi))
            del dh_
        wgrad_done(rbi)
        s_dx1 = ops.amax_slot(dev) if bound else None           # (max |dx1| rides on the product that completes it: LayerNorm 1's bound)
        ops.plane_gemm(qdh, ops.weight_planes(ri, True, view=wi, pair=True), dx1, accumulate=True, tile=tile(hid), form=1, a_amax=s_dh, c_amax=s_dx1)
        # ---- LayerNorm 1 backward -> dao (pair planes by the same bound, or fp32 + split)
        dg1, db1, sunk1 = _affine_dest(rg1, rb1)
        s_dao = ops.amax_slot(dev)
        if bound and sunk1:
            s_dao_ref = ops.amax_slot(dev)
            qdao, dx = ops.dropout_add_ln_bwd_pair(dx1, xh1, rs1, g1, p, seed, sid + 1, dg1, db1, wgrad_dest(rbo), s_dx1, s_dao, s_dao_ref)
        else:
            s_dao_ref = s_dao
            dao, dx = ops.dropout_add_ln_bwd(dx1, xh1, rs1, g1, p, seed, sid + 1, dg1, db1, dx_amax=s_dao)
            qdao = ops.split_planes_pair(dao, amax_slot_=s_dao, colsum_out=wgrad_dest(rbo))
            del dao
        dg1, db1 = _affine_done(rg1, rb1, dg1, db1, sunk1)
        wgrad_done(rbo)
        # ---- d(context), the dO operand of the fused attention backward: as the planes the forward's q / k / v planes call for -- fp16-pair
        #      planes scaled by a bound (max |dao| x the largest column L1 norm of W_o; the attention kernels read the scale from s_dctx),
        #      or three bf16 planes (six-product attention)
        pqkv = ops.Planes(bqkv, ntok, 3 * hid, bqkv.shape[2])
        s_dctx = None
        if bqkv.shape[0] == 2:
            pdctx = ops.pair_empty(ntok, hid, dev)
            s_dctx = ops.amax_slot(dev)
            ops.plane_gemm(qdao, ops.weight_planes(ro, True, view=wo, pair=True), None, tile=tile(hid), form=1, a_amax=s_dao_ref, out_pair=pdctx,
                           q_ref_in=s_dao, q_l1=ops.weight_col_l1max(ro, view=wo), q_mul=1.01, q_ref_out=s_dctx)
        else:
            pdctx = ops.planes_empty(ntok, hid, dev)
            ops.plane_gemm(qdao, ops.weight_planes(ro, True, view=wo, pair=True), None, out_planes=pdctx, tile=tile(hid), form=1, a_amax=s_dao_ref)
        delta = ctx.delta_buf
        dqkv = torch.empty((ntok, 3 * hid), device=dev, dtype=f32)
        sc = 1.0 / (dh ** 0.5)
        s_dqkv = ops.amax_slot(dev)                   # (the largest magnitude of d(qkv) rides on the two kernels that write it)
        ops.attn(meta, ATTN_DQ, pqkv, pdctx, dqkv, lse, delta, ctx.masks, sc, p, kbar=kbar, o=ctxv, out_amax=s_dqkv, do_amax=s_dctx)
        ops.attn(meta, ATTN_DKV, pqkv, pdctx, dqkv, lse, delta, ctx.masks, sc, p, out_amax=s_dqkv, do_amax=s_dctx)
        gq = [wgrad_dest(t) for t in (rq, rk, rv, rbq, rbk, rbv)]
        qdqkv = ops.split_planes_pair(dqkv, amax_slot_=s_dqkv, colsum_out=_stack3(gq[3]))
        del dqkv
        s_dx = ops.amax_slot(dev) if bound else None            # (max |dx| for the layer below: its LayerNorm 2 bound)
        ops.plane_gemm(qdqkv, ops.weight_planes(rq, True, view=_stack3(wq), also=(rk, rv), pair=True), dx, accumulate=True, tile=tile(hid), form=1,
                       a_amax=s_dqkv, c_amax=s_dx)
        if s_dx is not None:
            dx._vbg_amax = (s_dx, dx._version)
        # ---- the four weight gradients: one grouped TN launch on pair planes
        jobs = [(qdfo, qg, wgrad_dest(ro2)), (qdh, qx1, wgrad_dest(ri)), (qdao, qctx, wgrad_dest(ro)), (qdqkv, qx, _stack3(gq[0]))]
        scales = [s_dfo_ref, s_dh, s_dao_ref, s_dqkv]
        if ops.wgrad_stream_enabled():
            # nothing in the rest of the backward waits for these 216 tiles: on the weight-gradient stream they share the chip with the
            # data-gradient products of the layer below (198 / 408 / 594 tiles: 0.77 / 0.80 / 0.77 of their rounds).  The operands stay
            # reserved for that stream when this node releases them; the callback (end of backward) / FlatReducer's staging stream join it.
            cur, ws = torch.cuda.current_stream(dev), ops.side_stream(dev, "wgrad")
            ws.wait_stream(cur)
            with torch.cuda.stream(ws):
                ops.plane_gemm_grouped(jobs, trans=True, accumulate=True, form=1, a_amax=scales)
                for a_, b_, _ in jobs:
                    a_.buf.record_stream(ws)
                    b_.buf.record_stream(ws)
                for t in scales:
                    t.record_stream(ws)
                for t in (ro2, ri, ro, rq, rk, rv, rbq, rbk, rbv):
                    wgrad_done(t)
            torch.autograd.Variable._execution_engine.queue_callback(lambda: torch.cuda.current_stream(dev).wait_stream(ws))
        else:
            ops.plane_gemm_grouped(jobs, trans=True, accumulate=True, form=1, a_amax=scales)
            for t in (ro2, ri, ro, rq, rk, rv, rbq, rbk, rbv):
                wgrad_done(t)
        return (dx, None, None, None, None, None, None, None, None, None, dg1, db1, None, None, None, None, dg2, db2, None, None, None, None, None)

    @staticmethod
    def _backward_planes(ctx, dy):
        """backward of the plane path: every dy is split once (the A operand of its data-gradient product) and the four weight
        gradients of the layer run as ONE grouped TN launch from the untransposed planes of dy and of the saved activations"""
        flash = ctx.flash
        if flash:
            (wq, wk, wv, wo, g1, wi, wo2, g2, bqkv, ctxv, xh1, rs1, h, xh2, rs2, bx, bctx, bx1, bg, lse, kbar) = ctx.saved_tensors
        else:
            (wq, wk, wv, wo, g1, wi, wo2, g2, qkv, P, xh1, rs1, h, xh2, rs2, bx, bctx, bx1, bg) = ctx.saved_tensors
        meta = ctx.meta
        eps, p, seed, sid = ctx.cfg
        ntok, hid, inter = ctx.pl_shape
        H, dh = meta.heads, meta.dh
        dev = h.device
        mk = lambda buf, cols: ops.Planes(buf, ntok, cols, buf.shape[2])
        px, pctx, px1, pg = mk(bx, hid), mk(bctx, hid), mk(bx1, hid), mk(bg, inter)
        rbq, rbk, rbv, rbo, rbi, rbo2, rg1, rb1, rg2, rb2 = ctx.b_refs
        rq, rk, rv, ro, ri, ro2 = ctx.w_refs
        dg2, db2, sunk2 = _affine_dest(rg2, rb2)
        dst = wgrad_dest(rbo2)
        if dst is not None and hid % 32 == 0:     # dx leaves the LayerNorm backward as planes, its column sums as the bias gradient
            pdfo, dx1 = ops.dropout_add_ln_bwd_planes(_c(dy), xh2, rs2, g2, p, seed, sid + 2, dg2, db2, dst)
            wgrad_done(rbo2)
            dbo2 = None
            dg2, db2 = _affine_done(rg2, rb2, dg2, db2, sunk2)
        else:
            dfo, dx1 = ops.dropout_add_ln_bwd(_c(dy), xh2, rs2, g2, p, seed, sid + 2, dg2, db2)
            dg2, db2 = _affine_done(rg2, rb2, dg2, db2, sunk2)
            pdfo, dbo2 = _split_with_bias_grad(rbo2, dfo)
        # dL/dh = (dfo Wo2) o gelu'(h): the GELU backward rides in the product's epilogue
        dst_bi = wgrad_dest(rbi)
        if dst_bi is not None:
            # ... and dL/dh leaves as planes only (it is only ever a plane operand), its column sums go into the bias gradient
            pdh = ops.planes_empty(ntok, inter, dev)
            ops.plane_gemm(pdfo, ops.weight_planes(ro2, True, view=wo2), None, epi=EPI_MUL_GELU_GRAD, C2=h, out_planes=pdh, colsum_out=dst_bi,
                           tile=ops._dense_tile(ntok, inter, True))
            wgrad_done(rbi)
            dbi = None
        else:
            dh_ = ops.plane_gemm(pdfo, ops.weight_planes(ro2, True, view=wo2), torch.empty_like(h), epi=EPI_MUL_GELU_GRAD, C2=h,
                                 tile=ops._dense_tile(ntok, inter, True))
            pdh, dbi = _split_with_bias_grad(rbi, dh_)
        ops.plane_gemm(pdh, ops.weight_planes(ri, True, view=wi), dx1, accumulate=True, tile=ops._dense_tile(ntok, hid))
        dg1, db1, sunk1 = _affine_dest(rg1, rb1)
        dst = wgrad_dest(rbo)
        if dst is not None and hid % 32 == 0:
            pdao, dx = ops.dropout_add_ln_bwd_planes(dx1, xh1, rs1, g1, p, seed, sid + 1, dg1, db1, dst)
            wgrad_done(rbo)
            dbo = None
            dg1, db1 = _affine_done(rg1, rb1, dg1, db1, sunk1)
        else:
            dao, dx = ops.dropout_add_ln_bwd(dx1, xh1, rs1, g1, p, seed, sid + 1, dg1, db1)
            dg1, db1 = _affine_done(rg1, rb1, dg1, db1, sunk1)
            pdao, dbo = _split_with_bias_grad(rbo, dao)
        pdctx = ops.planes_empty(ntok, hid, dev) if flash else None
        dctx = ops.plane_gemm(pdao, ops.weight_planes(ro, True, view=wo), torch.empty((ntok, hid), device=dev, dtype=f32), out_planes=pdctx,
                              tile=ops._dense_tile(ntok, hid))
        if flash:
            # fused attention backward: delta = rowsum(dO o O), then dQ (queries stationary) and dK / dV (keys stationary), each
            # recomputing its score tile from the q / k / v planes and the saved log-sum-exp
            pqkv = ops.Planes(bqkv, ntok, 3 * hid, bqkv.shape[2])
            delta = ctx.delta_buf                      # zero in the padding rows; the DQ pass fills it
            dqkv = torch.empty((ntok, 3 * hid), device=dev, dtype=f32)
            sc = 1.0 / (dh ** 0.5)
            ops.attn(meta, ATTN_DQ, pqkv, pdctx, dqkv, lse, delta, ctx.masks, sc, p, kbar=kbar, o=ctxv)
            ops.attn(meta, ATTN_DKV, pqkv, pdctx, dqkv, lse, delta, ctx.masks, sc, p)
        else:
            dP = torch.empty_like(P)
            ops.gemm_raw(0, 0, 0, dctx, hid, OP_DENSE_K, qkv, 3 * hid, OP_DENSE_K, dP, meta.ld, grp=meta.t_dp, ngroups=meta.ngroups,
                         grp_max=(meta.maxlen, meta.maxlen), b_ptr_off=2 * hid, bk=16 if dh <= 128 else 0, tile=64064 if dh <= 128 else 0)
            dqkv = torch.empty_like(qkv)
            ops.gemm_raw(0, 0, 0, P, meta.ld, OP_DENSE_R, dctx, hid, OP_DENSE_R, dqkv, 3 * hid, grp=meta.t_dv, ngroups=meta.ngroups,
                         grp_max=(meta.maxlen, dh), c_ptr_off=2 * hid, a_relu_scale=1.0 / (1.0 - p))
            ops.softmax_bwd(P, dP, meta.soff, meta.lens, meta.ldp, meta.ngroups, H, meta.maxlen, 1.0 / (dh ** 0.5), p)
            ops.gemm_raw(0, 0, 0, dP, meta.ld, OP_DENSE_K, qkv, 3 * hid, OP_DENSE_R, dqkv, 3 * hid, grp=meta.t_dq, ngroups=meta.ngroups,
                         grp_max=(meta.maxlen, dh), b_ptr_off=hid)
            ops.gemm_raw(0, 0, 0, dP, meta.ld, OP_DENSE_R, qkv, 3 * hid, OP_DENSE_R, dqkv, 3 * hid, grp=meta.t_dk, ngroups=meta.ngroups,
                         grp_max=(meta.maxlen, dh), c_ptr_off=hid)
        stacked = _back_to_back(wq, wk, wv)
        gq = [wgrad_dest(t) for t in (rq, rk, rv, rbq, rbk, rbv)]
        qkv_sunk = stacked and all(t is not None for t in gq) and _back_to_back(*gq[:3]) and _back_to_back(*gq[3:])
        pdqkv = ops.split_planes(dqkv, colsum_out=_stack3(gq[3]) if qkv_sunk else None)
        if stacked:
            ops.plane_gemm(pdqkv, ops.weight_planes(rq, True, view=_stack3(wq), also=(rk, rv)), dx, accumulate=True, tile=ops._dense_tile(ntok, hid))
        else:
            for j, (w, wr) in enumerate(((wq, rq), (wk, rk), (wv, rv))):
                ops.plane_gemm(pdqkv.col_block(j * hid, hid), ops.weight_planes(wr, True, view=w), dx, accumulate=True, tile=ops._dense_tile(ntok, hid))
        # ---- the four weight gradients: one grouped TN launch (432 tiles at bert-base: two rounds on 256 CUs instead of four
        #      launches of 36-144 tiles each) ------------------------------------------------------------------------------
        dw_qkv = _stack3(gq[0]) if qkv_sunk else torch.zeros((3 * hid, hid), device=dev, dtype=f32)
        dests, fresh = [], []
        for wp in (ro2, ri, ro):
            d_ = wgrad_dest(wp)
            fresh.append(d_ is None)
            dests.append(d_ if d_ is not None else torch.zeros_like(wp))
        jobs = [(pdfo, pg, dests[0]), (pdh, px1, dests[1]), (pdao, pctx, dests[2]), (pdqkv, px, dw_qkv)]
        if qkv_sunk and not any(fresh) and ctx.side_ok and ops.wgrad_stream_enabled():
            # every destination is a flat gradient buffer nothing else in this backward touches: the launch goes on the
            # weight-gradient stream and shares the chip with the backward of the layer below.  The operands stay reserved for
            # that stream when this node releases them; JoinSideFn's callback / FlatReducer's staging stream wait for it.
            cur, ws = torch.cuda.current_stream(dev), ops.side_stream(dev, "wgrad")
            ws.wait_stream(cur)
            with torch.cuda.stream(ws):
                ops.plane_gemm_grouped(jobs, trans=True, accumulate=True)
                for a_, b_, _ in jobs:
                    a_.buf.record_stream(ws)
                    b_.buf.record_stream(ws)
                for t in (ro2, ri, ro, rq, rk, rv, rbq, rbk, rbv):
                    wgrad_done(t)
            return (dx, None, None, None, None, None, None, None, None, dbo, dg1, db1, None, dbi, None, dbo2, dg2, db2, None, None, None, None, None)
        ops.plane_gemm_grouped(jobs, trans=True, accumulate=True)
        for wp, fr in zip((ro2, ri, ro), fresh):
            if not fr:
                wgrad_done(wp)
        dwo2, dwi, dwo = (dests[i] if fresh[i] else None for i in range(3))
        if qkv_sunk:
            for t in (rq, rk, rv, rbq, rbk, rbv):
                wgrad_done(t)
            return (dx, None, None, None, None, None, None, None, dwo, dbo, dg1, db1, dwi, dbi, dwo2, dbo2, dg2, db2, None, None, None, None, None)
        dws, dbs = [], []
        for j, (wr, br) in enumerate(((rq, rbq), (rk, rbk), (rv, rbv))):
            dj = dw_qkv[j * hid:(j + 1) * hid]
            dst = wgrad_dest(wr)
            if dst is not None:            # sunk but not stacked: add the block into the parameter's own gradient view
                dst.add_(dj)
                wgrad_done(wr)
                dj = None
            dws.append(dj)
            dbs.append(_bias_grad(br, dqkv[:, j * hid:(j + 1) * hid]))
        return (dx, None, dws[0], dbs[0], dws[1], dbs[1], dws[2], dbs[2], dwo, dbo, dg1, db1, dwi, dbi, dwo2, dbo2,
                dg2, db2, None, None, None, None, None)

    @staticmethod
    def backward(ctx, dy, _dplanes=None):
        if ctx.planes:
            return BertLayerFn._backward_pair(ctx, dy) if ctx.pair_bwd else BertLayerFn._backward_planes(ctx, dy)
        (x, wq, wk, wv, wo, g1, wi, wo2, g2, qkv, P, ctxv, xh1, rs1, x1, h, g, xh2, rs2) = ctx.saved_tensors
        meta = ctx.meta
        eps, p, seed, sid = ctx.cfg
        ntok, hid = x.shape
        H, dh = meta.heads, meta.dh
        dev = x.device
        rbq, rbk, rbv, rbo, rbi, rbo2, rg1, rb1, rg2, rb2 = ctx.b_refs
        dg2, db2, sunk2 = _affine_dest(rg2, rb2)
        dfo, dx1 = ops.dropout_add_ln_bwd(_c(dy), xh2, rs2, g2, p, seed, sid + 2, dg2, db2)
        dg2, db2 = _affine_done(rg2, rb2, dg2, db2, sunk2)
        # FFN
        rq, rk, rv, ro, ri, ro2 = ctx.w_refs
        dwo2 = _linear_wgrad(ro2, dfo, g)
        dbo2 = _bias_grad(rbo2, dfo)
        dh_ = ops.linear_dgrad(dfo, wo2)
        ops.gelu_bwd_(h, dh_)
        dwi = _linear_wgrad(ri, dh_, x1)
        dbi = _bias_grad(rbi, dh_)
        ops.linear_dgrad(dh_, wi, out=dx1, accumulate=True)
        # LN1
        dg1, db1, sunk1 = _affine_dest(rg1, rb1)
        dao, dx = ops.dropout_add_ln_bwd(dx1, xh1, rs1, g1, p, seed, sid + 1, dg1, db1)
        dg1, db1 = _affine_done(rg1, rb1, dg1, db1, sunk1)
        dwo = _linear_wgrad(ro, dao, ctxv)
        dbo = _bias_grad(rbo, dao)
        dctx = ops.linear_dgrad(dao, wo)
        # attention backward (grouped GEMMs + row softmax backward)
        dP = torch.empty_like(P)
        ops.gemm_raw(0, 0, 0, dctx, hid, OP_DENSE_K, qkv, 3 * hid, OP_DENSE_K, dP, meta.ld, grp=meta.t_dp, ngroups=meta.ngroups,
                     grp_max=(meta.maxlen, meta.maxlen), b_ptr_off=2 * hid, bk=16 if dh <= 128 else 0, tile=64064 if dh <= 128 else 0)
        dqkv = torch.empty_like(qkv)
        ops.gemm_raw(0, 0, 0, P, meta.ld, OP_DENSE_R, dctx, hid, OP_DENSE_R, dqkv, 3 * hid, grp=meta.t_dv, ngroups=meta.ngroups,
                     grp_max=(meta.maxlen, dh), c_ptr_off=2 * hid, a_relu_scale=1.0 / (1.0 - p))
        ops.softmax_bwd(P, dP, meta.soff, meta.lens, meta.ldp, meta.ngroups, H, meta.maxlen, 1.0 / (dh ** 0.5), p)
        ops.gemm_raw(0, 0, 0, dP, meta.ld, OP_DENSE_K, qkv, 3 * hid, OP_DENSE_R, dqkv, 3 * hid, grp=meta.t_dq, ngroups=meta.ngroups,
                     grp_max=(meta.maxlen, dh), b_ptr_off=hid)
        ops.gemm_raw(0, 0, 0, dP, meta.ld, OP_DENSE_R, qkv, 3 * hid, OP_DENSE_R, dqkv, 3 * hid, grp=meta.t_dk, ngroups=meta.ngroups,
                     grp_max=(meta.maxlen, dh), c_ptr_off=hid)
        # QKV projections
        gq = [wgrad_dest(t) for t in (rq, rk, rv, rbq, rbk, rbv)]
        if (all(t is not None for t in gq) and _back_to_back(wq, wk, wv) and _back_to_back(*gq[:3]) and _back_to_back(*gq[3:])):
            ops.linear_wgrad(dqkv, x, _stack3(gq[0]), accumulate=True)
            ops.colsum(dqkv, out=_stack3(gq[3]), accumulate=True)
            ops.linear_dgrad(dqkv, _stack3(wq), out=dx, accumulate=True)
            for t in (rq, rk, rv, rbq, rbk, rbv):
                wgrad_done(t)
            return (dx, None, None, None, None, None, None, None, dwo, dbo, dg1, db1, dwi, dbi, dwo2, dbo2, dg2, db2, None, None, None, None, None)
        dws, dbs = [], []
        for j, (w, wr, br) in enumerate(((wq, rq, rbq), (wk, rk, rbk), (wv, rv, rbv))):
            dj = dqkv[:, j * hid:(j + 1) * hid]
            dws.append(_linear_wgrad(wr, dj, x))
            dbs.append(_bias_grad(br, dj))
            ops.linear_dgrad(dj, w, out=dx, accumulate=True)
        return (dx, None, dws[0], dbs[0], dws[1], dbs[1], dws[2], dbs[2], dwo, dbo, dg1, db1, dwi, dbi, dwo2, dbo2,
                dg2, db2, None, None, None, None, None)


class StepRootFn(torch.autograd.Function):
    """Identity on the loss ViBERTgridNet.forward returns.  Its backward is the first node of the model's graph to run (behind the
    caller's own scaling of the loss, e.g. GradScaler.scale): it re-arms the flat gradient buffers of the parameter groups the model
    is homed in (vbg.optim.FlatGroup.arm) -- the reference's loop drops every `.grad` between forward and backward
    (`optimizer.zero_grad()`, pipeline/train_val_utils.py:272-273) -- and, for graphs that depend on the data (classifier_mode full:
    a per-class net may see no sample), hands back `None` for the parameters that took no part in this step, as torch would have
    left them (`touched`: a set the groups' post-accumulate hooks fill)."""

    @staticmethod
    def forward(ctx, loss, home):
        ctx.home = home
        return loss.view_as(loss)

    @staticmethod
    def backward(ctx, dloss):
        home = ctx.home
        for g in home.groups:
            g.arm()
        if home.touched is not None:
            home.touched.clear()
            torch.autograd.Variable._execution_engine.queue_callback(home.drop_untouched)
        return dloss, None


class JoinSideFn(torch.autograd.Function):
    """Identity on a tensor that was produced on the side stream (vbg/ops.side_stream) and is consumed on the caller's stream.
    Forward: nothing (the caller has already made its stream wait).  Backward: the gradient passes through unchanged -- the
    autograd engine hands it to the producer's node on the side stream with its own event -- and a callback is queued that makes
    the stream `backward()` was called on wait for the side stream once the whole graph has run: gradients that the side
    stream's nodes wrote straight into the flat gradient buffers (sunk parameters: no AccumulateGrad node the engine could
    synchronise on) are complete for whatever the caller enqueues next (clipping, optimizer steps, all-reduce)."""

    @staticmethod
    def forward(ctx, x):
        ctx.dev = x.device
        return x.view_as(x)

    @staticmethod
    def backward(ctx, dy):
        dev = ctx.dev

        def join():
            cur = torch.cuda.current_stream(dev)
            for s in ops.side_streams():
                if s.device == dev:
                    cur.wait_stream(s)
        torch.autograd.Variable._execution_engine.queue_callback(join)
        return dy


class GatherRowsFn(torch.autograd.Function):
    """rows `idx` of x (`fuse_embeddings[pred_pos_neg_mask]`, field_type_classification_head.py:371); bwd scatters them back"""

    @staticmethod
    def forward(ctx, x, idx):
        x = _c(x)
        ctx.shape = x.shape
        ctx.save_for_backward(idx)
        return ops.gather_rows(x, idx)

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        dx = torch.zeros(ctx.shape, device=dy.device, dtype=f32)
        ops.scatter_rows_add(_c(dy), idx, dx)
        return dx, None


class CrfNllFn(torch.autograd.Function):
    """per-document CRF negative log-likelihood / length (model/crf.py:147-151) for all documents of the batch at once"""

    @staticmethod
    def forward(ctx, em, trans, tags, doc_off, start, stop):
        em, trans = _c(em), _c(trans)
        nll, alpha, logz = ops.crf_nll_fwd(em, tags, doc_off, trans, start, stop)
        ctx.cfg = (start, stop)
        ctx.t_ref = trans
        ctx.save_for_backward(em, trans, tags, doc_off, alpha, logz)
        return nll

    @staticmethod
    def backward(ctx, dnll):
        em, trans, tags, doc_off, alpha, logz = ctx.saved_tensors
        start, stop = ctx.cfg
        dtrans = torch.zeros_like(trans)
        dem = ops.crf_nll_bwd(em, tags, doc_off, trans, start, stop, alpha, logz, _c(dnll).to(f32), dtrans)
        return dem, dtrans, None, None, None, None


# ----------------------------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------------------------
class SelectedCEFn(torch.autograd.Function):
    """sum_i w[t_i] * CE(logits[row(e_i)], t_i) * scale over a selected element list `elem` (int32,
    already chosen by the sampling / OHEM logic); returns a 0-dim fp32 tensor."""

    @staticmethod
    def forward(ctx, logits2d, elem, labels, weight, scale, up_shift, H, W):
        n = int(elem.numel())
        loss = ops.ce_fwd(logits2d, elem, labels, n, weight, up_shift, H, W)
        out = torch.zeros((1,), device=logits2d.device, dtype=f32)
        ops.sum_f32(loss, out)
        ctx.cfg = (n, scale, up_shift, H, W)
        ctx.save_for_backward(logits2d, elem, labels, weight)
        return (out * scale).view(())

    @staticmethod
    def backward(ctx, dout):
        logits2d, elem, labels, weight = ctx.saved_tensors
        n, scale, up_shift, H, W = ctx.cfg
        dl = torch.zeros_like(logits2d)
        g = _c(dout).to(f32).view(1)
        ops.ce_bwd(logits2d, elem, labels, n, weight, g, scale, up_shift, H, W, dl)
        return dl, None, None, None, None, None, None, None


# ----------------------------------------------------------------------------------------------
# The arithmetic form (amp latch, split / fp32 precision) a Function's FORWARD ran with is the one its BACKWARD runs with:
# both are recorded on ctx and re-established around backward, so an eval / inference forward of this or another model between
# a training forward and its backward (or a second model under a different autocast setting) cannot switch the backward's
# products to another form.
# ----------------------------------------------------------------------------------------------
class _NoGradCtx:
    """stand-in for the autograd context when grad mode is off (inference / validation): nothing is saved, nothing needs a gradient"""
    needs_input_grad = (False,) * 64

    def save_for_backward(self, *tensors):
        pass

    def mark_non_differentiable(self, *tensors):
        pass

    def mark_dirty(self, *tensors):
        pass

    def set_materialize_grads(self, value):
        pass


def _pin_arithmetic(cls):
    fwd, bwd = cls.forward, cls.backward
    graph_apply = cls.apply

    def apply(*args):
        # grad mode off: the forward runs directly -- torch.autograd.Function.apply costs ~15 us of host time per call (72 calls in a
        # single-document inference, which is host-bound) and would record nothing
        if not torch.is_grad_enabled():
            return fwd(_NoGradCtx(), *args)
        return graph_apply(*args)

    cls.apply = staticmethod(apply)

    def forward(ctx, *a, **k):
        ctx._vbg_form = (ops.amp_enabled(), ops.precision())
        return fwd(ctx, *a, **k)

    def backward(ctx, *g):
        amp, prec = ctx._vbg_form
        prev_amp, prev_prec = ops.amp_enabled(), ops.precision()
        ops.set_amp(amp)
        ops.set_precision(prec)
        try:
            return bwd(ctx, *g)
        finally:
            ops.set_amp(prev_amp)
            ops.set_precision(prev_prec)

    cls.forward, cls.backward = staticmethod(forward), staticmethod(backward)


for _name, _obj in list(globals().items()):
    if isinstance(_obj, type) and issubclass(_obj, torch.autograd.Function) and _obj is not torch.autograd.Function:
        _pin_arithmetic(_obj)
