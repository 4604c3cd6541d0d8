"""Per-kernel parity tests: every libvbg entry point (through the C-ABI) vs the CPU oracle / a plain
torch-CPU fp32 statement of the same op.  Integer/index work is bit-exact; fp32 tolerances are
written at each assert.  Needs a real MI355X."""
import math
import random

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import vbg_oracle as O


@pytest.fixture(scope="module")
def ops():
    from vbg import ops as _ops
    return _ops


def dev():
    return torch.device("cuda")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def close(a, b, rtol, atol):
    a = a.detach().cpu()
    b = b.detach().cpu()
    ok = torch.allclose(a, b, rtol=rtol, atol=atol)
    if not ok:
        d = (a - b).abs()
        print("max abs", float(d.max()), "at", int(d.argmax()), "ref", float(b.flatten()[d.argmax()]), "rel-l2",
              float((a - b).norm() / (b.norm() + 1e-30)))
    return ok


# ------------------------------------------------------------------------------------------
# GEMM
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 100), (4128, 768, 768), (77, 5, 512), (1000, 2, 37), (64, 3072, 768)])
def test_gemm_nt(ops, M, N, K):
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3)
    ref = x @ w.t() + b
    y = ops.linear_fwd(x.to(dev()), w.to(dev()), b.to(dev()))
    assert close(y, ref, 1e-4, 1e-4 * math.sqrt(K))
    # asymmetric identity check: A = I  -> C = W^T (catches row/col swaps)
    if M == N == 128:
        eye = torch.eye(128, 64)
        y = ops.linear_fwd(eye.to(dev()), w.to(dev()))
        assert close(y, eye @ w.t(), 0, 1e-6)


@pytest.mark.parametrize("M,N,K,relu", [(128, 1024, 13312, True), (15, 1024, 13312, False), (128, 256, 4100, True), (200, 512, 8192, False)])
def test_linear_slab_split_is_deterministic(ops, M, N, K, relu):
    """round 6: a forward linear layer with a handful of output tiles over a very long reduction (the field-type head's first layer on one
    document) cuts the reduction into slabs that a second launch adds in split order (include/vbg.h vbg_gemm_desc.slab_stride,
    vbg_slab_reduce): against fp64, bit-identical run to run, with and without autograd (no atomics anywhere), and actually taken"""
    from vbg.lib import EPI_NONE, EPI_RELU
    assert ops.slab_split(M, N, K) >= 2 and ops.slab_split(1024, 1024, 13312) == 1 and ops.slab_split(128, 1024, 1024) == 1
    x, w, b = rnd(M, K, seed=11).to(dev()), (rnd(N, K, seed=12) / math.sqrt(K)).to(dev()), rnd(N, seed=13).to(dev())
    ref = x.double() @ w.double().t() + b.double()
    if relu:
        ref = ref.clamp_min(0)
    scale = float((x.double().abs() @ w.double().abs().t()).max())
    log = ops.dispatch_log(True)
    y = ops.linear_fwd(x, w, b, EPI_RELU if relu else EPI_NONE)
    ops.dispatch_log(False)
    assert log.get("gemm:slab_split", 0) == 1
    assert float((y.double() - ref).abs().max()) <= 2e-6 * scale
    with torch.no_grad():
        y2 = ops.linear_fwd(x, w, b, EPI_RELU if relu else EPI_NONE)
    assert torch.equal(y, y2) and torch.equal(y, ops.linear_fwd(x, w, b, EPI_RELU if relu else EPI_NONE))
    was = ops._SLAB_SPLIT[0]
    ops._SLAB_SPLIT[0] = False
    try:
        with torch.no_grad():
            y1 = ops.linear_fwd(x, w, b, EPI_RELU if relu else EPI_NONE)
    finally:
        ops._SLAB_SPLIT[0] = was
    assert float((y1 - y).abs().max()) <= 2e-6 * scale
    # inside an autocast region (one bf16 product per product) the slabs work the same way
    with ops.amp_scope(True), torch.no_grad():
        ya = ops.linear_fwd(x, w, b, EPI_RELU if relu else EPI_NONE)
        assert torch.equal(ya, ops.linear_fwd(x, w, b, EPI_RELU if relu else EPI_NONE))
    assert float((ya.double() - ref).abs().max()) <= 2e-2 * scale and float((ya - y).abs().max()) > 0
    # a split without k-tiles of its own is refused (its slab would stay unwritten)
    with pytest.raises(Exception):
        ops.gemm_raw(64, 64, 256, x, K, 0, w, K, 0, torch.empty(8, 64, 64, device=dev()), 64, splitk=8, slab_stride=64 * 64)


def test_gemm_fp16_pair_form_of_the_forward_kinds(ops):
    """round 6, include/vbg.h vbg_gemm_desc.bf16 = 2: the generic kernels' FORWARD products (dense NT, K-segmented NT with on-the-fly
    upsampling, strided / 1x1 / 3x3 implicit-GEMM convolutions) on two fp16 pieces per operand, three piece products, the split done in
    registers: the bound of the six-product form against fp64, every tile the dispatcher picks, bias / ReLU / GELU epilogues, fused
    BatchNorm statistics; out-of-range operands visible; the backward kinds given the form fall back to six products"""
    from vbg.lib import EPI_GELU_DUAL, EPI_NONE, EPI_RELU, OP_DENSE_K
    d = dev()
    g = torch.Generator().manual_seed(22)
    torch.set_grad_enabled(False)          # (with autograd on, vbg.ops.gemm_raw lets the library split long reductions over atomics: not bit-reproducible)
    request_grad = lambda: torch.set_grad_enabled(True)
    try:
        _fp16_form_body(ops, d, g, EPI_GELU_DUAL, EPI_NONE, EPI_RELU, OP_DENSE_K)
    finally:
        request_grad()


def _fp16_form_body(ops, d, g, EPI_GELU_DUAL, EPI_NONE, EPI_RELU, OP_DENSE_K):
    for (M, N, K) in ((333, 260, 160), (4128, 768, 768), (1024, 1024, 12544), (131072, 256, 64), (40000, 512, 96)):
        x = (torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-6, 6, (M, 1), generator=g).float())).to(d)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(d)
        b = torch.randn(N, generator=g).to(d)
        ref = x.double() @ w.double().t() + b.double()
        scale = float((x.double().abs() @ w.double().abs().t()).max())
        for tile in (0, 64, 128):
            log = ops.dispatch_log(True)
            out = torch.empty(M, N, device=d)
            ops.gemm_raw(M, N, K, x, K, OP_DENSE_K, w, K, OP_DENSE_K, out, N, bias=b, tile=tile, f16=True)
            ops.dispatch_log(False)
            assert log.get("gemm:f16x2", 0) == 1, log
            assert float((out.double() - ref).abs().max()) <= 2e-6 * scale, (M, N, K, tile)
            six = torch.empty(M, N, device=d)
            ops.gemm_raw(M, N, K, x, K, OP_DENSE_K, w, K, OP_DENSE_K, six, N, bias=b, tile=tile)
            assert float((six - out).abs().max()) <= 2e-6 * scale
            # a different arithmetic where the form is really taken: the shapes the six-product form runs on 64 x 64 tiles (on 128 x 128
            # tiles the fp16 form measured slower -- tools/gemm_f16_bench.py -- and the library keeps six products there)
            small = tile == 64 or (tile == 0 and not (-(-M // 128) * -(-N // 128) >= 192 and N >= 128))
            assert torch.equal(six, out) != small, (M, N, K, tile)
        out = torch.empty(M, N, device=d)
        ops.gemm_raw(M, N, K, x, K, OP_DENSE_K, w, K, OP_DENSE_K, out, N, bias=b, epi=EPI_RELU, f16=True)
        assert float((out.double() - ref.clamp_min(0)).abs().max()) <= 2e-6 * scale
        if M <= 4128:
            out2 = torch.empty(M, N, device=d)
            ops.gemm_raw(M, N, K, x, K, OP_DENSE_K, w, K, OP_DENSE_K, out, N, bias=b, epi=EPI_GELU_DUAL, C2=out2, f16=True)
            assert float((out2.double() - F.gelu(ref)).abs().max()) <= 3e-6 * scale
    # out of range: visible, never clipped
    big = torch.full((128, 64), 70000.0, device=d)
    o = torch.empty(128, 64, device=d)
    ops.gemm_raw(128, 64, 64, big, 64, OP_DENSE_K, torch.ones(64, 64, device=d), 64, OP_DENSE_K, o, 64, f16=True)
    assert not bool(torch.isfinite(o).any())
    # convolutions through the implicit-GEMM loader (3x3 stride 2, 1x1 stride 2, 3x3 stride 1 on a shape the row-reuse kernel does not take)
    for (B, H, W, Ci, Co, k, s_, p_) in ((2, 64, 64, 64, 128, 3, 2, 1), (2, 64, 64, 64, 128, 1, 2, 0), (1, 24, 40, 32, 96, 3, 1, 1)):
        xi = torch.randn(B, Ci, H, W, generator=g)
        wi = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
        ref = F.conv2d(xi.double(), wi.double(), None, s_, p_).permute(0, 2, 3, 1)
        xh, wh = xi.permute(0, 2, 3, 1).contiguous().to(d), wi.permute(0, 2, 3, 1).contiguous().to(d)
        was = ops._CONV3[0]
        ops._CONV3[0] = False
        try:
            log = ops.dispatch_log(True)
            y = ops.conv2d_fwd(xh, wh, s_, p_)
            ops.dispatch_log(False)
        finally:
            ops._CONV3[0] = was
        assert log.get("gemm:f16x2", 0) == 1, log
        assert float((y.double().cpu() - ref).abs().max()) <= 3e-6 * float(ref.abs().max()) * (Ci * k * k) ** 0.5
    # the backward kinds do not take the form (their gradient operands are not scaled into fp16's range here)
    dy, wt = torch.randn(512, 256, generator=g).to(d) * 1e-7, torch.randn(256, 128, generator=g).to(d)
    from vbg.lib import OP_DENSE_R
    a, c = torch.empty(512, 128, device=d), torch.empty(512, 128, device=d)
    ops.gemm_raw(512, 128, 256, dy, 256, OP_DENSE_K, wt, 128, OP_DENSE_R, a, 128, f16=True)
    ops.gemm_raw(512, 128, 256, dy, 256, OP_DENSE_K, wt, 128, OP_DENSE_R, c, 128)
    assert torch.equal(a, c)


@pytest.mark.parametrize("tile", [64, 128])
def test_gemm_tiles_epilogues(ops, tile):
    from vbg.lib import EPI_GELU_DUAL, EPI_RELU, OP_DENSE_K
    M, N, K = 333, 257, 129
    x, w, b = rnd(M, K, seed=4), rnd(N, K, seed=5), rnd(N, seed=6)
    xd, wd, bd = x.to(dev()), w.to(dev()), b.to(dev())
    out = torch.empty(M, N, device=dev())
    ops.gemm_raw(M, N, K, xd, K, OP_DENSE_K, wd, K, OP_DENSE_K, out, N, bias=bd, epi=EPI_RELU, tile=tile)
    assert close(out, torch.relu(x @ w.t() + b), 1e-4, 2e-4)
    out2 = torch.empty(M, N, device=dev())
    ops.gemm_raw(M, N, K, xd, K, OP_DENSE_K, wd, K, OP_DENSE_K, out, N, bias=bd, epi=EPI_GELU_DUAL, C2=out2, tile=tile)
    h = x @ w.t() + b
    assert close(out, h, 1e-4, 2e-4) and close(out2, F.gelu(h), 1e-4, 2e-4)
    # accumulate + split-K
    base = rnd(M, N, seed=7)
    acc = base.to(dev()).clone()
    ops.gemm_raw(M, N, K, xd, K, OP_DENSE_K, wd, K, OP_DENSE_K, acc, N, accumulate=True, splitk=3, tile=tile)
    assert close(acc, base + x @ w.t(), 1e-4, 2e-4)


def test_gemm_nn_tn(ops):
    M, N, K = 515, 96, 200
    dy, w, x = rnd(M, N, seed=8), rnd(N, K, seed=9), rnd(M, K, seed=10)
    dx = ops.linear_dgrad(dy.to(dev()), w.to(dev()))
    assert close(dx, dy @ w, 1e-4, 1e-3)
    dw = torch.zeros(N, K, device=dev())
    ops.linear_wgrad(dy.to(dev()), x.to(dev()), dw, accumulate=True)
    assert close(dw, dy.t() @ x, 1e-4, 2e-3)
    # tiny N (classifier) backward: unaligned K=5 reduction
    dy5, w5 = rnd(M, 5, seed=11), rnd(5, 512, seed=12)
    assert close(ops.linear_dgrad(dy5.to(dev()), w5.to(dev())), dy5 @ w5, 1e-4, 1e-4)
    x5 = rnd(M, 512, seed=13)
    dw5 = torch.zeros(5, 512, device=dev())
    ops.linear_wgrad(dy5.to(dev()), x5.to(dev()), dw5)
    assert close(dw5, dy5.t() @ x5, 1e-4, 2e-3)
    assert close(ops.colsum(dy.to(dev())), dy.sum(0), 1e-4, 1e-3)


def test_gemm_relu_scale_prologue(ops):
    from vbg.lib import OP_DENSE_K, OP_DENSE_R
    M, N, K = 130, 64, 70            # P-like matrix with negative (= dropped) entries
    p, v = rnd(M, K, seed=14), rnd(K, N, seed=15)
    pad = 72
    pp = torch.zeros(M, pad)
    pp[:, :K] = p
    out = torch.empty(M, N, device=dev())
    ops.gemm_raw(M, N, K, pp.to(dev()), pad, OP_DENSE_K, v.to(dev()), N, OP_DENSE_R, out, N, a_relu_scale=1.25)
    assert close(out, (torch.relu(p) * 1.25) @ v, 1e-4, 1e-3)
    # transposed use (dV = P^T dO)
    do = rnd(M, N, seed=16)
    out = torch.empty(K, N, device=dev())
    ops.gemm_raw(K, N, M, pp.to(dev()), pad, OP_DENSE_R, do.to(dev()), N, OP_DENSE_R, out, N, a_relu_scale=1.25)
    assert close(out, (torch.relu(p) * 1.25).t() @ do, 1e-4, 1e-3)


def test_gemm_grouped(ops):
    from vbg.lib import OP_DENSE_K
    # 3 sequences x 2 heads of Q K^T with different lengths, packed [ntok, 2*64]
    lens = [5, 130, 64]
    heads, dh = 2, 64
    ntok = sum(lens)
    q, k = rnd(ntok, heads * dh, seed=17), rnd(ntok, heads * dh, seed=18)
    offs, rows = [], 0
    s_off, total = [], 0
    grp = []
    for L in lens:
        ld = (L + 3) // 4 * 4
        for h in range(heads):
            grp += [L, L, dh, rows * heads * dh + h * dh, rows * heads * dh + h * dh, total, 0, 0]
            s_off.append((total, L, ld))
            total += L * ld
        rows += L
    S = torch.zeros(total, device=dev())
    g = torch.tensor(grp, dtype=torch.int64, device=dev())
    # per-group ldc differs -> launch per distinct ld is not needed: ldc is taken from desc, so use max-ld trick:
    # here every group has its own ld, so run groups with equal ld together
    for ldv in sorted(set(x[2] for x in s_off)):
        sel = [i for i, x in enumerate(s_off) if x[2] == ldv]
        gg = torch.tensor([v for i in sel for v in grp[8 * i:8 * i + 8]], dtype=torch.int64, device=dev())
        Lm = max(s_off[i][1] for i in sel)
        ops.gemm_raw(0, 0, 0, q.to(dev()), heads * dh, OP_DENSE_K, k.to(dev()), heads * dh, OP_DENSE_K, S, ldv, grp=gg, ngroups=len(sel), grp_max=(Lm, Lm))
    r = 0
    i = 0
    for L in lens:
        for h in range(heads):
            o, _, ld = s_off[i]
            got = S[o:o + L * ld].view(L, ld)[:, :L]
            ref = q[r:r + L, h * dh:(h + 1) * dh] @ k[r:r + L, h * dh:(h + 1) * dh].t()
            assert close(got, ref, 1e-4, 1e-3), (L, h)
            i += 1
        r += L


def test_gemm_segments(ops):
    from vbg.lib import OP_DENSE_K
    # P_fuse-style: 4 sources at 1/8,1/4,1/2,1/1 resolution, 32 channels each, 1x1 conv to 48
    B, H, W, Cs = 2, 16, 24, 32
    srcs = [rnd(B, H >> s, W >> s, Cs, seed=20 + s) for s in (3, 2, 1, 0)]
    w = rnd(48, 4 * Cs, seed=25)
    ups = [t.permute(0, 3, 1, 2) for t in srcs]
    cat = torch.cat([F.interpolate(u, scale_factor=f, mode="nearest") if f > 1 else u for u, f in zip(ups, (8, 4, 2, 1))], 1)
    ref = F.conv2d(cat, w.view(48, 4 * Cs, 1, 1)).permute(0, 2, 3, 1)
    d = [t.to(dev()) for t in srcs]
    out = torch.empty(B * H * W, 48, device=dev())
    segs = [(d[0], Cs, Cs, 3), (d[1], 2 * Cs, Cs, 2), (d[2], 3 * Cs, Cs, 1), (d[3], 4 * Cs, Cs, 0)]
    ops.gemm_raw(B * H * W, 48, 4 * Cs, d[0], Cs, OP_DENSE_K, w.to(dev()), 4 * Cs, OP_DENSE_K, out, 48, segs=segs, a_hw=(H, W))
    assert close(out.view(B, H, W, 48), ref, 1e-4, 1e-3)


@pytest.mark.parametrize("Cin,Cout,k,stride,pad,H,W", [(16, 32, 3, 1, 1, 9, 11), (64, 128, 3, 2, 1, 16, 20), (64, 128, 1, 2, 0, 16, 20),
                                                        (128, 48, 1, 1, 0, 7, 5), (256, 256, 3, 1, 1, 7, 7), (32, 16, 7, 2, 3, 20, 18),
                                                        (512, 512, 3, 1, 1, 3, 4), (256, 512, 3, 2, 1, 6, 8), (256, 512, 1, 2, 0, 6, 8),
                                                        (256, 256, 3, 1, 1, 6, 8), (128, 256, 3, 2, 1, 12, 16), (512, 256, 1, 1, 0, 3, 4),
                                                        # weight-gradient gather fast paths: a k-tile is a piece of one pixel row ...
                                                        (32, 32, 3, 1, 1, 5, 32), (32, 64, 3, 2, 1, 6, 64), (16, 32, 3, 1, 1, 4, 16), (64, 64, 3, 1, 1, 3, 64),
                                                        # ... or whole rows of one image
                                                        (64, 32, 3, 1, 1, 8, 8), (32, 64, 3, 2, 1, 16, 16), (64, 128, 1, 2, 0, 16, 16), (64, 64, 3, 1, 1, 16, 4)])
def test_conv(ops, Cin, Cout, k, stride, pad, H, W):
    B = 3
    x = rnd(B, Cin, H, W, seed=30).requires_grad_(True)
    w = (rnd(Cout, Cin, k, k, seed=31) / math.sqrt(Cin * k * k)).requires_grad_(True)
    y = F.conv2d(x, w, None, stride, pad)
    gy = rnd(*y.shape, seed=32)
    y.backward(gy)
    xh = x.detach().permute(0, 2, 3, 1).contiguous().to(dev())
    wh = w.detach().permute(0, 2, 3, 1).contiguous().to(dev())
    yh = ops.conv2d_fwd(xh, wh, stride, pad)
    assert close(yh.permute(0, 3, 1, 2), y, 1e-4, 1e-4)
    gyh = gy.permute(0, 2, 3, 1).contiguous().to(dev())
    dx = ops.conv2d_dgrad(gyh, wh, tuple(xh.shape), stride, pad)
    assert close(dx.permute(0, 3, 1, 2), x.grad, 1e-4, 2e-4)
    dw = torch.zeros_like(wh)
    ops.conv2d_wgrad(gyh, xh, dw, stride, pad)
    assert close(dw.permute(0, 3, 1, 2), w.grad, 1e-4, 2e-3)


@pytest.mark.parametrize("B,H,W,Cs,N", [(2, 8, 128, 32, 128), (3, 4, 64, 48, 136), (2, 8, 32, 16, 256), (1, 128, 128, 64, 128),
                                        # 64-pixel tiles (few pixels): two / four image rows per tile
                                        (3, 6, 32, 32, 128), (2, 12, 16, 32, 256), (8, 32, 32, 64, 256),
                                        # rows wider than a tile: the neighbouring pixels are fetched into the halo rows
                                        (2, 5, 256, 32, 128), (1, 3, 512, 16, 132),
                                        # 64-filter tiles (odd multiples of 64 filters)
                                        (1, 128, 128, 64, 64), (2, 8, 128, 32, 64), (2, 4, 128, 16, 192),
                                        # 7 x 7 region maps stored compactly: two images per tile (an odd count leaves half a tile)
                                        (5, 7, 7, 32, 128), (8, 7, 7, 64, 256), (1, 7, 7, 16, 64)])
def test_conv3x3_row_reuse(ops, B, H, W, Cs, N):
    """csrc/conv3.hip (activation rows shared by the three horizontal taps) against fp64 torch conv2d: forward with bias and fused
    BatchNorm statistics, accumulate, the input gradient through the turned filter; and against the generic implicit GEMM (same
    piece products, other tap order: equal to fp32 rounding)."""
    x = rnd(B, Cs, H, W, seed=60).requires_grad_(True)
    w = (rnd(N, Cs, 3, 3, seed=61) / math.sqrt(Cs * 9)).requires_grad_(True)
    b = rnd(N, seed=62)
    y = F.conv2d(x.double(), w.double(), b.double(), 1, 1)
    gy = rnd(*y.shape, seed=63)
    y.backward(gy.double())
    xh = x.detach().permute(0, 2, 3, 1).contiguous().to(dev())
    wh = w.detach().permute(0, 2, 3, 1).contiguous().to(dev())
    bh = b.to(dev())
    stats = torch.zeros(ops.bn_slots() * 2 * N, device=dev(), dtype=torch.float64)
    yh = ops.conv3x3(xh, wh, bh, stats=stats)
    ref = y.float().permute(0, 2, 3, 1)
    tol = 3e-6 * float(ref.abs().max())          # fp32-grade: errors scale with the summed magnitudes, not with the element
    assert close(yh, ref, 2e-5, tol)
    # the two-piece fp16 form of the forward (three piece products): the same bound, also for operands far below / above 1 (the scaled
    # low piece keeps small values exact to 2^-21; the accumulators are fp32 either way)
    assert close(ops.conv3x3(xh, wh, bh, f16x2=True), ref, 2e-5, tol)
    for sc in (2.0 ** -12, 2.0 ** 9):
        ys = ops.conv3x3(xh * sc, wh, None, f16x2=True)
        assert close(ys / sc, ref - b.view(1, 1, 1, N), 2e-5, tol), sc
    st = stats.view(-1, 2, N).sum(0).cpu()
    y2 = y.detach().permute(0, 2, 3, 1).reshape(-1, N)
    assert torch.allclose(st[0], y2.sum(0), rtol=1e-5, atol=1e-4) and torch.allclose(st[1], (y2 * y2).sum(0), rtol=1e-5, atol=1e-4)
    # generic kernel, same arithmetic form
    ops.set_conv3(False)
    try:
        yg = ops.conv2d_fwd(xh, wh, 1, 1, bh)
    finally:
        ops.set_conv3(True)
    assert close(yh, yg, 1e-5, 2 * tol)
    # accumulate
    acc = yh.clone()
    ops.conv3x3(xh, wh, None, out=acc, accumulate=True)
    assert close(acc, 2 * ref - b.view(1, 1, 1, N), 2e-5, 2 * tol)
    # input gradient: the same kernel over dy with the turned filter
    if N % 16 == 0 and Cs % 4 == 0:
        gyh = gy.permute(0, 2, 3, 1).contiguous().to(dev())
        wf = ops.conv3x3_wflip(wh)
        assert torch.equal(wf.cpu(), w.detach().permute(1, 2, 3, 0).flip(1, 2).contiguous())
        dx = ops.conv3x3(gyh, wf)
        assert close(dx.permute(0, 3, 1, 2), x.grad.float(), 2e-5, 3e-6 * float(x.grad.abs().max()))
        # ... in the two-piece fp16 form, dy at the magnitudes gradients really have (1e-9 ... 1e+6 x the test's): the kernel scales dy
        # by the power of two derived from its largest magnitude (vbg_amax) and the result back -- the same bound at every scale
        for sc in (1.0, 2.0 ** -30, 2.0 ** -20, 2.0 ** 20):
            g = gyh * sc
            am = ops.amax(g)
            assert int(am.max().item()) == int((g.abs().max()).view(torch.int32).item())
            dxs = ops.conv3x3(g, wf, f16x2=True, x_amax=am)
            assert close(dxs.permute(0, 3, 1, 2) / sc, x.grad.float(), 2e-5, 3e-6 * float(x.grad.abs().max())), sc
        # without the scale a large operand is visible as inf / nan, never clipped
        assert not bool(torch.isfinite(ops.conv3x3(gyh * 1e6, wf, f16x2=True)).all())


@pytest.mark.parametrize("B,H,W,Cs,N,nz", [(8, 32, 32, 64, 256, 3), (8, 32, 32, 64, 256, 4), (2, 16, 16, 96, 128, 6), (2, 16, 16, 96, 128, 2),
                                           (4, 16, 16, 192, 384, 12), (1, 8, 16, 32, 128, 3), (1, 16, 16, 512, 512, 12)])
def test_conv3x3_split(ops, B, H, W, Cs, N, nz):
    """split form of csrc/conv3.hip (nz workgroups per tile, one filter row and / or channel group each, slabs + arrival ticket, the
    last arriver adds them in block order): against fp64 conv2d with bias and fused BatchNorm statistics, both arithmetic forms, the
    input-gradient use (scaled operand, accumulate), run-to-run bit-identical, the tickets left at zero"""
    x = rnd(B, Cs, H, W, seed=160)
    w = rnd(N, Cs, 3, 3, seed=161) / math.sqrt(Cs * 9)
    b = rnd(N, seed=162)
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1).float().permute(0, 2, 3, 1)
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev())
    wh = w.permute(0, 2, 3, 1).contiguous().to(dev())
    bh = b.to(dev())
    tol = 3e-6 * float(ref.abs().max())
    for f16 in (False, True):
        stats = torch.zeros(ops.bn_slots() * 2 * N, device=dev(), dtype=torch.float64)
        y = ops.conv3x3(xh, wh, bh, stats=stats, f16x2=f16, nsplit=nz)
        assert close(y, ref, 2e-5, tol), f16
        y1 = ops.conv3x3(xh, wh, bh, f16x2=f16, nsplit=1)
        assert close(y, y1, 1e-5, tol)
        assert torch.equal(y, ops.conv3x3(xh, wh, bh, f16x2=f16, nsplit=nz))
        st = stats.view(-1, 2, N).sum(0).cpu()
        y2 = ref.double().reshape(-1, N)
        assert torch.allclose(st[0], y2.sum(0), rtol=1e-5, atol=1e-4) and torch.allclose(st[1], (y2 * y2).sum(0), rtol=1e-5, atol=1e-4)
    # gradient-like operand: scaled through its amax slot, accumulated into an existing tensor
    g = xh * 2.0 ** -24
    base = ((ref - b.view(1, 1, 1, N)) * 2.0 ** -24).contiguous().to(dev())
    acc = base.clone()
    ops.conv3x3(g, wh, None, out=acc, accumulate=True, f16x2=True, x_amax=ops.amax(g), nsplit=nz)
    assert close(acc * 2.0 ** 23, ref - b.view(1, 1, 1, N), 2e-5, 2 * tol)
    assert all(int(t.abs().max().item()) == 0 for t in ops._CONV3_TICKETS.values())
    # the library's own choice for the late trunk stages puts at least 256 workgroups on the chip
    assert ops.conv3_split(8, 32, 32, 256, 256) * 128 >= 256 and ops.conv3_split(8, 16, 16, 512, 512) * 64 >= 256
    assert ops.conv3_ok(8, 32, 32, 256, 256, 3, 3, 1, 1) and ops.conv3_ok(8, 16, 16, 512, 512, 3, 3, 1, 1)
    # ... and a single document's last stage (8 tiles) is split instead of running 32 workgroups of the generic kernel
    assert ops.conv3_split(1, 16, 16, 512, 512) == 4 and ops.conv3_ok(1, 16, 16, 512, 512, 3, 3, 1, 1)


@pytest.mark.parametrize("B,H,W,Cs,N,nz", [(2, 8, 128, 32, 128, 1), (1, 128, 128, 64, 64, 1), (2, 5, 256, 32, 128, 1), (1, 3, 512, 16, 132, 1),
                                           (2, 4, 128, 16, 192, 1), (8, 64, 64, 32, 128, 1),
                                           (5, 7, 7, 32, 128, 1), (8, 7, 7, 64, 256, 1), (1, 7, 7, 16, 64, 1),
                                           (8, 32, 32, 64, 256, 3), (2, 16, 16, 96, 128, 6), (4, 16, 16, 192, 384, 12), (8, 16, 16, 64, 128, 4)])
def test_conv3x3_presplit_filter(ops, B, H, W, Cs, N, nz):
    """PW form of csrc/conv3.hip (round 4): the filter arrives as the fp16-pair plane image conv3_wprep_kernel wrote (k-tile order,
    LDS-DMA) instead of being split by every workgroup.  Same pieces, same products, same order: BIT-IDENTICAL to the in-kernel form, for
    the forward (bias, statistics, accumulate), the input gradient (turned filter written as planes only, scaled dy), whole rows /
    wide rows / 64-filter tiles / region maps / split reductions; and the images follow the weights (in-place update seen by torch's
    version counter, library kernels seen through the weight epoch)."""
    x = rnd(B, Cs, H, W, seed=260)
    w = rnd(N, Cs, 3, 3, seed=261) / math.sqrt(Cs * 9)
    b = rnd(N, seed=262)
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev())
    wd = w.to(dev()).contiguous(memory_format=torch.channels_last)          # the parameter as the model holds it (OIHW, channels_last)
    w4 = wd.permute(0, 2, 3, 1)
    assert w4.is_contiguous() and w4.data_ptr() == wd.data_ptr()
    bh = b.to(dev())
    wp = ops.conv3_planes(wd, w4, False)
    bn = 64 if (N % 128 != 0 and N % 64 == 0) else 128                     # rows per filter tile of the image
    assert wp is not None and wp.numel() == ((N + bn - 1) // bn) * 9 * (Cs // 16) * 64 * bn
    assert nz > 1 or ops.conv3_pw_ok(B, H, W, Cs, N)
    s0 = torch.zeros(ops.bn_slots() * 2 * N, device=dev(), dtype=torch.float64)
    s1 = torch.zeros_like(s0)
    y0 = ops.conv3x3(xh, w4, bh, stats=s0, f16x2=True, nsplit=nz)
    y1 = ops.conv3x3(xh, w4, bh, stats=s1, f16x2=True, nsplit=nz, w_planes=wp)
    assert torch.equal(y0, y1)
    assert torch.allclose(s0.view(-1, 2, N).sum(0), s1.view(-1, 2, N).sum(0), rtol=1e-12, atol=1e-9)
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1).float().permute(0, 2, 3, 1)
    assert close(y1, ref, 2e-5, 3e-6 * float(ref.abs().max()))
    acc0, acc1 = y0.clone(), y0.clone()
    ops.conv3x3(xh, w4, None, out=acc0, accumulate=True, f16x2=True, nsplit=nz)
    ops.conv3x3(xh, w4, None, out=acc1, accumulate=True, f16x2=True, nsplit=nz, w_planes=wp)
    assert torch.equal(acc0, acc1)
    # input gradient: dy [B, H, W, N] with the turned filter; reduction over the N output channels
    def split_ok(nz_, red, nout):
        cs = nz_ // 3 if nz_ % 3 == 0 else nz_
        return nz_ == 1 or (nout % 128 == 0 and (H * W) % 128 == 0 and H != 7 and red % cs == 0 and (red // cs) % 16 == 0)
    if N % 16 == 0 and Cs % 8 == 0:
        nzb = nz if split_ok(nz, N, Cs) else 1
        if nzb > 1 or ops.conv3_pw_ok(B, H, W, N, Cs):
            gy = rnd(B, N, H, W, seed=263).permute(0, 2, 3, 1).contiguous().to(dev()) * 2.0 ** -22
            am = ops.amax(gy)
            wpf = ops.conv3_planes(wd, w4, True)
            assert wpf is not None
            d0 = ops.conv3x3(gy, ops.conv3x3_wflip(w4), f16x2=True, x_amax=am, nsplit=nzb)
            d1 = ops.conv3x3(gy, w4, f16x2=True, x_amax=am, nsplit=nzb, w_planes=wpf, n_out=Cs)
            assert torch.equal(d0, d1)
    # 64-filter tiles by the caller's choice (the late trunk stages: twice the tiles), its own image, any split the shape allows
    if N % 64 == 0 and H != 7 and (H * W) % 128 == 0:
        wp64 = ops.conv3_planes(wd, w4, False, bn=64)
        assert wp64 is not None and wp64.data_ptr() != wp.data_ptr()
        for z in (1, nz):
            if z > 1 and N % 64:
                continue
            y64 = ops.conv3x3(xh, w4, bh, f16x2=True, nsplit=z, w_planes=wp64, bn=64)
            assert close(y64, ref, 2e-5, 3e-6 * float(ref.abs().max())), z
            assert torch.equal(y64, ops.conv3x3(xh, w4, bh, f16x2=True, nsplit=z, w_planes=wp64, bn=64))
        if nz == 1:
            assert torch.equal(ops.conv3x3(xh, w4, bh, f16x2=True, nsplit=1, w_planes=wp64, bn=64), y0)     # (same pieces, products, order)
    # the images follow the weights: torch's version counter ...
    with torch.no_grad():
        wd.mul_(2.0)
    wp2 = ops.conv3_planes(wd, w4, False)
    assert wp2.data_ptr() == wp.data_ptr()                                   # same storage, rewritten
    y2 = ops.conv3x3(xh, w4, None, f16x2=True, nsplit=nz, w_planes=wp2)
    assert torch.equal(y2, ops.conv3x3(xh, w4, None, f16x2=True, nsplit=nz))
    assert close(y2, 2 * (ref - b.view(1, 1, 1, N)), 2e-5, 6e-6 * float(ref.abs().max()))
    # ... and the library's own writes (optimizer kernels) through the weight epoch
    ops.scale_(wd.permute(0, 2, 3, 1).reshape(-1), 0.5)
    stale = ops.conv3x3(xh, w4, None, f16x2=True, nsplit=nz, w_planes=wp2)
    assert torch.equal(stale, y2)                                            # (nobody told the cache yet)
    ops.bump_weight_epoch()
    y3 = ops.conv3x3(xh, w4, None, f16x2=True, nsplit=nz, w_planes=ops.conv3_planes(wd, w4, False))
    assert torch.equal(y3, ops.conv3x3(xh, w4, None, f16x2=True, nsplit=nz))
    assert close(y3, ref - b.view(1, 1, 1, N), 2e-5, 3e-6 * float(ref.abs().max()))


@pytest.mark.parametrize("B,H,W,Cs,Cout", [(2, 8, 32, 32, 128), (3, 4, 16, 64, 64), (1, 16, 64, 96, 256), (2, 5, 48, 64, 192),
                                           # image rows of 128 / 256 pixels (cfg2's 128 x 128 and cfg5's 256 x 256 maps): 8 / 16 k-tiles per row
                                           (2, 6, 128, 64, 128), (1, 5, 256, 32, 128), (2, 3, 256, 64, 256),
                                           # 7 x 7 region maps: a k-tile is two rows of the image's 8 x 8 slot grid
                                           (5, 7, 7, 32, 128), (9, 7, 7, 64, 256)])
@pytest.mark.parametrize("slabs", [True, False])
def test_conv3x3_wgrad(ops, B, H, W, Cs, Cout, slabs):
    """csrc/conv3.hip weight gradient (transposing LDS reads, X loaded once for nine taps) against fp64 autograd; accumulates into dw;
    the slab form is run-to-run bit-identical"""
    x = rnd(B, Cs, H, W, seed=70)
    w = (rnd(Cout, Cs, 3, 3, seed=71) / math.sqrt(Cs * 9)).double().requires_grad_(True)
    gy = rnd(B, Cout, H, W, seed=72)
    F.conv2d(x.double(), w, None, 1, 1).backward(gy.double())
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev())
    gyh = gy.permute(0, 2, 3, 1).contiguous().to(dev())
    init = rnd(Cout, 3, 3, Cs, seed=73)
    dw = init.clone().to(dev())
    ops.conv3x3_wgrad(gyh, xh, dw, slabs=slabs)
    ref = w.grad.permute(0, 2, 3, 1).float() + init
    scale = float(w.grad.abs().max())
    assert float((dw.cpu() - ref).abs().max()) <= 3e-6 * scale + 1e-6
    if slabs:
        dw2 = init.clone().to(dev())
        ops.conv3x3_wgrad(gyh, xh, dw2, slabs=True)
        assert torch.equal(dw, dw2)
    # the two-piece fp16 form: both operands scaled by their largest magnitudes (exact powers of two), three piece products -- the
    # same bound against fp64 at the magnitudes real gradients / activations have, far outside fp16's own range
    for sy, sx in ((1.0, 1.0), (2.0 ** -30, 1.0), (2.0 ** -22, 2.0 ** 7), (2.0 ** 18, 2.0 ** -9)):
        dwf = torch.zeros_like(init).to(dev())
        ops.conv3x3_wgrad(gyh * sy, xh * sx, dwf, slabs=slabs, f16x2=True)
        assert float((dwf.cpu() / (sy * sx) - (ref - init)).abs().max()) <= 3e-6 * scale, (sy, sx)
    # an outlier 2^20 above the rest of dy: the small elements then sit below 2^-17 of the maximum and keep only their high piece
    # exactly -- the error bound is absolute (2^-25 of the scaled range), still inside the fp32-grade bound of the whole sum
    gyo = gyh.clone()
    gyo[0, 0, 0, 0] = 2.0 ** 20
    dwo = torch.zeros_like(init).to(dev())
    ops.conv3x3_wgrad(gyo, xh, dwo, slabs=slabs, f16x2=True)
    dwr = torch.zeros_like(init).to(dev())
    ops.conv3x3_wgrad(gyo, xh, dwr, slabs=slabs)
    assert float((dwo - dwr).abs().max()) <= 3e-6 * float(dwr.abs().max())


def _fp16_scaled(t, slot=None):
    """what the one-product forms multiply: t rounded to fp16 after the power-of-two scaling of its amax slot (exact), scaled back"""
    if slot is None:
        return t.half().double()
    mx = float(slot.view(torch.float32).max().item())
    e = 13 - int(math.floor(math.log2(mx)))
    return (t * 2.0 ** e).half().double() * 2.0 ** -e


@pytest.mark.parametrize("B,H,W,Cs,N,nz,bn", [(2, 8, 128, 32, 128, 1, 0), (1, 128, 128, 64, 64, 1, 0), (8, 64, 64, 32, 128, 1, 0), (8, 7, 7, 64, 256, 1, 0),
                                              (8, 32, 32, 64, 256, 3, 0), (8, 32, 32, 64, 256, 1, 64), (8, 16, 16, 64, 128, 4, 64)])
def test_conv3x3_one_product_form(ops, B, H, W, Cs, N, nz, bn):
    """`amp` form of the pre-split-filter kernels (round 4): ONE product on the hi pieces.  The result equals the fp64 convolution of the
    fp16-ROUNDED operands (what the reference's fp16 autocast multiplies) up to fp32 accumulation -- forward with bias / statistics,
    64-filter tiles, split reductions, region maps, and the input gradient with a dy far below fp16's range (scaled by its amax slot) --
    and it IS a reduced-precision product (differs from the fp32-grade form)."""
    x = rnd(B, Cs, H, W, seed=360)
    w = rnd(N, Cs, 3, 3, seed=361) / math.sqrt(Cs * 9)
    b = rnd(N, seed=362)
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev())
    wd = w.to(dev()).contiguous(memory_format=torch.channels_last)
    w4 = wd.permute(0, 2, 3, 1)
    bh = b.to(dev())
    wp = ops.conv3_planes(wd, w4, False, bn=bn)
    full = ops.conv3x3(xh, w4, bh, f16x2=True, nsplit=nz, w_planes=wp, bn=bn)
    s1 = torch.zeros(ops.bn_slots() * 2 * N, device=dev(), dtype=torch.float64)
    log = ops.dispatch_log(True)
    with ops.amp_scope(True):
        y = ops.conv3x3(xh, w4, bh, stats=s1, f16x2=True, nsplit=nz, w_planes=wp, bn=bn)
    ops.dispatch_log(False)
    assert log.get("conv3:onep", 0) == 1
    ref = F.conv2d(x.half().double(), w.half().double(), b.double(), 1, 1).permute(0, 2, 3, 1)
    assert float((y.cpu().double() - ref).abs().max()) <= 3e-6 * float(ref.abs().max())
    assert torch.allclose(s1.view(-1, 2, N).sum(0)[0].cpu(), y.double().sum((0, 1, 2)).cpu(), rtol=1e-6, atol=1e-6 * B * H * W)
    assert float((y - full).abs().max()) > 1e-5 * float(ref.abs().max())                      # (really one product)
    # input gradient: the turned filter as planes, dy at gradient magnitudes
    if bn == 0 and H != 7 and ops.conv3_pw_ok(B, H, W, N, Cs):
        gy = rnd(B, N, H, W, seed=363).permute(0, 2, 3, 1).contiguous().to(dev()) * 2.0 ** -22
        am = ops.amax(gy)
        wpf = ops.conv3_planes(wd, w4, True)
        with ops.amp_scope(True):
            d = ops.conv3x3(gy, w4, f16x2=True, x_amax=am, nsplit=1, w_planes=wpf, n_out=Cs)
        gref = F.conv_transpose2d(_fp16_scaled(gy.cpu(), am.cpu()).permute(0, 3, 1, 2), w.half().double(), None, 1, 1).permute(0, 2, 3, 1)
        assert float((d.cpu().double() - gref).abs().max()) <= 3e-6 * float(gref.abs().max())


@pytest.mark.parametrize("B,H,W,Cs,Cout", [(2, 16, 16, 32, 128), (1, 32, 64, 64, 64), (2, 6, 128, 64, 128), (9, 7, 7, 64, 256)])
def test_conv3x3_wgrad_one_product_form(ops, B, H, W, Cs, Cout):
    """`amp` form of the weight-gradient kernel: hi x hi only, both operands scaled by their amax slots -- equals fp64 on the fp16-rounded
    scaled operands up to fp32 accumulation, at magnitudes far outside fp16's own range"""
    x = rnd(B, Cs, H, W, seed=370)
    gy = rnd(B, Cout, H, W, seed=372)
    for sy, sx in ((1.0, 1.0), (2.0 ** -26, 2.0 ** 6)):
        xh = (x.permute(0, 2, 3, 1).contiguous() * sx).to(dev())
        gyh = (gy.permute(0, 2, 3, 1).contiguous() * sy).to(dev())
        ay, ax = ops.amax(gyh), ops.amax(xh)
        dw = torch.zeros(Cout, 3, 3, Cs, device=dev())
        log = ops.dispatch_log(True)
        with ops.amp_scope(True):
            ops.conv3x3_wgrad(gyh, xh, dw, f16x2=True, dy_amax=ay, x_amax=ax)
        ops.dispatch_log(False)
        assert log.get("conv3:wgrad_onep", 0) == 1
        xr = _fp16_scaled(xh.cpu(), ax.cpu()).permute(0, 3, 1, 2)
        gr = _fp16_scaled(gyh.cpu(), ay.cpu()).permute(0, 3, 1, 2)
        wz = torch.zeros(Cout, Cs, 3, 3, dtype=torch.float64, requires_grad=True)
        F.conv2d(xr, wz, None, 1, 1).backward(gr)
        ref = wz.grad.permute(0, 2, 3, 1)
        assert float((dw.cpu().double() - ref).abs().max()) <= 3e-6 * float(ref.abs().max()), (sy, sx)
        full = torch.zeros_like(dw)
        ops.conv3x3_wgrad(gyh, xh, full, f16x2=True, dy_amax=ay, x_amax=ax)
        assert float((dw - full).abs().max()) > 1e-5 * float(ref.abs().max())


@pytest.mark.parametrize("tile", [128129, 256128])
def test_plane_gemm_one_product_form(ops, tile):
    """`amp` form of the fp16-pair plane products: only the hi planes are loaded, one product.  NT (bias, scaled gradient operand),
    single and grouped TN (the weight gradients) against fp64 on the fp16-rounded operands."""
    d = dev()
    g = torch.Generator().manual_seed(91)
    M, N, K = 1100, 384, 320
    a, b = torch.randn(M, K, generator=g).to(d), (torch.randn(N, K, generator=g) / K ** 0.5).to(d)
    bias = torch.randn(N, generator=g).to(d)
    pa, pb = ops.split_planes_pair(a), ops.split_planes_pair(b)
    out, full = torch.empty(M, N, device=d), torch.empty(M, N, device=d)
    ops.plane_gemm(pa, pb, full, bias=bias, form=1, tile=tile)
    log = ops.dispatch_log(True)
    with ops.amp_scope(True):
        ops.plane_gemm(pa, pb, out, bias=bias, form=1, tile=tile)
    ops.dispatch_log(False)
    assert log.get("plane_gemm:onep", 0) == 1
    ref = a.half().double() @ b.half().double().t() + bias.double()
    assert float((out.double() - ref).abs().max()) <= 3e-6 * float(ref.abs().max())
    assert float((out - full).abs().max()) > 1e-5 * float(ref.abs().max())
    # a gradient operand: planes scaled by the amax slot
    gq = (torch.randn(M, K, generator=g) * 2.0 ** -24).to(d)
    sl = ops.amax(gq)
    pg = ops.split_planes_pair(gq, amax_slot_=sl)
    with ops.amp_scope(True):
        ops.plane_gemm(pg, pb, out, form=1, tile=tile, a_amax=sl)
    ref = _fp16_scaled(gq, sl).to(d) @ b.half().double().t()
    assert float((out.double() - ref).abs().max()) <= 3e-6 * float(ref.abs().max())
    # weight gradients: dW = g^T x, single and grouped
    x2 = torch.randn(M, 256, generator=g).to(d)
    px = ops.split_planes_pair(x2)
    dw = torch.zeros(K, 256, device=d)
    dw2 = torch.zeros(K, 256, device=d)
    with ops.amp_scope(True):
        ops.plane_gemm(pg, px, dw, trans=True, accumulate=True, form=1, tile=tile, a_amax=sl)
        ops.plane_gemm_grouped([(pg, px, dw2), (pa, px, torch.zeros(K, 256, device=d))], trans=True, accumulate=True, form=1, tile=tile, a_amax=[sl, None])
    refw = _fp16_scaled(gq, sl).to(d).t() @ x2.half().double()
    for got in (dw, dw2):
        assert float((got.double() - refw).abs().max()) <= 3e-6 * float(refw.abs().max())


def test_conv3x3_dispatch(ops):
    """conv2d_fwd / conv2d_dgrad take the row-reuse kernel for a wide trunk shape and agree with the generic path"""
    B, H, W, C = 2, 128, 128, 128
    assert ops.conv3_ok(B, H, W, C, C, 3, 3, 1, 1) and not ops.conv3_ok(B, H, W, C, C, 3, 3, 2, 1) and not ops.conv3_ok(1, 16, 16, C, C, 3, 3, 1, 1)
    xh = rnd(B, H, W, C, seed=64).to(dev())
    wh = (rnd(C, 3, 3, C, seed=65) / 34).to(dev())
    gy = rnd(B, H, W, C, seed=66).to(dev())
    y1, dx1 = ops.conv2d_fwd(xh, wh, 1, 1), ops.conv2d_dgrad(gy, wh, tuple(xh.shape), 1, 1)
    ops.set_conv3(False)
    try:
        y0, dx0 = ops.conv2d_fwd(xh, wh, 1, 1), ops.conv2d_dgrad(gy, wh, tuple(xh.shape), 1, 1)
    finally:
        ops.set_conv3(True)
    assert close(y1, y0, 1e-5, 1e-5) and close(dx1, dx0, 1e-5, 1e-5)


def test_stem_im2col(ops):
    B, H, W = 2, 20, 18
    x = rnd(B, 3, H, W, seed=33)
    w = rnd(64, 3, 7, 7, seed=34) / 12
    ref = F.conv2d(x, w, None, 2, 3)
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev())
    col = ops.im2col(xh, 7, 7, 2, 3, 160)
    wp = torch.zeros(64, 160)
    wp[:, :147] = w.permute(0, 2, 3, 1).reshape(64, 147)
    y = ops.linear_fwd(col, wp.to(dev()))
    assert close(y.view(B, ref.shape[2], ref.shape[3], 64).permute(0, 3, 1, 2), ref, 1e-4, 1e-4)


# ------------------------------------------------------------------------------------------
# BERT row kernels
# ------------------------------------------------------------------------------------------
def test_embed_ln(ops):
    V, Pn, Hd, n = 300, 40, 768, 77
    word, pos, typ = rnd(V, Hd, seed=40).requires_grad_(True), rnd(Pn, Hd, seed=41).requires_grad_(True), rnd(2, Hd, seed=42).requires_grad_(True)
    gam, bet = (1 + 0.1 * rnd(Hd, seed=43)).requires_grad_(True), rnd(Hd, seed=44).requires_grad_(True)
    g = torch.Generator().manual_seed(45)
    ids = torch.randint(0, V, (n,), generator=g)
    pid = torch.randint(0, Pn, (n,), generator=g)
    y = F.layer_norm(word[ids] + typ[0] + pos[pid], (Hd,), gam, bet, 1e-12)
    gy = rnd(n, Hd, seed=46)
    y.backward(gy)
    d = dev()
    out, xhat, rstd = ops.embed_ln_fwd(ids.int().to(d), pid.int().to(d), word.detach().to(d), pos.detach().to(d), typ.detach()[0].contiguous().to(d),
                                       gam.detach().to(d), bet.detach().to(d), 1e-12, 0.0, 1, 0)
    assert close(out, y, 1e-4, 1e-5)
    dword, dpos, dtyp = torch.zeros(V, Hd, device=d), torch.zeros(Pn, Hd, device=d), torch.zeros(Hd, device=d)
    dg, db = torch.zeros(Hd, device=d), torch.zeros(Hd, device=d)
    ops.embed_ln_bwd(gy.to(d), xhat, rstd, ids.int().to(d), pid.int().to(d), gam.detach().to(d), 0.0, 1, 0, dword, dpos, dtyp, dg, db)
    assert close(dword, word.grad, 1e-3, 1e-4) and close(dpos, pos.grad, 1e-3, 1e-4) and close(dtyp, typ.grad[0], 1e-3, 1e-3)
    assert close(dg, gam.grad, 1e-3, 1e-3) and close(db, bet.grad, 1e-3, 1e-3)


@pytest.mark.parametrize("rows,Hd", [(133, 768), (1037, 768), (600, 256)])
def test_dropout_add_ln(ops, rows, Hd):
    """rows >= 512 takes the slot-workspace path of the backward (column sums spread over slot rows, folded, workspace left zero)"""
    x, r = rnd(rows, Hd, seed=50).requires_grad_(True), rnd(rows, Hd, seed=51).requires_grad_(True)
    gam, bet = (1 + 0.1 * rnd(Hd, seed=52)).requires_grad_(True), rnd(Hd, seed=53).requires_grad_(True)
    y = F.layer_norm(x + r, (Hd,), gam, bet, 1e-12)
    gy = rnd(rows, Hd, seed=54)
    y.backward(gy)
    d = dev()
    yo, xhat, rstd = ops.dropout_add_ln_fwd(x.detach().to(d), r.detach().to(d), gam.detach().to(d), bet.detach().to(d), 1e-12, 0.0, 1, 0)
    ypl = ops.planes_empty(rows, Hd, d)
    y2, _, _ = ops.dropout_add_ln_fwd(x.detach().to(d), r.detach().to(d), gam.detach().to(d), bet.detach().to(d), 1e-12, 0.0, 1, 0, out_planes=ypl)
    assert torch.equal(y2, yo) and torch.equal(ypl.buf, ops.split_planes(yo).buf)          # planes written by the LayerNorm pass == split(y)
    assert close(yo, y, 1e-4, 1e-5)
    dg, db = torch.zeros(Hd, device=d), torch.zeros(Hd, device=d)
    slot = ops.amax_slot(d)
    dx, dres = ops.dropout_add_ln_bwd(gy.to(d), xhat, rstd, gam.detach().to(d), 0.0, 1, 0, dg, db, dx_amax=slot)
    assert int(slot.max().item()) == int(dx.abs().max().view(torch.int32).item())          # max |dx| rides on the kernel
    assert close(dx, x.grad, 1e-3, 1e-5) and close(dres, r.grad, 1e-3, 1e-5)
    assert close(dg, gam.grad, 1e-3, 1e-3) and close(db, bet.grad, 1e-3, 1e-3)
    # a second call accumulates into dgamma / dbeta; the partials workspace (rows >= 512) needs no particular content on entry
    for w in ops._LN_WS.values():
        w.fill_(float("nan"))
    ops.dropout_add_ln_bwd(gy.to(d), xhat, rstd, gam.detach().to(d), 0.0, 1, 0, dg, db)
    assert close(dg, 2 * gam.grad, 1e-3, 2e-3) and close(db, 2 * bet.grad, 1e-3, 2e-3)
    # dropout: same mask forward and backward, keep rate ~ 1-p, kept values scaled by 1/(1-p)
    p = 0.1
    one = torch.ones(rows, Hd, device=d)
    zero = torch.zeros(rows, Hd, device=d)
    gam1, bet0 = torch.ones(Hd, device=d), torch.zeros(Hd, device=d)
    yo, xhat, rstd = ops.dropout_add_ln_fwd(one, zero, gam1, bet0, 1e-12, p, 123, 7)
    # dropout(1)+0 is {0, 1/(1-p)} -> after LN two distinct values per row; recover the mask from the sign
    keep = (yo > 0).float()
    rate = float(keep.mean())
    assert abs(rate - (1 - p)) < 0.01
    dx, dres = ops.dropout_add_ln_bwd(torch.ones_like(yo), xhat, rstd, gam1, p, 123, 7, dg, db)
    assert torch.equal((dx != 0), (dres != 0) & (keep > 0)) or float(((dx != 0).float() - keep).abs().mean()) < 1e-3
    # planes form of the backward: dx as bf16 planes (== split(dx) bit for bit), its column sums into the bias gradient, same dres
    dg2, db2, dbias = torch.zeros(Hd, device=d), torch.zeros(Hd, device=d), torch.full((Hd,), 3.0, device=d)
    gy2 = rnd(rows, Hd, seed=55).to(d)
    dx_ref, dres_ref = ops.dropout_add_ln_bwd(gy2, xhat, rstd, gam1, p, 123, 7, torch.zeros(Hd, device=d), torch.zeros(Hd, device=d))
    pdx, dres2 = ops.dropout_add_ln_bwd_planes(gy2, xhat, rstd, gam1, p, 123, 7, dg2, db2, dbias)
    assert torch.equal(dres2, dres_ref) and torch.equal(pdx.buf, ops.split_planes(dx_ref).buf)
    assert torch.allclose(dbias.double() - 3.0, dx_ref.double().sum(0), rtol=1e-4, atol=1e-4 * rows ** 0.5)
    # the partials workspace needs no initialisation, and its rows are added in a fixed order: a second run over a workspace full of NaN
    # gives the same bits (the slot scheme of rounds 2-4 used float atomics and a workspace that had to be left zero)
    for w in ops._LN_WS.values():
        w.fill_(float("nan"))
    dg3, db3, dbias3 = torch.zeros(Hd, device=d), torch.zeros(Hd, device=d), torch.full((Hd,), 3.0, device=d)
    ops.dropout_add_ln_bwd_planes(gy2, xhat, rstd, gam1, p, 123, 7, dg3, db3, dbias3)
    assert torch.equal(dg3, dg2) and torch.equal(db3, db2) and torch.equal(dbias3, dbias)


def test_softmax(ops):
    heads = 2
    lens = [7, 130, 512]
    d = dev()
    off, ldp, total = [], [], 0
    for L in lens:
        ld = (L + 3) // 4 * 4
        ldp.append(ld)
        for h in range(heads):
            off.append(total)
            total += L * ld
    s = rnd(total, seed=60) * 3
    sd = s.to(d).clone()
    offd = torch.tensor(off, dtype=torch.int64, device=d)
    lend = torch.tensor(lens, dtype=torch.int32, device=d)
    ldd = torch.tensor(ldp, dtype=torch.int32, device=d)
    ng = len(off)
    ops.softmax_fwd(sd, offd, lend, ldd, ng, heads, max(lens), 0.125, 0.0, 1, 0)
    dp = rnd(total, seed=61)
    dpd = dp.to(d).clone()
    ops.softmax_bwd(sd, dpd, offd, lend, ldd, ng, heads, max(lens), 0.125, 0.0)
    i = 0
    for si, L in enumerate(lens):
        for h in range(heads):
            ld = ldp[si]
            blk = s[off[i]:off[i] + L * ld].view(L, ld)[:, :L].clone().requires_grad_(True)
            ref = torch.softmax(blk * 0.125, -1)
            got = sd[off[i]:off[i] + L * ld].view(L, ld)
            assert close(got[:, :L], ref, 1e-4, 1e-6)
            assert float(got[:, L:].abs().sum()) == 0.0
            g = dp[off[i]:off[i] + L * ld].view(L, ld)[:, :L]
            ref.backward(g)
            assert close(dpd[off[i]:off[i] + L * ld].view(L, ld)[:, :L], blk.grad, 1e-3, 1e-6)
            i += 1
    # dropout: sign encodes the mask; |P| unchanged; keep rate ~ 0.9
    sd2 = s.to(d).clone()
    ops.softmax_fwd(sd2, offd, lend, ldd, ng, heads, max(lens), 0.125, 0.1, 9, 3)
    assert close(sd2.abs(), sd.abs(), 0, 0)
    L, ld = 512, 512
    blk = sd2[off[4]:off[4] + L * ld]
    rate = float((blk > 0).float().mean())
    assert abs(rate - 0.9) < 0.01


def test_gelu_erf_epilogues(ops):
    """the library's branch-free erf (vbg_common.h vbg_erff: both pieces of the single-precision erf evaluated, one selected; exp through
    v_exp_f32) seen through the two GELU epilogues that use it: gelu(h) and gelu'(h) against fp64 over a dense grid of [-6, 6] plus
    normal samples -- absolute error <= 2.5e-7 max(1, |h|) (fp32 rounding of the result itself is 6e-8 |gelu|), exact zeros at 0"""
    from vbg.lib import EPI_GELU_DUAL
    d = dev()
    h = torch.cat([torch.linspace(-6, 6, 1 << 20), torch.randn(1 << 20) * 1.5, torch.zeros(64)]).view(-1, 64).contiguous()
    hd = h.double()
    cdf = 0.5 * (1 + torch.special.erf(hd / 2 ** 0.5))
    ref_d = cdf + hd * torch.exp(-0.5 * hd * hd) / (2 * math.pi) ** 0.5
    ones = torch.ones_like(h).to(d)
    got_d = ops.gelu_bwd_(h.to(d), ones.clone()).cpu().double()                    # 1 * gelu'(h)
    assert float(((got_d - ref_d).abs() / hd.abs().clamp_min(1.0)).max()) <= 2.5e-7
    # gelu(h) through the GELU-dual epilogue of a product with the identity: h = I h
    n = 64
    x = h[:4096].to(d)
    eye = torch.eye(n, device=d)
    out, gl = ops.linear_fwd(x, eye, None, EPI_GELU_DUAL)
    assert float((out - x).abs().max()) <= 1e-6
    od = out.cpu().double()
    err = (gl.cpu().double() - od * 0.5 * (1 + torch.special.erf(od / 2 ** 0.5))).abs() / od.abs().clamp_min(1.0)
    assert float(err.max()) <= 2.5e-7
    z = torch.zeros(128, n, device=d)
    assert float(ops.linear_fwd(z, eye, None, EPI_GELU_DUAL)[1].abs().max()) == 0.0


def test_gelu_relu_bwd(ops):
    n = 4099
    h, g = rnd(n, seed=62) * 2, rnd(n, seed=63)
    hh = h.clone().requires_grad_(True)
    F.gelu(hh).backward(g)
    got = ops.gelu_bwd_(h.to(dev()), g.to(dev()).clone())
    assert close(got, hh.grad, 1e-4, 1e-6)
    y = torch.relu(h)
    got = ops.relu_bwd_(y.to(dev()), g.to(dev()).clone())
    assert close(got, g * (y > 0), 0, 0)


# ------------------------------------------------------------------------------------------
# BERTgrid (bit-exact parts)
# ------------------------------------------------------------------------------------------
def _pack_boxes(boxes):
    off = [0]
    for b in boxes:
        off.append(off[-1] + b.shape[0])
    allb = torch.cat([b.int() for b in boxes], 0) if off[-1] else torch.zeros((0, 4), dtype=torch.int32)
    doc = torch.cat([torch.full((b.shape[0],), i, dtype=torch.int32) for i, b in enumerate(boxes)]) if off[-1] else torch.zeros((0,), dtype=torch.int32)
    return allb.contiguous(), torch.tensor(off, dtype=torch.int32), doc


def test_seg_reduce_bitexact(ops, golden):
    g = golden("aggregate.npz")
    for mode, mi in (("mean", 0), ("first", 1)):
        tok, mask = torch.from_numpy(g[f"{mode}_tok"]), torch.from_numpy(g[f"{mode}_mask"])
        B, T, Hd = tok.shape
        tok2d = tok.reshape(B * T, Hd)
        rows, starts, lens = [], [], []
        base = 0
        for b in range(B):
            r = torch.nonzero(mask[b] == 1).flatten() + b * T
            st, ln = O.seg_runs(torch.from_numpy(g[f"{mode}_seg{b}"]))
            starts += list(st + base)
            lens += list(ln)
            base += r.numel()
            rows.append(r)
        d = dev()
        rows = torch.cat(rows).int().to(d)
        out = ops.seg_reduce_fwd(tok2d.to(d), rows, torch.tensor(starts, dtype=torch.int32, device=d), torch.tensor(lens, dtype=torch.int32, device=d), mi)
        ref = np.concatenate([g[f"{mode}_out0"], g[f"{mode}_out1"]], 0)
        assert np.array_equal(out.cpu().numpy(), ref)        # bit exact
        # backward
        gy = rnd(out.shape[0], Hd, seed=70)
        dt = torch.zeros(B * T, Hd, device=d)
        ops.seg_reduce_bwd(gy.to(d), rows, torch.tensor(starts, dtype=torch.int32, device=d), torch.tensor(lens, dtype=torch.int32, device=d), mi, dt)
        tk = tok2d.clone().requires_grad_(True)
        embs = [O.seg_aggregate(tk.view(B, T, Hd)[b], mask[b], torch.from_numpy(g[f"{mode}_seg{b}"]), mode) for b in range(B)]
        torch.cat(embs).backward(gy)
        assert close(dt, tk.grad, 1e-6, 1e-7)


def test_owner_scatter_bitexact(ops, golden):
    g = golden("scatter.npz")
    H, W = int(g["H"]), int(g["W"])
    boxes = [torch.from_numpy(g[f"box{b}"]) for b in range(3)]
    embs = [torch.from_numpy(g[f"emb{b}"]) for b in range(3)]
    allb, off, doc = _pack_boxes(boxes)
    d = dev()
    own = ops.owner_map(allb.to(d), off.to(d), 3, H // 8, W // 8, 8)
    base = 0
    for b in range(3):
        ref = O.owner_map(boxes[b].numpy(), H // 8, W // 8, 8)
        ref = np.where(ref >= 0, ref + base, -1)
        assert np.array_equal(own[b].cpu().numpy(), ref)
        base += boxes[b].shape[0]
    # C=6 is not a multiple of 4 -> NCHW path (reference layout), bit exact vs the reference's grid
    emb = torch.cat(embs, 0).contiguous()
    grid = ops.grid_scatter_fwd(emb.to(d), own, 6, layout=1)
    assert np.array_equal(grid.cpu().numpy(), g["grid"])
    # NHWC path with C=8
    emb8 = torch.cat([emb, emb[:, :2]], 1).contiguous()
    g8 = ops.grid_scatter_fwd(emb8.to(d), own, 8, layout=0)
    assert np.array_equal(g8.cpu().numpy()[..., :6], np.transpose(g["grid"], (0, 2, 3, 1)))
    # backward (CopySlices semantics)
    gout = torch.from_numpy(g["gout"]).permute(0, 2, 3, 1).contiguous()
    demb = torch.zeros(emb.shape[0], 6, device=d)
    ops.grid_scatter_bwd(gout.to(d), own, allb.to(d), doc.to(d), 8, demb)
    ref = np.concatenate([g[f"gemb{b}"] for b in range(3)], 0)
    assert close(demb, torch.from_numpy(ref), 1e-5, 1e-6)


def test_label_raster_bitexact(ops, golden):
    g = golden("labels.npz")
    coors = [torch.from_numpy(g[f"coor{b}"]) for b in range(2)]
    classes = torch.cat([torch.from_numpy(g[f"class{b}"]) for b in range(2)]).int()
    allb, off, _ = _pack_boxes(coors)
    d = dev()
    own = ops.owner_map(allb.to(d), off.to(d), 2, 32, 64, 1)
    pn, cl = ops.label_raster(own, classes.to(d))
    assert np.array_equal(pn.cpu().numpy(), g["pos_neg"]) and np.array_equal(cl.cpu().numpy(), g["cls"])


def test_owner_map_random_large(ops):
    # cfg2-like: 128 overlapping boxes on 512x512, stride 8 and stride 1, incl. border-crossing ones
    g = torch.Generator().manual_seed(5)
    boxes = []
    for _ in range(3):
        x1 = torch.randint(-10, 500, (200,), generator=g)
        y1 = torch.randint(-10, 500, (200,), generator=g)
        w = torch.randint(0, 73, (200,), generator=g)
        h = torch.randint(0, 25, (200,), generator=g)
        boxes.append(torch.stack([x1, y1, x1 + w, y1 + h], 1).int())
    allb, off, _ = _pack_boxes(boxes)
    d = dev()
    for stride, n in ((8, 64), (1, 512)):
        own = ops.owner_map(allb.to(d), off.to(d), 3, n, n, stride).cpu().numpy()
        base = 0
        for b in range(3):
            ref = O.owner_map(boxes[b].numpy(), n, n, stride)
            assert np.array_equal(own[b], np.where(ref >= 0, ref + base, -1))
            base += 200


# ------------------------------------------------------------------------------------------
# conv helpers
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(3, 64, 9, 7), (4, 64, 64, 48), (2, 512, 5, 3), (2, 16, 33, 31), (1, 256, 40, 24)])
@pytest.mark.parametrize("relu,res", [(True, False), (True, True), (False, False)])
def test_batchnorm(ops, relu, res, shape):
    B, C_, H, W = shape
    x = rnd(B, C_, H, W, seed=80).requires_grad_(True)
    r = rnd(B, C_, H, W, seed=81).requires_grad_(True)
    gam, bet = (1 + 0.1 * rnd(C_, seed=82)).requires_grad_(True), rnd(C_, seed=83).requires_grad_(True)
    rm, rv = torch.zeros(C_), torch.ones(C_)
    y = F.batch_norm(x, rm, rv, gam, bet, True, 0.1, 1e-5)
    if res:
        y = y + r
    if relu:
        y = torch.relu(y)
    gy = rnd(B, C_, H, W, seed=84)
    y.backward(gy)
    d = dev()
    x2 = x.detach().permute(0, 2, 3, 1).reshape(-1, C_).contiguous().to(d)
    r2 = r.detach().permute(0, 2, 3, 1).reshape(-1, C_).contiguous().to(d) if res else None
    stats = ops.bn_stats(x2)
    rmd, rvd = torch.zeros(C_, device=d), torch.ones(C_, device=d)
    mean, invstd = ops.bn_finalize(stats, C_, ops.bn_slots(), x2.shape[0], 1e-5, 0.1, rmd, rvd)
    assert float(stats.abs().max()) == 0.0                       # the slot workspace is cleared behind the read
    folded = ops.bn_fold(ops.bn_stats(x2), C_)
    assert float(stats.abs().max()) == 0.0
    assert close(folded[:C_].float() / x2.shape[0], x.detach().mean((0, 2, 3)), 1e-5, 1e-6)
    m1, i1 = ops.bn_finalize(folded, C_, 1, x2.shape[0], 1e-5, 0.1, None, None)          # the SyncBN route: folded sums, one slot
    assert torch.equal(m1, mean) and torch.equal(i1, invstd)
    assert close(rmd, rm, 1e-5, 1e-6) and close(rvd, rv, 1e-5, 1e-6)
    yo = ops.bn_apply(x2, r2, mean, invstd, gam.detach().to(d), bet.detach().to(d), relu)
    assert close(yo, y.permute(0, 2, 3, 1).reshape(-1, C_), 1e-4, 1e-5)
    g2 = gy.permute(0, 2, 3, 1).reshape(-1, C_).contiguous().to(d)
    slots = ops.bn_bwd_reduce(g2, yo, x2, mean, invstd, relu)
    dg, db = torch.zeros(C_, device=d), torch.zeros(C_, device=d)
    sums = ops.bn_param_grad(slots, C_, dg, db)
    slot = ops.amax_slot(d)
    dx, dres = ops.bn_bwd_apply(g2, yo, x2, mean, invstd, gam.detach().to(d), sums, x2.shape[0], relu, res, None, None, dx_amax=slot)
    assert close(dx, x.grad.permute(0, 2, 3, 1).reshape(-1, C_), 1e-3, 1e-5)
    assert int(slot.max().item()) == int(dx.abs().max().view(torch.int32).item())       # the bit pattern of max |dx| rides on the kernel
    assert close(dg, gam.grad, 1e-3, 1e-4) and close(db, bet.grad, 1e-3, 1e-4)
    if res:
        assert close(dres, r.grad.permute(0, 2, 3, 1).reshape(-1, C_), 1e-5, 1e-6)


@pytest.mark.parametrize("relu,res,shape", [(True, True, (2, 64, 33, 20)), (False, False, (3, 128, 16, 16)), (True, False, (1, 256, 7, 9)),
                                            (True, True, (8, 64, 128, 128))])
def test_batchnorm_folding_apply_kernels(ops, relu, res, shape):
    """vbg_bn_apply_fold / vbg_bn_bwd_apply_fold (round 5: the finalize and the affine-gradient fold in the apply kernels' prologues, one
    launch instead of two) against the separate entry points on the same slot rows: the fold runs in fold_slots' order in both, so every
    output -- y, mean, invstd, the running statistics, dx, dres, dgamma, dbeta, the published maxima -- is BIT-equal"""
    B, C_, H, W = shape
    d = dev()
    M = B * H * W
    x2, r2 = rnd(M, C_, seed=180).to(d), (rnd(M, C_, seed=181).to(d) if res else None)
    gam, bet = (1 + 0.1 * rnd(C_, seed=182)).to(d), rnd(C_, seed=183).to(d)
    g2 = (rnd(M, C_, seed=184) * 1e-3).to(d)
    # separate launches (persistent workspace, cleared behind the reads)
    rm_a, rv_a = torch.zeros(C_, device=d), torch.ones(C_, device=d)
    mean_a, inv_a = ops.bn_finalize(ops.bn_stats(x2), C_, ops.bn_slots(), M, 1e-5, 0.1, rm_a, rv_a)
    sa = ops.amax_slot(d)
    y_a = ops.bn_apply(x2, r2, mean_a, inv_a, gam, bet, relu, y_amax=sa)
    dg_a, db_a = torch.zeros(C_, device=d), torch.zeros(C_, device=d)
    sums = ops.bn_param_grad(ops.bn_bwd_reduce(g2, y_a, x2, mean_a, inv_a, relu), C_, dg_a, db_a)
    da = ops.amax_slot(d)
    dx_a, dres_a = ops.bn_bwd_apply(g2, y_a, x2, mean_a, inv_a, gam, sums, M, relu, res, None, None, dx_amax=da)
    # folding launches (zeroed rows from the pool, not cleared)
    rm_b, rv_b = torch.zeros(C_, device=d), torch.ones(C_, device=d)
    slots = ops.bn_stats(x2, ops.bn_zero_slots(d, C_))
    sb = ops.amax_slot(d)
    y_b, mean_b, inv_b = ops.bn_apply_fold(x2, r2, slots, M, 1e-5, 0.1, rm_b, rv_b, gam, bet, relu, y_amax=sb)
    dg_b, db_b = torch.zeros(C_, device=d), torch.zeros(C_, device=d)
    bslots = ops.bn_bwd_reduce(g2, y_b, x2, mean_b, inv_b, relu, ops.bn_zero_slots(d, C_))
    dbb = ops.amax_slot(d)
    dx_b, dres_b = ops.bn_bwd_apply_fold(g2, y_b, x2, mean_b, inv_b, gam, bslots, M, relu, res, dg_b, db_b, dx_amax=dbb)
    # (the statistics themselves are fp64 atomics into slot rows: the two reductions above may differ in the last bit of a double, which
    #  the float conversion of mean / invstd almost always hides -- compared at 1 ulp of fp32, everything downstream exactly when they agree)
    assert close(mean_b, mean_a, 2e-7, 1e-9) and close(inv_b, inv_a, 2e-7, 0)
    if torch.equal(mean_b, mean_a) and torch.equal(inv_b, inv_a):
        assert torch.equal(y_b, y_a) and torch.equal(rm_b, rm_a) and torch.equal(rv_b, rv_a)
        assert int(sb.max().item()) == int(sa.max().item()) == int(y_a.abs().max().view(torch.int32).item())
    else:
        assert close(y_b, y_a, 1e-6, 1e-6)
    assert close(dx_b, dx_a, 1e-5, 1e-9) and close(dg_b, dg_a, 1e-6, 1e-9) and close(db_b, db_a, 1e-6, 1e-9)
    assert int(dbb.max().item()) == int(dx_b.abs().max().view(torch.int32).item())
    if res:
        assert torch.equal(dres_b, dres_a)
    # against torch's own batch_norm as well (training statistics, fp32)
    xt = x2.cpu().view(B, H, W, C_).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    yt = F.batch_norm(xt, torch.zeros(C_), torch.ones(C_), gam.cpu(), bet.cpu(), True, 0.1, 1e-5)
    if res:
        yt = yt + r2.cpu().view(B, H, W, C_).permute(0, 3, 1, 2)
    if relu:
        yt = torch.relu(yt)
    yt.backward(g2.cpu().view(B, H, W, C_).permute(0, 3, 1, 2))
    assert close(y_b, yt.detach().permute(0, 2, 3, 1).reshape(-1, C_), 1e-4, 1e-5)
    assert close(dx_b, xt.grad.permute(0, 2, 3, 1).reshape(-1, C_), 1e-3, 1e-8)


def test_pool_resample_layout(ops):
    d = dev()
    x = rnd(2, 16, 13, 10, seed=90).requires_grad_(True)
    y = F.max_pool2d(x, 3, 2, 1)
    gy = rnd(*y.shape, seed=91)
    y.backward(gy)
    xh = x.detach().permute(0, 2, 3, 1).contiguous().to(d)
    yo, am = ops.maxpool_fwd(xh)
    assert close(yo.permute(0, 3, 1, 2), y, 0, 0)
    dx = ops.maxpool_bwd(gy.permute(0, 2, 3, 1).contiguous().to(d), am, 13, 10)
    assert close(dx.permute(0, 3, 1, 2), x.grad, 1e-6, 1e-6)
    lo, sk = rnd(2, 4, 6, 8, seed=92), rnd(2, 8, 12, 8, seed=93)
    ref = F.interpolate(lo.permute(0, 3, 1, 2), scale_factor=2, mode="nearest").permute(0, 2, 3, 1) + sk
    assert close(ops.upsample2_add(lo.to(d), sk.to(d)), ref, 0, 0)
    hi = rnd(2, 8, 16, 8, seed=94)
    for f in (2, 4, 8):
        ref = F.avg_pool2d(hi.permute(0, 3, 1, 2), f).permute(0, 2, 3, 1) * (f * f)
        assert close(ops.sumpool(hi.to(d), f), ref, 1e-5, 1e-5)
    t = rnd(2, 5, 7, 9, seed=95)
    assert close(ops.nchw_to_nhwc(t.to(d)), t.permute(0, 2, 3, 1), 0, 0)
    assert close(ops.nhwc_to_nchw(t.permute(0, 2, 3, 1).contiguous().to(d)), t, 0, 0)
    lowr = rnd(2, 3, 4, 5, seed=96)
    ref = F.interpolate(lowr.permute(0, 3, 1, 2), scale_factor=4, mode="nearest")
    assert close(ops.upsample_nhwc_to_nchw(lowr.to(d), 4), ref, 0, 0)


def test_transform(ops, golden):
    g = golden("transform.npz")
    d = dev()
    cfg = O.NetCfg(image_min_size=(48, 64), image_max_size=80, test_image_min_size=56)
    sizes = g["eval_sizes"]
    Hh, Ww = g["eval_batch"].shape[-2:]
    batch = torch.zeros(3, Hh, Ww, 3, device=d)
    for i in range(3):
        img = torch.from_numpy(g[f"img{i}"])
        oh, ow = int(sizes[i][0]), int(sizes[i][1])
        ops.normalize_resize(img.to(d), oh, ow, cfg.image_mean, cfg.image_std, batch, i)
        h, w = img.shape[-2:]
        c = ops.rescale_boxes(torch.from_numpy(g[f"coor{i}"]).long().to(d), oh / h, ow / w)
        assert np.array_equal(c.cpu().numpy(), g[f"eval_coor{i}"])          # bit exact (swapped ratios)
    assert close(batch.permute(0, 3, 1, 2), torch.from_numpy(g["eval_batch"]), 1e-4, 1e-4)
    # identity resize = exact normalisation
    img = torch.from_numpy(g["img2"])
    b1 = torch.zeros(1, 64, 64, 3, device=d)
    ops.normalize_resize(img.to(d), 64, 64, cfg.image_mean, cfg.image_std, b1, 0)
    assert close(b1.permute(0, 3, 1, 2), torch.from_numpy(g["ident_batch"]), 1e-6, 1e-6)



def test_transform_module_vs_reference_golden(golden):
    """a1 through the drop-in class: eval size, the TRAIN-mode size draw from torch's global CPU RNG, zero padding to multiples of
    32, the (ImageList, boxes) return convention and bit-exact boxes against the reference's outputs"""
    from pipeline.transform import GeneralizedViBERTgridTransform, ImageList
    g = golden("transform.npz")
    d = dev()
    tr = GeneralizedViBERTgridTransform([0.9248, 0.9224, 0.9215], [0.1532, 0.1545, 0.1536], [48, 64], 56, 80)
    imgs = tuple(torch.from_numpy(g[f"img{i}"]).to(d) for i in range(3))
    coors = tuple(torch.from_numpy(g[f"coor{i}"]).to(d) for i in range(3))
    for mode, seed in (("eval", None), ("train", 123)):
        getattr(tr, mode)()
        if seed is not None:
            torch.manual_seed(seed)
        il, oc = tr(imgs, coors)
        assert isinstance(il, ImageList)
        assert np.array_equal(np.array(il.image_sizes), g[mode + "_sizes"])
        assert close(il.tensors, torch.from_numpy(g[mode + "_batch"]), 1e-4, 1e-4)
        for i in range(3):
            assert oc[i].dtype == torch.int32 and np.array_equal(oc[i].cpu().numpy(), g[f"{mode}_coor{i}"])


# ------------------------------------------------------------------------------------------
# conv + batch-stat BN (+res, +relu) block, forward and backward, against an fp64 CPU reference:
# the rigorous check behind the looser end-to-end gradient tolerances (tests/test_gpu_model.py)
# ------------------------------------------------------------------------------------------
def _block_ref(B, Cin, Cout, H, W, stride, k, relu, res, dt):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    gam = 1 + 0.1 * torch.randn(Cout, generator=g)
    bet = 0.05 * torch.randn(Cout, generator=g)
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    r = torch.randn(B, Cout, Ho, Wo, generator=g) if res else None
    gy = torch.randn(B, Cout, Ho, Wo, generator=g)
    xr, wr, gr, br = (t.to(dt).requires_grad_(True) for t in (x, w, gam, bet))
    y = F.batch_norm(F.conv2d(xr, wr, None, stride, k // 2), None, None, gr, br, True, 0.1, 1e-5)
    if res:
        y = y + r.to(dt)
    if relu:
        y = torch.relu(y)
    y.backward(gy.to(dt))
    return (y.detach(), xr.grad, wr.grad, gr.grad, br.grad), (x, w, gam, bet, r, gy)


@pytest.mark.parametrize("cfgk", [(2, 256, 512, 6, 8, 2, 3, True, False), (2, 256, 512, 6, 8, 2, 1, False, False),
                                  (2, 512, 512, 3, 4, 1, 3, True, True), (4, 64, 64, 32, 32, 1, 3, True, True)])
def test_conv_bn_block_vs_fp64(cfgk):
    from vbg import functions as Fn
    B, Cin, Cout, H, W, stride, k, relu, res = cfgk
    ref64, (x, w, gam, bet, r, gy) = _block_ref(*cfgk, torch.float64)
    d = dev()
    xh = x.permute(0, 2, 3, 1).contiguous().to(d).requires_grad_(True)
    wd = w.to(d).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gd, bd = gam.to(d).requires_grad_(True), bet.to(d).requires_grad_(True)
    rm, rv = torch.zeros(Cout, device=d), torch.ones(Cout, device=d)
    rh = None if r is None else r.permute(0, 2, 3, 1).contiguous().to(d)
    y = Fn.ConvBnFn.apply(xh, wd, gd, bd, rm, rv, rh, stride, k // 2, relu, True, 0.1, 1e-5, False)
    y.backward(gy.permute(0, 2, 3, 1).contiguous().to(d))
    got = (y.detach().permute(0, 3, 1, 2).cpu(), xh.grad.permute(0, 3, 1, 2).cpu(), wd.grad.cpu(), gd.grad.cpu(), bd.grad.cpu())
    for a, b in zip(got, ref64):
        rel = float((a.double() - b).norm() / (b.norm() + 1e-30))
        assert rel < 5e-6, rel          # fp32 rounding level


# ------------------------------------------------------------------------------------------
# RoIAlign
# ------------------------------------------------------------------------------------------
def test_roi_align(ops):
    B, C_, H, W = 2, 32, 24, 32
    feat = rnd(B, C_, H, W, seed=100).requires_grad_(True)
    boxes = [torch.tensor([[0, 0, 128, 96], [3, 5, 40, 22], [100, 80, 140, 120], [10, 10, 11, 11], [60, 40, 61, 90]], dtype=torch.int32),
             torch.tensor([[5, 7, 77, 30], [120, 90, 128, 96], [0, 50, 30, 52]], dtype=torch.int32)]
    ref = O.roi_align(feat, [b.float() for b in boxes], 7, 0.25)
    gy = rnd(*ref.shape, seed=101)
    ref.backward(gy)
    allb, off, doc = _pack_boxes(boxes)
    d = dev()
    fh = feat.detach().permute(0, 2, 3, 1).contiguous().to(d)
    y = ops.roi_align_fwd(fh, allb.to(d), doc.to(d), 7, 0.25)
    assert close(y.permute(0, 3, 1, 2), ref, 1e-4, 1e-5)
    df = torch.zeros_like(fh)
    ops.roi_align_bwd(gy.permute(0, 2, 3, 1).contiguous().to(d), tuple(fh.shape), allb.to(d), doc.to(d), 7, 0.25, df)
    assert close(df.permute(0, 3, 1, 2), feat.grad, 1e-4, 1e-5)


@pytest.mark.parametrize("C_", [256, 40])
def test_roi_align_cfg2_like_boxes(ops, C_):
    """RoIAlign forward against the CPU oracle on cfg2-like boxes (8-72 x 8-24 pixels) plus clipped, degenerate, one-pixel and
    completely-outside ones and a whole-map RoI (42 samples per bin), at the model's channel count and at a ragged one.  (A RoI of
    6 x 7 samples x 4 taps sums 168 fp32 terms in another order than the oracle: atol 3e-5.)"""
    B, H, W = 2, 40, 48
    g = torch.Generator().manual_seed(321)
    feat = torch.randn(B, C_, H, W, generator=g)
    boxes = []
    for b in range(B):
        n = 37
        x1 = torch.randint(0, 4 * W - 72, (n,), generator=g); y1 = torch.randint(0, 4 * H - 24, (n,), generator=g)
        w = torch.randint(8, 73, (n,), generator=g); h = torch.randint(8, 25, (n,), generator=g)
        bx = torch.stack([x1, y1, x1 + w, y1 + h], 1).int()
        extra = torch.tensor([[0, 0, 4 * W, 4 * H], [4 * W - 6, 4 * H - 5, 4 * W + 40, 4 * H + 30], [10, 10, 10, 10], [0, 0, 3, 2],
                              [4 * W + 50, 4 * H + 50, 4 * W + 90, 4 * H + 70], [17, 3, 18, 150]], dtype=torch.int32)
        boxes.append(torch.cat([bx, extra]))
    ref = O.roi_align(feat, [b.float() for b in boxes], 7, 0.25)
    allb, off, doc = _pack_boxes(boxes)
    d = dev()
    fh = feat.permute(0, 2, 3, 1).contiguous().to(d)
    y = ops.roi_align_fwd(fh, allb.to(d), doc.to(d), 7, 0.25)
    err = (y.permute(0, 3, 1, 2).cpu() - ref).abs()
    print("RoIAlign vs oracle: max abs error", float(err.max()), "max |ref|", float(ref.abs().max()))
    assert close(y.permute(0, 3, 1, 2), ref, 1e-4, 3e-5)


# ------------------------------------------------------------------------------------------
# losses / selection primitives
# ------------------------------------------------------------------------------------------
def test_ce_and_selection(ops):
    d = dev()
    n, ncls = 1000, 5
    x = rnd(n, ncls, seed=110).requires_grad_(True)
    g = torch.Generator().manual_seed(111)
    t = torch.randint(0, ncls, (n,), generator=g)
    w = torch.tensor([0.5, 1.0, 2.0, 1.5, 1.0])
    ref = F.cross_entropy(x, t, weight=w, reduction="none")
    got = ops.ce_fwd(x.detach().to(d), None, t.int().to(d), n, w.to(d))
    assert close(got, ref, 1e-5, 1e-6)
    sel = torch.tensor([5, 17, 17, 900, 3], dtype=torch.int32)
    got = ops.ce_fwd(x.detach().to(d), sel.to(d), t.int().to(d), 5, None)
    assert close(got, F.cross_entropy(x, t, reduction="none")[sel.long()], 1e-5, 1e-6)
    (F.cross_entropy(x, t, weight=w, reduction="none")[sel.long()].sum() * 0.3).backward()
    dl = torch.zeros(n, ncls, device=d)
    ops.ce_bwd(x.detach().to(d), sel.to(d), t.int().to(d), 5, w.to(d), None, 0.3, 0, 0, 0, dl)
    assert close(dl, x.grad, 1e-4, 1e-6)
    # upsampled-label mode: logits at 1/4 resolution
    B, Hh, Ww = 2, 16, 24
    lo = rnd(B * (Hh // 4) * (Ww // 4), 3, seed=112)
    lab = torch.randint(0, 3, (B * Hh * Ww,), generator=g)
    full = F.interpolate(lo.view(B, Hh // 4, Ww // 4, 3).permute(0, 3, 1, 2), scale_factor=4, mode="nearest")
    ref = F.cross_entropy(full, lab.view(B, Hh, Ww), reduction="none").flatten()
    got = ops.ce_fwd(lo.to(d), None, lab.int().to(d), B * Hh * Ww, None, 2, Hh, Ww)
    assert close(got, ref, 1e-5, 1e-6)
    # compaction keeps order
    idx, cnt = ops.compact(lab.int().to(d), 0, True)
    k = int(cnt.item())
    assert np.array_equal(idx[:k].cpu().numpy(), torch.nonzero(lab == 0).flatten().numpy())
    idx, cnt = ops.compact(lab.int().to(d), 0, False)
    k = int(cnt.item())
    assert np.array_equal(idx[:k].cpu().numpy(), torch.nonzero(lab != 0).flatten().numpy())
    # stable descending sort with ties
    keys = torch.round(rnd(5000, seed=113) * 4) / 4
    ko, io = ops.sort_desc(keys.to(d))
    rs, ri = torch.sort(keys, descending=True, stable=True)
    assert np.array_equal(ko.cpu().numpy(), rs.numpy()) and np.array_equal(io.cpu().numpy(), ri.numpy())
    assert close(ops.gather_f32(keys.to(d), io), rs, 0, 0)
    out = torch.zeros(1, device=d)
    ops.sum_f32(keys.to(d), out)
    assert abs(float(out) - float(keys.double().sum())) < 1e-2


# ------------------------------------------------------------------------------------------
# optimizers
# ------------------------------------------------------------------------------------------
def test_optimizers(ops):
    d = dev()
    n = 10007
    p0, g1, g2 = rnd(n, seed=120), rnd(n, seed=121), rnd(n, seed=122)
    p = p0.clone().requires_grad_(True)
    opt = torch.optim.SGD([p], lr=0.005, momentum=0.9, weight_decay=0.005)
    pd, md = p0.to(d).clone(), torch.zeros(n, device=d)
    for i, g in enumerate((g1, g2)):
        p.grad = g.clone()
        opt.step()
        ops.sgd_step(pd, g.to(d), md, 0.005, 0.9, 0.005, i == 0)
    assert close(pd, p, 1e-6, 1e-7)
    p = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([p], lr=5e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    pd, md, vd = p0.to(d).clone(), torch.zeros(n, device=d), torch.zeros(n, device=d)
    for i, g in enumerate((g1, g2)):
        p.grad = g.clone()
        opt.step()
        ops.adamw_step(pd, g.to(d), md, vd, 5e-5, 0.9, 0.999, 1e-8, 0.01, i + 1)
    assert close(pd, p, 1e-6, 1e-7)


def _bf16_exact(t):
    return t.bfloat16().float()


@pytest.mark.parametrize("tile", [0, 64064, 128128])
def test_amp_products_match_fp32_on_bf16_values(ops, tile):
    """amp = the same products on the bf16 matrix cores: with operands that are exactly representable in bf16 the rounding inside
    the kernel is the identity, every partial product is exact in fp32, and the result may differ from the fp32 form only by the
    summation order -- checked for all six operand-kind pairs (linear / conv x fwd, dgrad, wgrad), ragged M/N/K included"""
    d = dev()
    ops._FORCE[0] = tile
    try:
        # linear: fwd (K-contig x K-contig), dgrad (K-contig x row-contig), wgrad (row-contig x row-contig), K tail 200 % 32 = 8
        for M, N, K in [(300, 132, 200), (1100, 256, 512), (64, 64, 32)]:
            x, w, dy = _bf16_exact(rnd(M, K, seed=1).to(d)), _bf16_exact(rnd(N, K, seed=2).to(d)), _bf16_exact(rnd(M, N, seed=3).to(d))
            b = rnd(N, seed=4).to(d)
            with torch.no_grad():
                ref = (ops.linear_fwd(x, w, b), ops.linear_dgrad(dy, w), ops.linear_wgrad(dy, x, torch.zeros(N, K, device=d), accumulate=False))
                with ops.amp_scope(True):
                    got = (ops.linear_fwd(x, w, b), ops.linear_dgrad(dy, w), ops.linear_wgrad(dy, x, torch.zeros(N, K, device=d), accumulate=False))
            for r, g in zip(ref, got):
                assert torch.allclose(r, g, rtol=1e-5, atol=1e-5 * float(r.abs().max())), (M, N, K, float((r - g).abs().max()))
        # convolutions (NHWC, OHWI): 3x3 s1, 3x3 s2, 1x1, 7x7-like taps on 64 channels
        for (B, H, W, Ci, Co, k, st, pd) in [(2, 20, 24, 64, 96, 3, 1, 1), (2, 32, 32, 64, 128, 3, 2, 1), (3, 16, 16, 128, 64, 1, 1, 0), (1, 64, 64, 32, 256, 3, 1, 1)]:
            x = _bf16_exact(rnd(B, H, W, Ci, seed=5).to(d))
            w = _bf16_exact(rnd(Co, k, k, Ci, seed=6).to(d))
            Ho, Wo = ops.conv_out_hw(H, W, k, st, pd)
            dy = _bf16_exact(rnd(B, Ho, Wo, Co, seed=7).to(d))
            with torch.no_grad():
                ref = (ops.conv2d_fwd(x, w, st, pd), ops.conv2d_dgrad(dy, w, x.shape, st, pd), ops.conv2d_wgrad(dy, x, torch.zeros_like(w), st, pd, accumulate=False))
                with ops.amp_scope(True):
                    got = (ops.conv2d_fwd(x, w, st, pd), ops.conv2d_dgrad(dy, w, x.shape, st, pd), ops.conv2d_wgrad(dy, x, torch.zeros_like(w), st, pd, accumulate=False))
            for r, g in zip(ref, got):
                assert torch.allclose(r, g, rtol=1e-5, atol=1e-5 * float(r.abs().max())), ((B, H, W, Ci, Co, k, st), float((r - g).abs().max()))
    finally:
        ops._FORCE[0] = 0


def test_split_form_is_fp32_grade(ops):
    """the default arithmetic form (exact 3-way bf16 split, six piece products, vbg_gemm_desc.bf16 = 3) against fp64, next to the
    fp32 matrix pipe on the same operands: errors of the same size (both are fp32 roundings of exact products summed in fp32) for
    every operand-kind pair, operands with a wide dynamic range included"""
    d = dev()
    g = torch.Generator().manual_seed(5)
    scale = torch.exp2(torch.randint(-12, 12, (1, 640), generator=g).float())          # column scales 2^-12 .. 2^11
    x, w, dy = (torch.randn(900, 640, generator=g) * scale).to(d), (torch.randn(300, 640, generator=g) / scale).to(d), torch.randn(900, 300, generator=g).to(d)
    refs = (x.double() @ w.double().t(), dy.double() @ w.double(), dy.double().t() @ x.double())

    def run():
        with torch.no_grad():
            return (ops.linear_fwd(x, w), ops.linear_dgrad(dy, w), ops.linear_wgrad(dy, x, torch.zeros(300, 640, device=d), accumulate=False))

    def errs(outs):
        # relative to the size of the terms that were summed (|a| @ |b|), the natural unit of a dot product's rounding error
        mags = (x.double().abs() @ w.double().abs().t(), dy.double().abs() @ w.double().abs(), dy.double().abs().t() @ x.double().abs())
        return [float(((o.double() - r).abs() / m).max()) for o, r, m in zip(outs, refs, mags)]

    assert ops.precision() == "split"
    e_split = errs(run())
    ops.set_precision("fp32")
    try:
        e_fp32 = errs(run())
    finally:
        ops.set_precision("split")
    for a, b in zip(e_split, e_fp32):
        assert a <= 2e-6 and b <= 2e-6 and a <= 3 * b + 1e-8, (e_split, e_fp32)
    # convolution kinds
    xi, wi = rnd(2, 24, 24, 64, seed=21).to(d), rnd(96, 3, 3, 64, seed=22).to(d)
    dyi = rnd(2, 24, 24, 96, seed=23).to(d)
    xn, wn = xi.permute(0, 3, 1, 2).double().requires_grad_(True), wi.permute(0, 3, 1, 2).double().requires_grad_(True)
    y = torch.nn.functional.conv2d(xn, wn, padding=1)
    y.backward(dyi.permute(0, 3, 1, 2).double())
    with torch.no_grad():
        got = (ops.conv2d_fwd(xi, wi, 1, 1), ops.conv2d_dgrad(dyi, wi, xi.shape, 1, 1), ops.conv2d_wgrad(dyi, xi, torch.zeros_like(wi), 1, 1, accumulate=False))
    for o, r in zip(got, (y.detach().permute(0, 2, 3, 1), xn.grad.permute(0, 2, 3, 1), wn.grad.permute(0, 2, 3, 1))):
        assert float((o.double() - r).abs().max()) <= 2e-6 * float(r.abs().max())


def test_amp_rounds_operands_to_bf16(ops):
    """general fp32 operands: the amp product equals the fp64 product of the bf16-rounded (nearest-even) operands"""
    d = dev()
    x, w = rnd(500, 320, seed=11).to(d), rnd(192, 320, seed=12).to(d)
    with torch.no_grad(), ops.amp_scope(True):
        y = ops.linear_fwd(x, w)
    ref = (x.bfloat16().double() @ w.bfloat16().double().t())
    assert torch.allclose(y.double(), ref, rtol=1e-5, atol=1e-5 * float(ref.abs().max()))
    exact = x.double() @ w.double().t()
    assert float((y.double() - exact).abs().max()) > 1e-4 * float(exact.abs().max())          # (it IS a reduced-precision product)


@pytest.mark.parametrize("M,N,K,tile", [(300, 64, 96, 0), (4100, 256, 64, 0), (777, 132, 200, 0), (5000, 256, 288, 128128), (130, 16, 147, 0)])
def test_gemm_fused_bn_statistics(ops, M, N, K, tile):
    """the GEMM epilogue adds the per-column sum / sum of squares of its output into the BatchNorm slot rows (what vbg_bn_stats
    computes in a separate pass): output unchanged, folded statistics equal the fp64 sums of the stored values"""
    from vbg.lib import OP_DENSE_K
    d = dev()
    x, w = rnd(M, K, seed=700 + N).to(d), rnd(N, K, seed=701 + N).to(d)
    y0, y1 = torch.empty(M, N, device=d), torch.empty(M, N, device=d)
    with torch.no_grad():
        ops.gemm_raw(M, N, K, x, K, OP_DENSE_K, w, K, OP_DENSE_K, y0, N, tile=tile)
        ws = ops._bn_workspace(d, N)
        assert float(ws.abs().max()) == 0.0
        ops.gemm_raw(M, N, K, x, K, OP_DENSE_K, w, K, OP_DENSE_K, y1, N, tile=tile, stats=ws)
    assert torch.equal(y0, y1)
    folded = ops.bn_fold(ws, N)
    ref = torch.cat([y0.double().sum(0), (y0.double() ** 2).sum(0)])
    assert torch.allclose(folded, ref, rtol=1e-6, atol=1e-6 * M)
    assert float(ws.abs().max()) == 0.0


@pytest.mark.parametrize("M,N,ld", [(300, 768, 768), (4128, 3072, 3072), (97, 64, 64), (513, 16, 16), (50, 13, 13), (1000, 2304, 2304),
                                    (700, 768, 2304), (33, 1000, 1000), (5, 4, 4)])
def test_colsum(ops, M, N, ld):
    """bias gradients: vector path (float4, N % 4 == 0, aligned rows), strided views (a Q/K/V slice of dqkv) and the scalar path"""
    x = rnd(M, ld, seed=400 + N)
    xd = x.to(dev())
    v = xd[:, :N] if ld == N else xd[:, N:2 * N]
    ref = (x[:, :N] if ld == N else x[:, N:2 * N]).double().sum(0)
    out = ops.colsum(v)
    assert close(out, ref.float(), 1e-5, 1e-4 * math.sqrt(M))
    acc = torch.ones(N, device=dev())
    ops.colsum(v, out=acc, accumulate=True)
    assert close(acc, (ref + 1).float(), 1e-5, 1e-4 * math.sqrt(M))


@pytest.mark.parametrize("M,N", [(131072, 3), (131072, 5), (200000, 12), (70000, 4), (50000, 40)])
def test_colsum_cancelling_columns(ops, M, N):
    """the bias gradient of a 1x1 segmentation classifier (model/semantic_segmentation_head.py:66-78): per-pixel terms whose partial sums
    run to ~1e3 x the final value (page regions of one label).  The fp64 column sums reproduce the exactly rounded result; the fp32
    atomics of vbg_colsum do not (that is what failed the cfg4e / cfg5e bias gradients at 1.3e-3 / 1.8e-3)."""
    g = torch.Generator().manual_seed(900 + N)
    x = torch.rand(M, N, generator=g) * 1e-5
    x[M // 2:] -= x[: M - M // 2].flip(0) * (1.0 - 1e-3)          # second half cancels the first to 1e-3
    ref = x.double().sum(0)
    out = ops.colsum(x.to(dev()))
    rel = float(((out.cpu().double() - ref).abs() / ref.abs()).max())
    assert rel < 2e-7, rel
    acc = torch.full((N,), 0.5, device=dev())
    ops.colsum(x.to(dev()), out=acc, accumulate=True)
    assert torch.equal(acc.cpu(), (ref + 0.5).float())


# ------------------------------------------------------------------------------------------
# classifier_mode full / crf: row subsets, BCE losses, linear-chain CRF
# ------------------------------------------------------------------------------------------
def test_gather_scatter_rows(ops):
    from vbg import functions as Fn
    d = dev()
    x = rnd(37, 24, seed=200).to(d).requires_grad_(True)
    idx = torch.tensor([5, 0, 36, 17, 3], dtype=torch.int32, device=d)
    y = Fn.GatherRowsFn.apply(x, idx)
    assert torch.equal(y, x.detach()[idx.long()])
    gy = rnd(5, 24, seed=201).to(d)
    y.backward(gy)
    ref = torch.zeros(37, 24, device=d)
    ref[idx.long()] = gy
    assert torch.equal(x.grad, ref)
    assert Fn.GatherRowsFn.apply(x, idx[:0]).shape == (0, 24)


def test_bce_losses_vs_reference_golden(golden):
    """a13: BCELossRandomSample / BCELossOHEM on the HIP path against the reference's values and gradients"""
    from pipeline.custom_loss import BCELossOHEM, BCELossRandomSample
    g = golden("losses_bce.npz")
    d = dev()
    for tag, seed, sl in (("rs", 3, [32, 48]), ("rs2", 4, [16, 16])):
        x = torch.from_numpy(g[tag + "_x"]).to(d).requires_grad_(True)
        random.seed(seed)
        l = BCELossRandomSample(sample_list=sl)(x, torch.from_numpy(g[tag + "_t"]).to(d))
        assert l.dtype == torch.float64 and l.shape == (1,)
        assert abs(float(l) - float(g[tag + "_loss"])) <= 2e-6 * abs(float(g[tag + "_loss"]))
        l.backward()
        assert torch.allclose(x.grad.cpu(), torch.from_numpy(g[tag + "_grad"]), rtol=1e-4, atol=1e-7)
    for tag, seed, kp, kn, rnd_ in (("oh", None, 32, 32, False), ("ohr", 5, 16, 16, True), ("few", None, 16, 4, False)):
        x = torch.from_numpy(g[tag + "_x"]).to(d).requires_grad_(True)
        if seed is not None:
            random.seed(seed)
        l = BCELossOHEM(num_hard_positive=kp, num_hard_negative=kn, random=rnd_)(x, torch.from_numpy(g[tag + "_t"]).to(d))
        assert l.dim() == 0 and abs(float(l) - float(g[tag + "_loss"])) <= 2e-6 * abs(float(g[tag + "_loss"]))
        l.backward()
        assert torch.allclose(x.grad.cpu(), torch.from_numpy(g[tag + "_grad"]), rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("ntag,lens", [(7, [8, 7]), (14, [1, 30, 0, 5]), (3, [4]), (64, [6, 9])])
def test_crf_vs_oracle(ops, ntag, lens):
    """a12 crf: forward-algorithm NLL (+ gradients w.r.t. emissions and transitions) and Viterbi against the oracle
    (itself checked against brute force and the reference run, tests/test_oracle_golden.py)"""
    from vbg import functions as Fn
    g = torch.Generator().manual_seed(300 + ntag)
    N = sum(lens)
    em = torch.randn(N, ntag, generator=g)
    trans = torch.randn(ntag, ntag, generator=g) * 2
    start, stop = ntag - 2, ntag - 1
    trans[start, :] = -10000
    trans[:, stop] = -10000
    tags = torch.randint(0, ntag - 2, (N,), generator=g)
    off = [0]
    for n in lens:
        off.append(off[-1] + n)
    d = dev()
    emd, trd = em.to(d).requires_grad_(True), trans.to(d).requires_grad_(True)
    doc_off = torch.tensor(off, dtype=torch.int32, device=d)
    nll = Fn.CrfNllFn.apply(emd, trd, tags.int().to(d), doc_off, start, stop)
    w = torch.randn(len(lens), generator=g)
    (nll * w.to(d)).sum().backward()
    emr, trr = em.clone().requires_grad_(True), trans.clone().requires_grad_(True)
    ref = []
    for k, n in enumerate(lens):
        f, t = emr[off[k]:off[k + 1]], tags[off[k]:off[k + 1]]
        ref.append((O.crf_forward_alg(f, trr, start, stop) - O.crf_score(f, t, trr, start, stop)) / n if n else torch.zeros(()))
    ref = torch.stack(ref)
    (ref * w).sum().backward()
    assert torch.allclose(nll.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-5)
    assert torch.allclose(emd.grad.cpu(), emr.grad, rtol=1e-4, atol=1e-6)
    assert torch.allclose(trd.grad.cpu(), trr.grad, rtol=1e-4, atol=2e-6)
    path, score = ops.crf_viterbi(em.to(d), doc_off, trans.to(d), start, stop)
    for k, n in enumerate(lens):
        if n == 0:
            continue
        sc, p = O.crf_viterbi(em[off[k]:off[k + 1]], trans, start, stop)
        assert path[off[k]:off[k + 1]].cpu().tolist() == p
        assert abs(float(score[k]) - float(sc)) <= 1e-4 * max(1.0, abs(float(sc)))


# ----------------------------------------------------------------------------------------------
# plane GEMM (csrc/gemm_planes.hip): split kernels, NT / TN / grouped products, epilogues
# ----------------------------------------------------------------------------------------------
def _bf16_planes_ref(x):
    """exact three-way truncation split in torch: (hi, mid, lo) as int16 bit patterns"""
    def top16(t):
        return (t.view(torch.int32) & -65536).view(torch.float32)
    h = top16(x)
    r1 = x - h
    m = top16(r1)
    r2 = r1 - m
    bits = lambda t: (t.view(torch.int32) >> 16).to(torch.int16)
    return bits(h), bits(m), bits(r2), h, m, r2


def test_split_planes_exact_and_transposed():
    from vbg import ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(3)
    for rows, cols in ((5, 7), (64, 96), (130, 770), (1000, 264), (4128, 2304)):
        x = (torch.randn(rows, cols, generator=g) * torch.exp2(torch.randint(-20, 20, (rows, cols), generator=g).float())).to(dev)
        h, m, l, hf, mf, lf = _bf16_planes_ref(x)
        assert torch.equal(hf + mf + lf, x)                                   # the three pieces are exact: x = hi + mid + lo
        p = ops.split_planes(x)
        assert p.ld % 32 == 0 and p.ld >= cols and tuple(p.buf.shape) == (3, rows, p.ld)
        for q, ref in enumerate((h, m, l)):
            assert torch.equal(p.buf[q, :, :cols], ref)
            assert int(p.buf[q, :, cols:].abs().max() if p.ld > cols else 0) == 0     # zero padding of the reduction tail
        pt = ops.split_planes_t(x)
        assert tuple(pt.buf.shape) == (3, cols, (rows + 31) // 32 * 32)
        for q, ref in enumerate((h, m, l)):
            assert torch.equal(pt.buf[q, :, :rows], ref.t())
        pr = ops.split_planes(x, relu=True)
        assert torch.equal(pr.buf[0, :, :cols], _bf16_planes_ref(x.clamp(min=0))[0])
        # column sums riding on the split pass (bias gradients): same planes, sums within fp32 summation noise
        xs = torch.randn(rows, cols, generator=g).to(dev)
        cs = torch.full((cols,), 2.0, device=dev)
        pc = ops.split_planes(xs, colsum_out=cs)
        assert torch.equal(pc.buf, ops.split_planes(xs).buf)
        assert torch.allclose(cs.double() - 2.0, xs.double().sum(0), rtol=1e-5, atol=1e-5 * rows ** 0.5)


@pytest.mark.parametrize("tile", [0, 64064, 128064, 128128, 128129, 128130, 256128])
def test_plane_gemm_nt_vs_fp64(tile):
    from vbg import ops
    from vbg.lib import EPI_GELU_DUAL, EPI_RELU
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(tile + 1)
    for (M, N, K) in ((300, 264, 96), (129, 128, 32), (1000, 772, 800), (257, 512, 3072)):
        a = (torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-8, 8, (M, 1), generator=g).float())).to(dev)
        b = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        ref = a.double() @ b.double().t() + bias.double()
        scale = float((a.double().abs() @ b.double().abs().t()).max())
        pa, pb = ops.split_planes(a), ops.split_planes(b)
        out = torch.full((M, N), 7.0, device=dev)
        ops.plane_gemm(pa, pb, out, bias=bias, tile=tile)
        assert float((out.double() - ref).abs().max()) <= 2e-6 * scale, (M, N, K)
        # ReLU epilogue, GELU dual + planes of the stored value, accumulate, split-K with atomics
        o2 = torch.empty(M, N, device=dev)
        ops.plane_gemm(pa, pb, o2, bias=bias, epi=EPI_RELU, tile=tile)
        assert float((o2.double() - ref.clamp(min=0)).abs().max()) <= 2e-6 * scale
        h, gl, pg = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev), ops.planes_empty(M, N, dev)
        ops.plane_gemm(pa, pb, h, bias=bias, epi=EPI_GELU_DUAL, C2=gl, out_planes=pg, tile=tile)
        assert torch.equal(h, out)
        assert float((gl.double() - torch.nn.functional.gelu(out.double())).abs().max()) <= 1e-5 * max(1.0, float(out.abs().max()))
        assert torch.equal(pg.buf[:, :, :N], ops.split_planes(gl).buf[:, :, :N])
        # column sums of the stored values ride in the epilogue (a bias gradient): planes only, no fp32 output
        if tile != 256256:
            cs = torch.full((N,), 2.0, device=dev)
            pl = ops.planes_empty(M, N, dev)
            ops.plane_gemm(pa, pb, None, bias=bias, out_planes=pl, colsum_out=cs, tile=tile)
            assert torch.equal(pl.buf[:, :, :N], ops.split_planes(out).buf[:, :, :N])
            assert torch.allclose(cs.double() - 2.0, out.double().sum(0), rtol=1e-5, atol=2e-6 * scale * M ** 0.5)
        # GELU backward in the epilogue: stored value = product * gelu'(h), h an input in C2 (== vbg_gelu_bwd on the plain product)
        from vbg.lib import EPI_MUL_GELU_GRAD
        hh = (torch.randn(M, N, generator=g) * 2).to(dev)
        dh = torch.empty(M, N, device=dev)
        ops.plane_gemm(pa, pb, dh, bias=bias, epi=EPI_MUL_GELU_GRAD, C2=hh, tile=tile)
        want = out.clone()
        ops.gelu_bwd_(hh, want)
        assert torch.equal(dh, want)
        acc = torch.ones(M, N, device=dev)
        ops.plane_gemm(pa, pb, acc, accumulate=True, tile=tile)
        assert float((acc.double() - 1 - (ref - bias.double())).abs().max()) <= 2e-6 * scale
        acc = torch.ones(M, N, device=dev)
        ops.plane_gemm(pa, pb, acc, accumulate=True, splitk=3, tile=tile)
        assert float((acc.double() - 1 - (ref - bias.double())).abs().max()) <= 2e-6 * scale
        # same pieces, same products as the in-kernel split form of vbg_gemm: agreement to summation order
        old = ops.linear_fwd(a, b, bias)
        assert float((old - out).abs().max()) <= 1e-6 * scale


def test_plane_gemm_pair_forms_on_the_small_tile():
    """round 6: FORM 1 / FORM 2 on the 4-wave 64 x 64 NT tile (forward products of single documents).  The same piece products in the same
    k order as the 8-wave tiles: bit-identical results, incl. bias, the GELU-dual epilogue and both plane outputs."""
    from vbg import ops
    from vbg.lib import EPI_GELU_DUAL
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(64064)
    for (M, N, K) in ((512, 768, 768), (512, 2304, 768), (130, 3072, 768), (512, 768, 3072), (49, 264, 96)):
        a = torch.randn(M, K, generator=g).to(dev)
        b = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        qa, qb = ops.split_planes_pair(a), ops.split_planes_pair(b)
        for amp in (False, True):
            with ops.amp_scope(amp):
                big, small = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
                ops.plane_gemm(qa, qb, big, bias=bias, tile=128129, form=1)
                log = ops.dispatch_log(True)
                ops.plane_gemm(qa, qb, small, bias=bias, tile=64064, form=1)
                ops.dispatch_log(False)
                assert log.get("plane_gemm:onep" if amp else "plane_gemm:pair", 0) == 1 and log.get("plane_gemm:tile64064", 0) == 1
                assert torch.equal(big, small), (M, N, K, amp)
                deep = torch.empty(M, N, device=dev)
                ops.plane_gemm(qa, qb, deep, bias=bias, tile=64004, form=1)          # (four LDS stages: the same products in the same order)
                assert torch.equal(big, deep), (M, N, K, amp)
                outs = []
                for tile in (128129, 64064, 64004):
                    h = torch.empty(M, N, device=dev)
                    pg, pq = ops.planes_empty(M, N, dev), ops.pair_empty(M, N, dev)
                    ops.plane_gemm(qa, qb, h, bias=bias, epi=EPI_GELU_DUAL, out_planes=pg, out_pair=pq, tile=tile, form=1)
                    outs.append((h, pg.buf[:, :, :N].clone(), pq.buf[:, :, :N].clone()))
                for other in outs[1:]:
                    for x, y in zip(outs[0], other):
                        assert torch.equal(x, y), (M, N, K, amp)
                # planes only (the Q/K/V projection: no fp32 output at all)
                pq1, pq2 = ops.pair_empty(M, N, dev), ops.pair_empty(M, N, dev)
                ops.plane_gemm(qa, qb, None, bias=bias, out_pair=pq1, tile=128129, form=1)
                ops.plane_gemm(qa, qb, None, bias=bias, out_pair=pq2, tile=64064, form=1)
                assert torch.equal(pq1.buf[:, :, :N], pq2.buf[:, :, :N])
        ref = a.double() @ b.double().t() + bias.double()
        scale = float((a.double().abs() @ b.double().abs().t()).max())
        with ops.amp_scope(False):
            out = torch.empty(M, N, device=dev)
            ops.plane_gemm(qa, qb, out, bias=bias, tile=64064, form=1)
        assert float((out.double() - ref).abs().max()) <= 2e-6 * scale
    # what the small tile's epilogue does not have is refused, not ignored
    with pytest.raises(Exception):
        ops.plane_gemm(qa, qb, torch.empty(M, N, device=dev), tile=64064, form=1, c_amax=ops.amax_slot(dev))


@pytest.mark.parametrize("tile", [128129, 128130, 256128])
def test_plane_gemm_pair_form_vs_fp64(tile):
    """csrc/gemm_planes.hip FORM 1: operands as two fp16 planes (hi, lo' = (x - hi) 2^11, round to nearest), three piece products, the
    cross products in their own accumulators.  Same bound against fp64 as the six-product bf16 form for operands anywhere inside
    fp16's range (rows scaled by 2^-8 ... 2^8, and far smaller / larger whole operands); pair planes written by the split kernel, by
    the GEMM epilogue (GELU output) and by the LayerNorm forward are bit-identical; an operand outside the range is inf, not clipped."""
    from vbg import ops
    from vbg.lib import EPI_GELU_DUAL
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(tile + 5)
    for (M, N, K) in ((300, 264, 96), (1000, 772, 800), (257, 512, 3072), (4128, 768, 768)):
        a = (torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-8, 8, (M, 1), generator=g).float())).to(dev)
        b = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        ref = a.double() @ b.double().t() + bias.double()
        scale = float((a.double().abs() @ b.double().abs().t()).max())
        qa, qb = ops.split_planes_pair(a), ops.split_planes_pair(b)
        # the pieces: hi = fp16(x) (round to nearest), lo' = fp16((x - hi) * 2048)
        hi = a.half()
        assert torch.equal(qa.buf[0, :, :K].view(torch.float16), hi)
        assert torch.equal(qa.buf[1, :, :K].view(torch.float16), ((a - hi.float()) * 2048).half())
        out = torch.full((M, N), 7.0, device=dev)
        ops.plane_gemm(qa, qb, out, bias=bias, tile=tile, form=1)
        assert float((out.double() - ref).abs().max()) <= 2e-6 * scale, (M, N, K)
        # against the six-product form: both within fp32 summation noise of each other
        o6 = torch.empty(M, N, device=dev)
        ops.plane_gemm(ops.split_planes(a), ops.split_planes(b), o6, bias=bias, tile=tile)
        assert float((o6 - out).abs().max()) <= 2e-6 * scale
        # whole operands far from 1 (still inside fp16's range after the split: the scaled low piece keeps small values exact)
        for sa, sb in ((2.0 ** -10, 2.0 ** 4), (2.0 ** 4, 2.0 ** -12)):
            o = torch.empty(M, N, device=dev)
            ops.plane_gemm(ops.split_planes_pair(a * sa), ops.split_planes_pair(b * sb), o, tile=tile, form=1)
            assert float((o.double() / (sa * sb) - (ref - bias.double())).abs().max()) <= 2e-6 * scale, (sa, sb)
        # GELU-dual epilogue: h fp32, gelu(h) as bf16 planes (the weight gradient's operand) AND as pair planes (the next product's)
        h, gl = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
        pg, pq = ops.planes_empty(M, N, dev), ops.pair_empty(M, N, dev)
        ops.plane_gemm(qa, qb, h, bias=bias, epi=EPI_GELU_DUAL, C2=gl, out_planes=pg, out_pair=pq, tile=tile, form=1)
        assert torch.equal(h, out)
        assert torch.equal(pg.buf[:, :, :N], ops.split_planes(gl).buf[:, :, :N])
        assert torch.equal(pq.buf[:, :, :N], ops.split_planes_pair(gl).buf[:, :, :N])
    # out of range: visible
    big = torch.full((256, 64), 70000.0, device=dev)
    o = torch.empty(256, 128, device=dev)
    ops.plane_gemm(ops.split_planes_pair(big), ops.split_planes_pair(torch.ones(128, 64, device=dev)), o, tile=tile, form=1)
    assert not bool(torch.isfinite(o).any())
    # LayerNorm forward writes the pair planes of its output next to the bf16 planes
    x, r = torch.randn(520, 768, generator=g).to(dev), torch.randn(520, 768, generator=g).to(dev)
    gam, bet = (1 + 0.1 * torch.randn(768, generator=g)).to(dev), (0.1 * torch.randn(768, generator=g)).to(dev)
    pl, pq = ops.planes_empty(520, 768, dev), ops.pair_empty(520, 768, dev)
    y, _, _ = ops.dropout_add_ln_fwd(x, r, gam, bet, 1e-12, 0.0, 1, 2, out_planes=pl, out_pair=pq)
    assert torch.equal(pl.buf, ops.split_planes(y).buf) and torch.equal(pq.buf, ops.split_planes_pair(y).buf)


def test_layernorm_backward_bound_scaled_pair_planes():
    """round 4: the LayerNorm backward delivers dx as fp16-pair planes scaled by a rigorous bound (max |dy| x max |gamma| x max rstd x
    (2 + sqrt H) / keep) instead of fp32 + amax + split pass: the planes equal the scaled split of the fp32-path dx bit for bit, the bound
    really bounds (and is at most 2^10 loose on random data), dres / dgamma / dbeta / the bias column sums / max |dx| equal the fp32 path's,
    at gradient magnitudes 2^-30 ... 2^8, with and without dropout."""
    from vbg import ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(91)
    rows, hid = 1000, 768
    x, r = torch.randn(rows, hid, generator=g).to(dev), torch.randn(rows, hid, generator=g).to(dev)
    gam, bet = (1 + 0.2 * torch.randn(hid, generator=g)).to(dev), (0.1 * torch.randn(hid, generator=g)).to(dev)
    for p_drop in (0.0, 0.1):
        y, xhat, rstd = ops.dropout_add_ln_fwd(x, r, gam, bet, 1e-12, p_drop, 5, 3)
        for sc in (1.0, 2.0 ** -30, 2.0 ** -17, 2.0 ** 8):
            dy = (torch.randn(rows, hid, generator=g) * torch.exp2(torch.randint(-4, 4, (rows, 1), generator=g).float())).to(dev) * sc
            dg0, db0 = torch.zeros(hid, device=dev), torch.zeros(hid, device=dev)
            s0 = ops.amax_slot(dev)
            dx0, dres0 = ops.dropout_add_ln_bwd(dy, xhat, rstd, gam, p_drop, 5, 3, dg0, db0, dx_amax=s0)
            dg1, db1, dbias = torch.zeros(hid, device=dev), torch.zeros(hid, device=dev), torch.zeros(hid, device=dev)
            s_true, s_ref = ops.amax_slot(dev), ops.amax_slot(dev)
            q, dres1 = ops.dropout_add_ln_bwd_pair(dy, xhat, rstd, gam, p_drop, 5, 3, dg1, db1, dbias, ops.amax(dy), s_true, s_ref)
            assert torch.equal(dres0, dres1)
            assert int(s_true.max().item()) == int(s0.max().item())
            bound, true_max = float(s_ref.view(torch.float32).max().item()), float(dx0.abs().max())
            assert true_max <= bound <= 2.0 ** 10 * true_max, (bound, true_max)
            chk = ops.split_planes_pair(dx0, amax_slot_=s_ref)
            assert torch.equal(q.buf, chk.buf), (p_drop, sc)
            assert torch.allclose(dg0, dg1, rtol=1e-5, atol=1e-5 * float(dg0.abs().max())) and torch.allclose(db0, db1, rtol=1e-5, atol=1e-5 * float(db0.abs().max()))
            assert torch.allclose(dbias.double(), dx0.double().sum(0), rtol=1e-4, atol=2e-5 * float(dx0.abs().max()) * rows ** 0.5)


def test_plane_gemm_bound_scaled_pair_output():
    """round 4: a GRADIENT leaves the product's epilogue as fp16-pair planes scaled by the power of two of a rigorous BOUND of the stored
    values -- max |A| (amax slot) x the largest column L1 norm of W (vbg_col_l1_max) x the epilogue's constant -- instead of a measured
    maximum: no fp32 round trip, no split pass.  Checked: the L1 word against torch; the bound really bounds; the planes are those of
    (stored value) x 2^e with e from the bound (bit for bit against the scaled split of the fp32 result); the bias column sums; and the
    consumers (NT data gradient, TN weight gradient) with the published bound as their a_amax against fp64 -- at gradient magnitudes
    2^-30 ... 2^10, with the GELU-gradient epilogue of the BERT backward."""
    from vbg import ops
    from vbg.lib import EPI_MUL_GELU_GRAD
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(77)
    M, K, N = 1000, 96 * 3, 768 + 64                        # tokens, features of dfo, features of h
    w = (torch.randn(K, N, generator=g) / K ** 0.5).to(dev)            # nn.Linear(N -> K).weight: [K, N]; dh = dfo @ w
    w2 = (torch.randn(N, 160, generator=g) / N ** 0.5).to(dev)
    xs = torch.randn(M, 64, generator=g).to(dev)
    h = torch.randn(M, N, generator=g).to(dev) * 1.5
    l1 = ops.weight_col_l1max(w)
    assert abs(float(l1.view(torch.float32).item()) - float(w.abs().sum(0).max())) <= 1e-5 * float(w.abs().sum(0).max())
    qwt = ops.split_planes_pair(w.t().contiguous())                    # B operand of the NT product: rows = features of h
    for sc in (1.0, 2.0 ** -30, 2.0 ** -19, 2.0 ** 10):
        dfo = (torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-5, 5, (M, 1), generator=g).float())).to(dev) * sc
        s_in = ops.amax(dfo)
        qd = ops.split_planes_pair(dfo, amax_slot_=s_in)
        ref = (dfo.double() @ w.double()) * torch.special.erf(h.double() / 2 ** 0.5).mul(0.5).add(0.5).add(
            h.double() * torch.exp(-0.5 * h.double() ** 2) / (2 * math.pi) ** 0.5)                # dfo w o gelu'(h)
        scale = float((dfo.double().abs() @ w.double().abs()).max())
        # measured path: fp32 result, its own scaled split
        dh = torch.empty(M, N, device=dev)
        cam = ops.amax_slot(dev)
        ops.plane_gemm(qd, qwt, dh, epi=EPI_MUL_GELU_GRAD, C2=h, tile=128129, form=1, a_amax=s_in, c_amax=cam)
        assert float((dh.double() - ref).abs().max()) <= 3e-6 * scale, sc
        # bound path: planes straight from the epilogue
        s_b = ops.amax_slot(dev)
        cs = torch.zeros((N,), device=dev)
        qh = ops.pair_empty(M, N, dev)
        ops.plane_gemm(qd, qwt, None, epi=EPI_MUL_GELU_GRAD, C2=h, tile=128129, form=1, a_amax=s_in, out_pair=qh, q_ref_in=s_in, q_l1=l1,
                       q_mul=1.13 * 1.01, q_ref_out=s_b, colsum_out=cs)
        bound = float(s_b.view(torch.float32).max().item())
        true_max = float(dh.abs().max())
        assert bound >= true_max and bound <= 2.0 ** 12 * true_max, (bound, true_max)            # rigorous, and not absurdly loose
        e = 13 - int(math.floor(math.log2(bound)))
        assert torch.equal(qh.buf[0, :, :N].view(torch.float16), (dh * 2.0 ** e).half())          # hi plane = fp16(stored value * 2^e)
        chk = ops.split_planes_pair(dh, amax_slot_=s_b)                                            # the scaled split of the fp32 result, same slot
        assert torch.equal(qh.buf[:, :, :N], chk.buf[:, :, :N])
        assert torch.allclose(cs.double() / sc, dh.double().sum(0) / sc, rtol=1e-5, atol=2e-4 * M ** 0.5)
        # consumers with the bound as their scale: NT data gradient dh @ w2 and TN weight gradient dh^T xs
        dx = torch.empty(M, 160, device=dev)
        ops.plane_gemm(qh, ops.split_planes_pair(w2.t().contiguous()), dx, tile=128129, form=1, a_amax=s_b)
        r2 = dh.double() @ w2.double()
        assert float((dx.double() - r2).abs().max()) <= 3e-6 * float((dh.double().abs() @ w2.double().abs()).max()), sc
        dw = torch.zeros(N, 64, device=dev)
        ops.plane_gemm_grouped([(qh, ops.split_planes_pair(xs), dw)], trans=True, accumulate=True, form=1, a_amax=[s_b])
        r3 = dh.double().t() @ xs.double()
        assert float((dw.double() - r3).abs().max()) <= 3e-6 * float((dh.double().abs().t() @ xs.double().abs()).max()), sc
    # the word follows the weights (library writes are seen through the weight epoch)
    ops.scale_(w.view(-1), 2.0)
    ops.bump_weight_epoch()
    assert abs(float(ops.weight_col_l1max(w).view(torch.float32).item()) - float(w.abs().sum(0).max())) <= 1e-5 * float(w.abs().sum(0).max())


def test_plane_gemm_pair_form_gradients_vs_fp64():
    """FORM 1 with a GRADIENT as the A operand: planes written scaled by the power of two of the tensor's largest magnitude (amax slot),
    products scaled back -- the NT data gradient dX = dY W and the TN weight gradients dW = dY^T X, single and grouped, at gradient
    magnitudes 2^-30 ... 2^20; the column sums of dY (bias gradient) ride on the scaled split unscaled."""
    from vbg import ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(31)
    M, N, K = 1000, 772 - 4, 96 * 3                      # tokens, out features, in features
    x = (torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-4, 4, (M, 1), generator=g).float())).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    for sc in (1.0, 2.0 ** -30, 2.0 ** -21, 2.0 ** 20):
        dy = (torch.randn(M, N, generator=g) * torch.exp2(torch.randint(-6, 6, (M, 1), generator=g).float())).to(dev) * sc
        slot = ops.amax(dy)
        cs = torch.zeros((N,), device=dev)
        qdy = ops.split_planes_pair(dy, amax_slot_=slot, colsum_out=cs)
        assert torch.allclose(cs.double() / sc, dy.double().sum(0) / sc, rtol=1e-5, atol=1e-4 * M ** 0.5)
        # the stored planes are those of dy * 2^e with amax * 2^e in [2^13, 2^14)
        e = 13 - int(torch.floor(torch.log2(dy.abs().max())).item())
        assert torch.equal(qdy.buf[0, :, :N].view(torch.float16), (dy * 2.0 ** e).half())
        # NT data gradient: dx = dy @ w   (B operand = planes of w^T)
        qwt = ops.split_planes_pair(w.t().contiguous())
        dx = torch.empty(M, K, device=dev)
        ops.plane_gemm(qdy, qwt, dx, tile=128129, form=1, a_amax=slot)
        ref = dy.double() @ w.double()
        scale = float((dy.double().abs() @ w.double().abs()).max())
        assert float((dx.double() - ref).abs().max()) <= 2e-6 * scale, sc
        # ... with the largest magnitude of the stored values as a by-product
        cam = ops.amax_slot(dev)
        ops.plane_gemm(qdy, qwt, dx, tile=128129, form=1, a_amax=slot, c_amax=cam)
        assert int(cam.max().item()) == int(dx.abs().max().view(torch.int32).item())
        # TN weight gradient: dw = dy^T @ x, both tiles, accumulate, single and grouped (three problems, their own slots)
        qx = ops.split_planes_pair(x)
        refw = dy.double().t() @ x.double()
        scw = float((dy.double().abs().t() @ x.double().abs()).max())
        for tile in (128129, 256128):
            dw = torch.full((N, K), sc, device=dev)
            ops.plane_gemm(qdy, qx, dw, trans=True, accumulate=True, tile=tile, form=1, a_amax=slot)
            assert float((dw.double() - sc - refw).abs().max()) <= 2e-6 * scw, (sc, tile)
        dy2 = dy[:, :256].contiguous() * 4
        slot2 = ops.amax(dy2)
        qdy2 = ops.split_planes_pair(dy2, amax_slot_=slot2)
        outs = [torch.zeros(N, K, device=dev), torch.zeros(256, K, device=dev), torch.zeros(N, K, device=dev)]
        ops.plane_gemm_grouped([(qdy, qx, outs[0]), (qdy2, qx, outs[1]), (qdy, qx, outs[2])], trans=True, accumulate=True, form=1,
                               a_amax=[slot, slot2, slot])
        assert float((outs[0].double() - refw).abs().max()) <= 2e-6 * scw and torch.equal(outs[0], outs[2])
        assert float((outs[1].double() - 4 * refw[:256]).abs().max()) <= 8e-6 * scw


def test_plane_gemm_tn_and_grouped_vs_fp64():
    """weight-gradient form: dW = dY^T X straight from the untransposed planes (LDS transpose reads), ragged sizes, reduction
    lengths that are not a multiple of the k-tile, and the grouped launch"""
    from vbg import ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(11)
    probs, refs = [], []
    for (Mt, N1, N2) in ((77, 128, 128), (1000, 264, 136), (4100, 768, 96), (515, 40, 2304)):
        dy = (torch.randn(Mt, N1, generator=g) * torch.exp2(torch.randint(-6, 6, (Mt, 1), generator=g).float())).to(dev)
        x = torch.randn(Mt, N2, generator=g).to(dev)
        ref = dy.double().t() @ x.double()
        scale = float((dy.double().abs().t() @ x.double().abs()).max())
        pdy, px = ops.split_planes(dy), ops.split_planes(x)
        for tile in (128129, 128130, 256128):
            out = torch.full((N1, N2), 3.0, device=dev)
            ops.plane_gemm(pdy, px, out, trans=True, tile=tile)
            assert float((out.double() - ref).abs().max()) <= 2e-6 * scale, (Mt, N1, N2, tile)
        acc = torch.ones(N1, N2, device=dev)
        ops.plane_gemm(pdy, px, acc, trans=True, accumulate=True)
        assert float((acc.double() - 1 - ref).abs().max()) <= 2e-6 * scale
        # transposed planes + the NT kernel give the same numbers
        o3 = torch.empty(N1, N2, device=dev)
        ops.plane_gemm(ops.split_planes_t(dy), ops.split_planes_t(x), o3)
        assert float((o3.double() - ref).abs().max()) <= 2e-6 * scale
    Mt = 1031
    for (N1, N2) in ((768, 256), (256, 768), (96, 96), (320, 64)):
        dy, x = torch.randn(Mt, N1, generator=g).to(dev), torch.randn(Mt, N2, generator=g).to(dev)
        probs.append((ops.split_planes(dy), ops.split_planes(x), torch.ones(N1, N2, device=dev)))
        refs.append(dy.double().t() @ x.double())
    ops.plane_gemm_grouped(probs, trans=True, accumulate=True)
    for (_, _, out), ref in zip(probs, refs):
        assert float((out.double() - 1 - ref).abs().max()) <= 2e-6 * float(ref.abs().max()) * 30
    # the 256 x 128 tile form of the grouped launch (512-byte LDS rows for the A operand)
    for (_, _, out) in probs:
        out.fill_(1.0)
    ops.plane_gemm_grouped(probs, trans=True, accumulate=True, tile=256128)
    for (_, _, out), ref in zip(probs, refs):
        assert float((out.double() - 1 - ref).abs().max()) <= 2e-6 * float(ref.abs().max()) * 30


def test_ce_label_out_of_range_is_nan_not_oob():
    """a label >= ncls (misconfigured num_classes) yields a NaN loss and no out-of-bounds access; in-range elements are untouched"""
    from vbg import ops
    dev = torch.device("cuda")
    x = torch.randn(16, 5, device=dev)
    lab = torch.tensor([0, 1, 2, 3, 4, 5, -1, 2] * 2, dtype=torch.int32, device=dev)
    elem = torch.arange(16, dtype=torch.int32, device=dev)
    loss = ops.ce_fwd(x, elem, lab, 16, None, 0, 0, 0)
    bad = (lab < 0) | (lab >= 5)
    assert torch.isnan(loss[bad]).all() and torch.isfinite(loss[~bad]).all()
    ref = torch.nn.functional.cross_entropy(x[~bad], lab[~bad].long(), reduction="none")
    assert torch.allclose(loss[~bad], ref, rtol=1e-5, atol=1e-6)
    dl = torch.zeros_like(x)
    ops.ce_bwd(x, elem, lab, 16, None, torch.ones(1, device=dev), 1.0, 0, 0, 0, dl)
    assert torch.isfinite(dl).all() and float(dl[bad].abs().max()) == 0.0


@pytest.mark.parametrize("tile", [128129, 128130])
def test_plane_gemm_streamk_tail(tile):
    """stream-K tail of the NT plane product (the last, partly filled round of tiles is cut along k over all CUs and summed by the
    last block to finish each tile): fp32-grade against fp64, bit-identical to itself run to run (slabs are summed in block
    order), every epilogue, and the plain-rounds kernel as the second opinion"""
    from vbg import ops
    from vbg.lib import EPI_GELU_DUAL
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(tile)
    ops.set_streamk(True)
    for (M, N, K) in ((4128, 768, 768), (4128, 2304, 768), (4128, 768, 3072), (4128, 3072, 768), (2100, 1536, 1056)):
        a = (torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-8, 8, (M, 1), generator=g).float())).to(dev)
        b = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        ref = a.double() @ b.double().t() + bias.double()
        scale = float((a.double().abs() @ b.double().abs().t()).max())
        pa, pb = ops.split_planes(a), ops.split_planes(b)
        outs = []
        for rep in range(3):
            out = torch.full((M, N), 7.0, device=dev)
            ops.plane_gemm(pa, pb, out, bias=bias, tile=tile)
            outs.append(out)
        assert float((outs[0].double() - ref).abs().max()) <= 2e-6 * scale, (M, N, K)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
        ops.set_streamk(False)
        try:
            plain = torch.empty(M, N, device=dev)
            ops.plane_gemm(pa, pb, plain, bias=bias, tile=tile)
        finally:
            ops.set_streamk(True)
        assert float((plain - outs[0]).abs().max()) <= 1e-6 * scale
        h, gl, pg = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev), ops.planes_empty(M, N, dev)
        ops.plane_gemm(pa, pb, h, bias=bias, epi=EPI_GELU_DUAL, C2=gl, out_planes=pg, tile=tile)
        assert torch.equal(h, outs[0])
        assert torch.equal(pg.buf[:, :, :N], ops.split_planes(gl).buf[:, :, :N])
        only_planes = ops.planes_empty(M, N, dev)
        ops.plane_gemm(pa, pb, None, bias=bias, out_planes=only_planes, tile=tile)
        assert torch.equal(only_planes.buf[:, :, :N], ops.split_planes(outs[0]).buf[:, :, :N])
        acc = torch.ones(M, N, device=dev)
        ops.plane_gemm(pa, pb, acc, accumulate=True, tile=tile)
        assert float((acc.double() - 1 - (ref - bias.double())).abs().max()) <= 2e-6 * scale
    # the tile counters are left zero for the next launch
    _, cnt, _ = ops._sk_workspace(dev)
    ops.set_streamk(False)
    assert int(cnt.abs().sum()) == 0



def test_plane_gemm_256_tile():
    """256 x 256 tile form of the NT plane product (16-deep k-tiles, 32-byte LDS rows): fp32-grade vs fp64 and every epilogue"""
    from vbg import ops
    from vbg.lib import EPI_GELU_DUAL, EPI_MUL_GELU_GRAD
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(256)
    for (M, N, K) in ((4128, 2304, 768), (1000, 772, 800), (257, 512, 3072), (300, 264, 96), (256, 256, 32)):
        a = (torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-8, 8, (M, 1), generator=g).float())).to(dev)
        b = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        ref = a.double() @ b.double().t() + bias.double()
        scale = float((a.double().abs() @ b.double().abs().t()).max())
        pa, pb = ops.split_planes(a), ops.split_planes(b)
        out = torch.full((M, N), 7.0, device=dev)
        ops.plane_gemm(pa, pb, out, bias=bias, tile=256256)
        assert float((out.double() - ref).abs().max()) <= 2e-6 * scale, (M, N, K)
        other = torch.empty(M, N, device=dev)
        ops.plane_gemm(pa, pb, other, bias=bias, tile=128129)
        assert float((other - out).abs().max()) <= 1e-6 * scale
        h, gl, pg = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev), ops.planes_empty(M, N, dev)
        ops.plane_gemm(pa, pb, h, bias=bias, epi=EPI_GELU_DUAL, C2=gl, out_planes=pg, tile=256256)
        assert torch.equal(h, out)
        assert torch.equal(pg.buf[:, :, :N], ops.split_planes(gl).buf[:, :, :N])
        only = ops.planes_empty(M, N, dev)
        ops.plane_gemm(pa, pb, None, bias=bias, out_planes=only, tile=256256)
        assert torch.equal(only.buf[:, :, :N], ops.split_planes(out).buf[:, :, :N])
        hh = (torch.randn(M, N, generator=g) * 2).to(dev)
        dh = torch.empty(M, N, device=dev)
        ops.plane_gemm(pa, pb, dh, bias=bias, epi=EPI_MUL_GELU_GRAD, C2=hh, tile=256256)
        want = out.clone()
        ops.gelu_bwd_(hh, want)
        assert torch.equal(dh, want)
