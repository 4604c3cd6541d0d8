#!/usr/bin/env python3
"""Plane GEMM (csrc/gemm_planes.hip) against fp64 and against the in-kernel split form of vbg_gemm: accuracy and time per shape.

    python tools/plane_gemm_bench.py [--tiles 128128,64064,256128]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd"))
from vbg import ops  # noqa: E402
from vbg.lib import EPI_GELU_DUAL, EPI_NONE, OP_DENSE_K  # noqa: E402

SHAPES = [  # (M, N, K, what)
    (4128, 2304, 768, "QKV fwd"), (4128, 768, 768, "attn out fwd"), (4128, 3072, 768, "FFN1 fwd"), (4128, 768, 3072, "FFN2 fwd"),
    (3072, 768, 4128, "FFN1 wgrad"), (768, 3072, 4128, "FFN2 wgrad"), (2304, 768, 4128, "QKV wgrad"), (768, 768, 4128, "out wgrad"),
    (131072, 256, 64, "1x1 64->256"), (1024, 1024, 12544, "ROI linear fwd"), (1024, 512, 1024, "head mlp"),
    (1000, 772, 96, "ragged"),
]


def timed(fn, iters=20, warm_ms=50.0):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < warm_ms:
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", default="64064,128064,128128,128129,128130,256128")
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda")
    tiles = [int(t) for t in args.tiles.split(",")]
    g = torch.Generator(device="cpu").manual_seed(0)
    for (M, N, K, what) in SHAPES:
        a = (torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-6, 6, (M, 1), generator=g).float())).to(dev)
        b = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        ref = (a.double() @ b.double().t() + bias.double())
        scale = float((a.double().abs() @ b.double().abs().t()).max())
        pa, pb = ops.split_planes(a), ops.split_planes(b)
        # transposed split check: planes of a^T via split_planes_t(a^T-source)
        at = a.t().contiguous()
        pat = ops.split_planes_t(at)            # [K rows of at -> transposed] = planes of a again
        assert torch.equal(pat.buf, pa.buf), "transposed split differs"
        out = torch.empty(M, N, device=dev)
        line = [f"{what:16s} {M}x{N}x{K}"]
        # in-kernel split form of vbg_gemm
        o2 = torch.empty(M, N, device=dev)
        f_old = lambda: ops.gemm_raw(M, N, K, a, K, OP_DENSE_K, b, K, OP_DENSE_K, o2, N, bias=bias)
        with torch.no_grad():
            f_old()
            err_old = float((o2.double() - ref).abs().max()) / scale
            t_old = timed(f_old)
        line.append(f"gemm.hip split {t_old:7.1f} us {2e-6 * M * N * K / t_old:6.1f} TF err {err_old:.1e}")
        for tile in tiles:
            f = lambda: ops.plane_gemm(pa, pb, out, bias=bias, tile=tile)
            out.zero_()
            f()
            err = float((out.double() - ref).abs().max()) / scale
            t = timed(f)
            line.append(f"| {tile}: {t:7.1f} us {2e-6 * M * N * K / t:6.1f} TF err {err:.1e}")
        t_sp = timed(lambda: ops.split_planes(a, out=pa))
        t_spt = timed(lambda: ops.split_planes_t(at, out=pat))
        line.append(f"| split A {t_sp:6.1f} us ({(M * K * 10) / t_sp * 1e-6:.2f} TB/s) split_t {t_spt:6.1f} us")
        print(" ".join(line), flush=True)
    # epilogue variants: GELU dual + planes out, accumulate, split-K
    M, N, K = 1000, 768, 3072
    a = torch.randn(M, K, generator=g).to(dev)
    b = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    pa, pb = ops.split_planes(a), ops.split_planes(b)
    ref = a.double() @ b.double().t() + bias.double()
    c, c2 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    op = ops.planes_empty(M, N, dev)
    ops.plane_gemm(pa, pb, c, bias=bias, epi=EPI_GELU_DUAL, C2=c2, out_planes=op)
    gref = torch.nn.functional.gelu(ref)
    print("gelu dual: pre err", float((c.double() - ref).abs().max()), "gelu err", float((c2.double() - gref).abs().max()))
    chk = ops.split_planes(c2)
    print("planes out == split(C2):", bool(torch.equal(chk.buf, op.buf)))
    acc = torch.ones(M, N, device=dev)
    ops.plane_gemm(pa, pb, acc, accumulate=True)
    print("accumulate err", float((acc.double() - 1 - (ref - bias.double())).abs().max()))
    acc = torch.ones(M, N, device=dev)
    ops.plane_gemm(pa, pb, acc, accumulate=True, splitk=4)
    print("split-K 4 err", float((acc.double() - 1 - (ref - bias.double())).abs().max()))


if __name__ == "__main__":
    main()
