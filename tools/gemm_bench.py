#!/usr/bin/env python3
"""Per-shape throughput of the libvbg GEMM / implicit-conv variants on the cfg2 shapes (MI355X).
   python tools/gemm_bench.py [--tiles 64,128]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd"))
import torch

from vbg import ops
from vbg.lib import OP_CONV_K, OP_CONV_R, OP_DENSE_K, OP_DENSE_R, OP_WT_R

dev = torch.device("cuda")


def timeit(fn, reps=20):
    """time `reps` calls after ~50 ms of the same call: the device needs tens of milliseconds of sustained load to reach its loaded
    clock state (measured: the same 4128x3072x768 GEMM reads 209 us cold and 168 us after ~150 launches), and the training step,
    which is what these numbers stand for, never lets it idle"""
    import time
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.05:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def report(name, flops, t):
    print(f"{name:58s} {t * 1e6:9.1f} us  {flops / t / 1e12:7.1f} TF/s", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", default="0")
    ap.add_argument("--only", default="")
    ap.add_argument("--bks", default="0")
    ap.add_argument("--amp", action="store_true", help="bf16 matrix-core form (amp)")
    ap.add_argument("--split3", action="store_true", help="(default) fp32-grade bf16x3 split form")
    ap.add_argument("--fp32", action="store_true", help="fp32 MFMA form everywhere")
    args = ap.parse_args()
    ops.set_amp(args.amp)
    ops._SPLIT3[0] = not args.fp32 and not args.amp
    tiles = [int(t) for t in args.tiles.split(",")]
    R = lambda *s: torch.randn(*s, device=dev)
    import itertools
    for tile, bk in itertools.product(tiles, [int(b) for b in args.bks.split(",")]):
        print(f"---- tile={tile} bk={bk} (0 = library heuristic)")
        if not args.only or "lin" in args.only:
            for (M, N, K) in [(4128, 768, 768), (4128, 3072, 768), (4128, 768, 3072), (4128, 2304, 768), (1024, 1024, 12544), (1024, 512, 1024)]:
                x, w, y = R(M, K), R(N, K), torch.empty(M, N, device=dev)
                report(f"NT  linear fwd   M{M} N{N} K{K}", 2.0 * M * N * K,
                       timeit(lambda: ops.gemm_raw(M, N, K, x, K, OP_DENSE_K, w, K, OP_DENSE_K, y, N, tile=tile, bk=bk)))
                dy, dx = R(M, N), torch.empty(M, K, device=dev)
                report(f"NN  linear dgrad M{M} N{K} K{N}", 2.0 * M * N * K,
                       timeit(lambda: ops.gemm_raw(M, K, N, dy, N, OP_DENSE_K, w, K, OP_DENSE_R, dx, K, tile=tile, bk=bk)))
                dw = torch.zeros(N, K, device=dev)
                sk = ops._pick_splitk(N, K, M)
                report(f"TN  linear wgrad M{N} N{K} K{M} splitk{sk}", 2.0 * M * N * K,
                       timeit(lambda: ops.gemm_raw(N, K, M, dy, N, OP_DENSE_R, x, K, OP_DENSE_R, dw, K, accumulate=sk > 1, splitk=sk, tile=tile, bk=bk)))
        if not args.only or "conv" in args.only:
            for (B, H, W, Cin, Cout, k, s) in [(8, 128, 128, 64, 64, 3, 1), (8, 64, 64, 128, 128, 3, 1), (8, 32, 32, 256, 256, 3, 1),
                                               (8, 16, 16, 512, 512, 3, 1), (8, 128, 128, 256, 256, 3, 1), (1024, 7, 7, 256, 256, 3, 1),
                                               (8, 128, 128, 64, 128, 3, 2), (8, 128, 128, 64, 256, 1, 1)]:
                pad = k // 2
                x, w = R(B, H, W, Cin), R(Cout, k, k, Cin)
                Ho, Wo = ops.conv_out_hw(H, W, k, s, pad)
                y = torch.empty(B, Ho, Wo, Cout, device=dev)
                M, K = B * Ho * Wo, k * k * Cin
                fl = 2.0 * M * Cout * K
                tag = f"B{B} {H}x{W} {Cin}->{Cout} k{k}s{s}"
                if k == 1:
                    report(f"conv1x1 fwd  {tag}", fl, timeit(lambda: ops.gemm_raw(M, Cout, K, x, Cin, OP_DENSE_K, w, K, OP_DENSE_K, y, Cout, tile=tile, bk=bk)))
                    continue
                geo = ops.conv_geo(H, W, Cin, Ho, Wo, k, k, s, pad, 0)
                report(f"conv fwd   {tag}", fl, timeit(lambda: ops.gemm_raw(M, Cout, K, x, Cin, OP_CONV_K, w, K, OP_DENSE_K, y, Cout, geo=geo, tile=tile, bk=bk)))
                dy, dx = R(B, Ho, Wo, Cout), torch.empty(B, H, W, Cin, device=dev)
                gd = ops.conv_geo(Ho, Wo, Cout, H, W, k, k, s, pad, 1)
                report(f"conv dgrad {tag}", fl, timeit(lambda: ops.gemm_raw(B * H * W, Cin, k * k * Cout, dy, Cout, OP_CONV_K, w, Cin, OP_WT_R, dx, Cin, geo=gd, tile=tile, bk=bk)))
                dw = torch.zeros(Cout, k, k, Cin, device=dev)
                sk = ops._pick_splitk(Cout, K, M)
                report(f"conv wgrad {tag} splitk{sk}", fl,
                       timeit(lambda: ops.gemm_raw(Cout, K, M, dy, Cout, OP_DENSE_R, x, Cin, OP_CONV_R, dw, K, geo=geo, accumulate=sk > 1, splitk=sk, tile=tile, bk=bk)))


if __name__ == "__main__":
    main()
