#!/usr/bin/env python3
"""time of the plane GEMM vs K at fixed M x N (fixed overhead vs per-k-tile cost), and split-K sweeps for the wgrad shapes"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd"))
from vbg import ops  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tools"))
from plane_gemm_bench import timed  # noqa: E402

dev = torch.device("cuda")
tiles = [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "128129,128128,256128,64064").split(",")]
for (M, N) in ((4096, 3072), (4128, 3072), (4096, 768)):
    for tile in tiles:
        row = [f"{M}x{N} tile {tile}:"]
        for K in (64, 256, 768, 1536, 3072):
            a, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
            pa, pb = ops.split_planes(a), ops.split_planes(b)
            out = torch.empty(M, N, device=dev)
            t = timed(lambda: ops.plane_gemm(pa, pb, out, tile=tile))
            row.append(f"K={K}: {t:6.1f} us {2e-6 * M * N * K / t:5.0f} TF")
        print(" ".join(row), flush=True)
for (M, N, K) in ((3072, 768, 4128), (768, 768, 4128), (2304, 768, 4128)):
    a, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    pa, pb = ops.split_planes(a), ops.split_planes(b)
    out = torch.zeros(M, N, device=dev)
    for tile in (128129, 64064, 128064):
        row = [f"wgrad {M}x{N}x{K} tile {tile}:"]
        for sk in (1, 2, 3, 4, 5, 7, 8, 12, 16):
            t = timed(lambda: ops.plane_gemm(pa, pb, out, tile=tile, accumulate=True, splitk=sk))
            row.append(f"sk{sk}: {t:6.1f} ({2e-6 * M * N * K / t:4.0f})")
        print(" ".join(row), flush=True)
