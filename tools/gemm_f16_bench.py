#!/usr/bin/env python3
"""The generic kernels' forward products, six bf16 piece products (bf16 = 3) vs three fp16 piece products (bf16 = 2), alone on the chip."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from gemm_bench import report, timeit
from vbg import ops
from vbg.lib import OP_DENSE_K, OP_CONV_K

d = torch.device("cuda")
g = torch.Generator(device=d).manual_seed(1)
for (M, N, K) in ((1024, 1024, 12544), (131072, 256, 1024), (131072, 256, 64), (32768, 128, 896), (32768, 256, 128), (8192, 256, 256), (1024, 512, 1024)):
    x, w = torch.randn(M, K, device=d, generator=g), torch.randn(N, K, device=d, generator=g) / K ** 0.5
    out = torch.empty(M, N, device=d)
    for f in (False, True):
        t = timeit(lambda: ops.gemm_raw(M, N, K, x, K, OP_DENSE_K, w, K, OP_DENSE_K, out, N, f16=f))
        report(f"NT {M}x{N}x{K} {'fp16 pair (3 products)' if f else 'bf16 x 3 (6 products)'}", 2.0 * M * N * K, t)
for (B, H, W, Ci, Co, k, s_, p_) in ((8, 128, 128, 64, 128, 3, 2, 1), (8, 64, 64, 128, 256, 3, 2, 1), (8, 32, 32, 256, 512, 3, 2, 1)):
    x = torch.randn(B, H, W, Ci, device=d, generator=g)
    w = torch.randn(Co, k, k, Ci, device=d, generator=g) / (Ci * k * k) ** 0.5
    for f in (False, True):
        ops.set_gemm_f16(f)
        t = timeit(lambda: ops.conv2d_fwd(x, w, s_, p_))
        report(f"conv fwd B{B} {H}x{W} {Ci}->{Co} k{k}s{s_} {'fp16 pair' if f else 'bf16 x 3'}", 2.0 * B * (H // s_) * (W // s_) * Ci * Co * k * k, t)
ops.set_gemm_f16(True)
