#!/usr/bin/env python3
"""Row-reuse 3x3 convolution (csrc/conv3.hip) against the generic implicit GEMM (csrc/gemm.hip) on the wide cfg2 shapes.
   python tools/conv3_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch

from gemm_bench import report, timeit
from vbg import ops

dev = torch.device("cuda")
for (B, H, W, Ci, Co) in [(8, 128, 128, 256, 256), (8, 128, 128, 128, 128), (8, 64, 64, 128, 128), (8, 64, 64, 256, 256), (8, 32, 32, 256, 256), (8, 16, 16, 512, 512), (8, 128, 128, 64, 64), (1024, 7, 7, 256, 256)]:
    x = torch.randn(B, H, W, Ci, device=dev)
    w = torch.randn(Co, 3, 3, Ci, device=dev) / (3 * Ci ** 0.5)
    dy = torch.randn(B, H, W, Co, device=dev)
    fl = 2.0 * B * H * W * Ci * Co * 9
    tag = f"B{B} {H}x{W} {Ci}->{Co}"
    for on in (False, True):
        ops.set_conv3(on)
        name = "conv3 " if on else "generic"
        report(f"{name} fwd   {tag}", fl, timeit(lambda: ops.conv2d_fwd(x, w, 1, 1)))
        report(f"{name} dgrad {tag}", fl, timeit(lambda: ops.conv2d_dgrad(dy, w, tuple(x.shape), 1, 1)))
        dw = torch.zeros_like(w)
        report(f"{name} wgrad {tag}", fl, timeit(lambda: ops.conv2d_wgrad(dy, x, dw, 1, 1)))
        if on:
            report(f"{name} wgrad (atomics) {tag}", fl, timeit(lambda: ops.conv3x3_wgrad(dy, x, dw, slabs=False)))
            if Co % 128 == 0 or (Co % 64 == 0 and Ci % 64 == 0):
                ay, ax = ops.amax(dy), ops.amax(x)
                report(f"{name} wgrad (kernel, bf16x3, slabs) {tag}", fl, timeit(lambda: ops.conv3x3_wgrad(dy, x, dw, slabs=True)))
                report(f"{name} wgrad (kernel, f16x2, slabs) {tag}", fl, timeit(lambda: ops.conv3x3_wgrad(dy, x, dw, slabs=True, f16x2=True, dy_amax=ay, x_amax=ax)))
                report(f"{name} wgrad (kernel, f16x2, atomics) {tag}", fl, timeit(lambda: ops.conv3x3_wgrad(dy, x, dw, slabs=False, f16x2=True, dy_amax=ay, x_amax=ax)))
    ops.set_conv3(True)
    y1 = ops.conv2d_fwd(x, w, 1, 1)
    ops.set_conv3(False)
    y0 = ops.conv2d_fwd(x, w, 1, 1)
    ops.set_conv3(True)
    print("   max |conv3 - generic| =", float((y1 - y0).abs().max()), " max |y| =", float(y0.abs().max()), flush=True)

# split form of the forward / input-gradient kernel on the late trunk stages: workgroups per tile
ops.set_conv3(True)
for (B, H, W, Ci, Co) in [(8, 32, 32, 256, 256), (8, 16, 16, 512, 512), (8, 64, 64, 128, 128), (8, 64, 64, 256, 256)]:
    x = torch.randn(B, H, W, Ci, device=dev)
    w = torch.randn(Co, 3, 3, Ci, device=dev) / (3 * Ci ** 0.5)
    fl = 2.0 * B * H * W * Ci * Co * 9
    for f16 in (True, False):
        for nz in (1, 2, 3, 4, 6, 8, 12):
            report(f"conv3 {'f16x2' if f16 else 'bf16x3'} B{B} {H}x{W} {Ci}->{Co} nsplit {nz}", fl, timeit(lambda: ops.conv3x3(x, w, f16x2=f16, nsplit=nz)))
