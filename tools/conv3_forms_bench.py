#!/usr/bin/env python3
"""csrc/conv3.hip: arithmetic forms of the forward and the input gradient on the wide cfg2 / cfg5 shapes.
   python tools/conv3_forms_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch

from gemm_bench import report, timeit
from vbg import ops

dev = torch.device("cuda")
for (B, H, W, Ci, Co) in [(8, 128, 128, 256, 256), (8, 128, 128, 128, 128), (8, 64, 64, 128, 128), (8, 64, 64, 256, 256), (8, 32, 32, 256, 256), (8, 16, 16, 512, 512)]:
    x = torch.randn(B, H, W, Ci, device=dev)
    w = torch.randn(Co, 3, 3, Ci, device=dev) / (3 * Ci ** 0.5)
    dy = torch.randn(B, H, W, Co, device=dev) * 1e-6
    wf = ops.conv3x3_wflip(w)
    fl = 2.0 * B * H * W * Ci * Co * 9
    tag = f"B{B} {H}x{W} {Ci}->{Co}"
    report(f"fwd   bf16x3        {tag}", fl, timeit(lambda: ops.conv3x3(x, w)))
    report(f"fwd   f16x2         {tag}", fl, timeit(lambda: ops.conv3x3(x, w, f16x2=True)))
    report(f"dgrad bf16x3        {tag}", fl, timeit(lambda: ops.conv3x3(dy, wf)))
    am = ops.amax(dy)
    report(f"dgrad f16x2 (amax given) {tag}", fl, timeit(lambda: ops.conv3x3(dy, wf, f16x2=True, x_amax=am)))
    report(f"dgrad f16x2 + amax pass  {tag}", fl, timeit(lambda: ops.conv3x3(dy, wf, f16x2=True, x_amax=ops.amax(dy))))
    report(f"amax pass           {tag}", fl, timeit(lambda: ops.amax(dy)))
    dw = torch.zeros_like(w)
    report(f"wgrad bf16x3        {tag}", fl, timeit(lambda: ops.conv3x3_wgrad(dy, x, dw)))
    ax = ops.amax(x)
    report(f"wgrad f16x2 (amax given) {tag}", fl, timeit(lambda: ops.conv3x3_wgrad(dy, x, dw, f16x2=True, dy_amax=am, x_amax=ax)))
    d0, d1 = torch.zeros_like(w), torch.zeros_like(w)
    ops.conv3x3_wgrad(dy, x, d0); ops.conv3x3_wgrad(dy, x, d1, f16x2=True, dy_amax=am, x_amax=ax)
    print("   wgrad max |f16 - bf16| / max |dw| =", float((d0 - d1).abs().max() / d0.abs().max()), flush=True)
    a, b = ops.conv3x3(dy, wf), ops.conv3x3(dy, wf, f16x2=True, x_amax=am)
    print("   max |f16 - bf16| / max |dx| =", float((a - b).abs().max() / a.abs().max()), flush=True)
