#!/usr/bin/env python3
"""Row-reuse 3x3 convolution on a SINGLE document's late trunk stages (8-32 tiles): filters per tile x workgroups per tile, against the
generic 64 x 64 implicit-GEMM kernel the dispatch used to pick.   python tools/conv3_small_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from gemm_bench import report, timeit
from vbg import ops

dev = torch.device("cuda")
GEN = torch.Generator(device=dev).manual_seed(5)
for (B, H, W, Ci, Co) in [(1, 16, 16, 512, 512), (1, 32, 32, 256, 256), (1, 64, 64, 128, 128), (2, 16, 16, 512, 512)]:
    x = torch.randn(B, H, W, Ci, device=dev, generator=GEN)
    wd = (torch.randn(Co, Ci, 3, 3, device=dev, generator=GEN) / (3 * Ci ** 0.5)).contiguous(memory_format=torch.channels_last)
    w4 = wd.permute(0, 2, 3, 1)
    fl = 2.0 * B * H * W * Ci * Co * 9
    tag = f"B{B} {H}x{W} {Ci}->{Co}"
    w4c = w4.contiguous()
    ops._CONV3[0] = False
    report(f"generic 64 x 64 implicit GEMM    {tag}", fl, timeit(lambda: ops.conv2d_fwd(x, w4c, 1, 1)))
    ops._CONV3[0] = True
    report(f"conv2d_fwd as dispatched         {tag}", fl, timeit(lambda: ops.conv2d_fwd(x, w4c, 1, 1, w_owner=wd)))
    for bn in (128, 64):
        wp = ops.conv3_planes(wd, w4, False, bn=bn)
        for z in (1, 2, 3, 4, 6, 8, 12, 24):
            cs = z // 3 if z % 3 == 0 else z
            if Ci % cs or (Ci // cs) % 16:
                continue
            try:
                t = timeit(lambda: ops.conv3x3(x, w4, f16x2=True, w_planes=wp, nsplit=z, bn=bn))
            except Exception as e:
                print(f"   bn {bn} nsplit {z}: {type(e).__name__}")
                continue
            tiles = (B * H * W // 128) * (Co // bn)
            report(f"PW bn{bn:3d} nsplit {z:2d} ({tiles * z:4d} workgroups) {tag}", fl, t)
