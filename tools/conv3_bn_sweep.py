#!/usr/bin/env python3
"""PW forward kernel: filters per tile (128 / 64) x workgroups per tile on the shapes whose 128-filter tiles do not fill the chip.
   python tools/conv3_bn_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from gemm_bench import report, timeit
from vbg import ops
dev = torch.device("cuda")
for (B, H, W, Ci, Co) in [(8, 32, 32, 256, 256), (8, 16, 16, 512, 512), (8, 64, 64, 128, 128), (8, 64, 64, 256, 256), (8, 128, 128, 128, 128)]:
    x = torch.randn(B, H, W, Ci, device=dev)
    wd = (torch.randn(Co, Ci, 3, 3, device=dev) / (3 * Ci ** 0.5)).contiguous(memory_format=torch.channels_last)
    w4 = wd.permute(0, 2, 3, 1)
    fl = 2.0 * B * H * W * Ci * Co * 9
    ref = ops.conv3x3(x, w4, f16x2=True, nsplit=1) if ops.conv3_pw_ok(B, H, W, Ci, Co) and ops.conv3_split(B, H, W, Ci, Co) == 1 else ops.conv3x3(x, w4, f16x2=True)
    for bn in (128, 64):
        wp = ops.conv3_planes(wd, w4, False, bn=bn)
        for z in (1, 2, 3, 4, 6):
            cs = z // 3 if z % 3 == 0 else z
            if Ci % cs or (Ci // cs) % 16:
                continue
            y = ops.conv3x3(x, w4, f16x2=True, w_planes=wp, nsplit=z, bn=bn)
            err = float((y - ref).abs().max())
            tiles = (B * H * W // 128) * (Co // bn) * z
            report(f"B{B} {H}x{W} {Ci}->{Co} bn {bn:3d} nsplit {z} ({tiles:4d} blocks, max diff {err:.1e})", fl, timeit(lambda: ops.conv3x3(x, w4, f16x2=True, w_planes=wp, nsplit=z, bn=bn)))
