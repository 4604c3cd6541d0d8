#!/usr/bin/env python3
"""N launches of one 3x3 convolution form for counter collection: python tools/conv3_prof.py conv3|generic [fwd|wgrad] [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd"))
import torch

from vbg import ops

dev = torch.device("cuda")
which, what, reps = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "fwd"), int(sys.argv[3]) if len(sys.argv) > 3 else 30
B, H, W, C = 8, 128, 128, 256
x = torch.randn(B, H, W, C, device=dev)
w = torch.randn(C, 3, 3, C, device=dev) / 48
dw = torch.zeros_like(w)
ops.set_conv3(which in ("conv3", "pw"))
wd = (torch.randn(C, C, 3, 3, device=dev) / 48).contiguous(memory_format=torch.channels_last)
w4 = wd.permute(0, 2, 3, 1)
wp = ops.conv3_planes(wd, w4, False)
if len(sys.argv) > 4:          # B H W C
    B, H, W, C = (int(v) for v in sys.argv[4:8])
    x = torch.randn(B, H, W, C, device=dev)
    wd = (torch.randn(C, C, 3, 3, device=dev) / 48).contiguous(memory_format=torch.channels_last)
    w4 = wd.permute(0, 2, 3, 1)
    wp = ops.conv3_planes(wd, w4, False)
for _ in range(reps):
    if which == "pw":
        ops.conv3x3(x, w4, f16x2=True, w_planes=wp)
    elif what == "fwd":
        ops.conv2d_fwd(x, w, 1, 1)
    else:
        ops.conv2d_wgrad(x, x, dw, 1, 1)
torch.cuda.synchronize()
