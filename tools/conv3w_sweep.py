#!/usr/bin/env python3
"""weight-gradient strips sweep: VBG_CONV3W_BLOCKS / VBG_CONV3W_SLAB_MB are read once per process -> one process per setting"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch

from gemm_bench import timeit
from vbg import lib, ops

dev = torch.device("cuda")
out = []
for (B, H, W, Ci, Co) in [(8, 128, 128, 64, 64), (8, 128, 128, 256, 256), (8, 128, 128, 128, 128), (8, 64, 64, 128, 128), (8, 32, 32, 256, 256), (8, 16, 16, 512, 512)]:
    x = torch.randn(B, H, W, Ci, device=dev)
    dy = torch.randn(B, H, W, Co, device=dev)
    dw = torch.zeros(Co, 3, 3, Ci, device=dev)
    s = lib.lib.vbg_conv3x3_wgrad_strips(B, H, W, Ci, Co)
    t1 = timeit(lambda: ops.conv3x3_wgrad(dy, x, dw, slabs=True))
    t2 = timeit(lambda: ops.conv3x3_wgrad(dy, x, dw, slabs=False))
    out.append(f"{H}x{W} {Ci}->{Co}: strips {s} slab {t1 * 1e6:.0f} us atomics {t2 * 1e6:.0f} us")
print(os.environ.get("VBG_CONV3W_BLOCKS"), os.environ.get("VBG_CONV3W_SLAB_MB"), " | ".join(out), flush=True)
