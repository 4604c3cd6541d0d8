#!/usr/bin/env python3
"""One plane-GEMM shape / tile launched a few times (for rocprofv3 --pmc passes).  python tools/plane_gemm_prof.py M N K tile [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd"))
from vbg import ops  # noqa: E402

M, N, K, tile = (int(v) for v in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dev = torch.device("cuda")
a, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) / K ** 0.5
pa, pb = ops.split_planes(a), ops.split_planes(b)
out = torch.empty(M, N, device=dev)
for _ in range(iters):
    ops.plane_gemm(pa, pb, out, tile=tile)
torch.cuda.synchronize()
