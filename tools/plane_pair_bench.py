#!/usr/bin/env python3
"""The fp16-pair NT plane products of a bert-base layer at batch 8 (M = 4128 tokens), per tile: time and fp32-equivalent TF/s.
python tools/plane_pair_bench.py [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd"))
import torch
from vbg import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda")
torch.manual_seed(0)
SHAPES = [("QKV fwd", 4128, 2304, 768), ("attn-out fwd / dgrad", 4128, 768, 768), ("FFN1 fwd / FFN2 dgrad", 4128, 3072, 768),
          ("FFN2 fwd / FFN1 dgrad", 4128, 768, 3072), ("QKV dgrad", 4128, 768, 2304)]
TILES = [int(t) for t in os.environ.get("TILES", "128129,256128").split(",")]
for name, M, N, K in SHAPES:
    a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev) * 0.05
    pa, pb = ops.split_planes_pair(a), ops.split_planes_pair(b)
    out = torch.empty(M, N, device=dev)
    ref = (a.double() @ b.double().t())
    line = f"{name:24s} {M}x{N}x{K}"
    for tile in TILES:
        try:
            ops.plane_gemm(pa, pb, out, form=1, tile=tile)
        except Exception as e:
            line += f" | {tile}: n/a"; continue
        err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
        for _ in range(3): ops.plane_gemm(pa, pb, out, form=1, tile=tile)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): ops.plane_gemm(pa, pb, out, form=1, tile=tile)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        line += f" | {tile}: {us:7.1f} us {2.0 * M * N * K / us / 1e6:6.1f} TF err {err:.1e}"
    print(line, flush=True)
