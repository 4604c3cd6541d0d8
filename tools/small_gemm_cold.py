#!/usr/bin/env python3
"""The encoder's products for ONE document (514 token rows) with COLD weights: every launch multiplies by another copy of the weight
planes (more copies than the Infinity Cache holds), as a single-document inference call finds them.  Tiles x LDS stages x forms.
   python tools/small_gemm_cold.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd"))
import torch
from vbg import ops

dev = torch.device("cuda")
M = int(os.environ.get("M", "514"))
g = torch.Generator(device=dev).manual_seed(3)


def bench(fn, n):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for name, N, K in (("QKV", 2304, 768), ("AO", 768, 768), ("FFN1", 3072, 768), ("FFN2", 768, 3072)):
    copies = max(2, int(400e6 // (N * K * 4)) + 1)
    a = torch.randn(M, K, device=dev, generator=g)
    qa, pa = ops.split_planes_pair(a), ops.split_planes(a)
    ws = [torch.randn(N, K, device=dev, generator=g) / K ** 0.5 for _ in range(2)]
    qws = [ops.split_planes_pair(ws[i % 2]) for i in range(copies)]
    pws = [ops.split_planes(ws[i % 2]) for i in range(min(copies, 40))]
    out = torch.empty(M, N, device=dev)
    bias = torch.randn(N, device=dev, generator=g)
    ref = None
    line = [f"{name:5s} [{M} x {N} x {K}] {copies} weight copies:"]
    for form, tiles in ((1, (64064, 64004, 128129)), (0, (64064, 128129))):
        for tile in tiles:
            ops_w = qws if form else pws
            A = qa if form else pa
            try:
                hot = bench(lambda i: ops.plane_gemm(A, ops_w[0], out, bias=bias, tile=tile, form=form), 50)
                cold = bench(lambda i: ops.plane_gemm(A, ops_w[i % len(ops_w)], out, bias=bias, tile=tile, form=form), 3 * len(ops_w))
            except Exception as e:
                line.append(f"   form {form} tile {tile}: {type(e).__name__}")
                continue
            ops.plane_gemm(A, ops_w[0], out, bias=bias, tile=tile, form=form)
            if form and ref is None:
                ref = out.clone()
            same = (torch.equal(out, ref) if form else bool(((out - ref).abs().max() <= 2e-6 * ref.abs().max()))) if ref is not None else None
            line.append(f"   form {form} tile {tile:6d}: hot {hot:6.1f} us  cold {cold:6.1f} us   {'== 64064 pair result' if (form and same) else ('close' if same else 'DIFFERS')}")
    print("\n".join(line), flush=True)
