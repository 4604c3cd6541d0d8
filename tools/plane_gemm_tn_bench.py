#!/usr/bin/env python3
"""TN plane GEMM (weight gradients from untransposed planes): accuracy vs fp64 and time vs the NT kernel on transposed planes / gemm.hip"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from vbg import ops  # noqa: E402
from plane_gemm_bench import timed  # noqa: E402
dev = torch.device("cuda")
g = torch.Generator().manual_seed(1)
for (Mt, N1, N2) in ((4128, 3072, 768), (4128, 768, 3072), (4128, 2304, 768), (4128, 768, 768), (1000, 264, 136), (77, 128, 128), (4100, 1024, 512)):
    dy = (torch.randn(Mt, N1, generator=g) * torch.exp2(torch.randint(-6, 6, (Mt, 1), generator=g).float())).to(dev)
    x = torch.randn(Mt, N2, generator=g).to(dev)
    ref = dy.double().t() @ x.double()
    scale = float((dy.double().abs().t() @ x.double().abs()).max())
    pdy, px = ops.split_planes(dy), ops.split_planes(x)
    out = torch.zeros(N1, N2, device=dev)
    row = [f"dW {N1}x{N2} over {Mt}:"]
    for tile in (128129, 128130, 256128):
        out.zero_()
        ops.plane_gemm(pdy, px, out, trans=True, tile=tile)
        err = float((out.double() - ref).abs().max()) / scale
        t = timed(lambda: ops.plane_gemm(pdy, px, out, trans=True, tile=tile))
        row.append(f"TN {tile}: {t:6.1f} us {2e-6 * Mt * N1 * N2 / t:5.0f} TF err {err:.1e}")
    acc = torch.ones(N1, N2, device=dev)
    ops.plane_gemm(pdy, px, acc, trans=True, accumulate=True)
    row.append(f"acc err {float((acc.double() - 1 - ref).abs().max()) / scale:.1e}")
    o2 = torch.zeros(N1, N2, device=dev)
    t_old = timed(lambda: ops.linear_wgrad(dy, x, o2, accumulate=True))
    row.append(f"| gemm.hip {t_old:6.1f} us")
    print(" ".join(row), flush=True)
# grouped: the four weight gradients of one encoder layer in one launch
Mt = 4128
shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
probs, refs = [], []
for (N1, N2) in shapes:
    dy, x = torch.randn(Mt, N1, generator=g).to(dev), torch.randn(Mt, N2, generator=g).to(dev)
    out = torch.zeros(N1, N2, device=dev)
    probs.append((ops.split_planes(dy), ops.split_planes(x), out))
    refs.append(dy.double().t() @ x.double())
ops.plane_gemm_grouped(probs)
for (a, b, out), ref in zip(probs, refs):
    print("grouped err", float((out.double() - ref).abs().max() / ref.abs().max()))
fl = sum(2.0 * Mt * a * b for a, b in shapes)
for tile in (0, 256128):
    for (_, _, out) in probs:
        out.zero_()
    ops.plane_gemm_grouped(probs, tile=tile)
    errs = [float((out.double() - ref).abs().max() / ref.abs().max()) for (a, b, out), ref in zip(probs, refs)]
    t = timed(lambda: ops.plane_gemm_grouped(probs, tile=tile))
    print(f"grouped 4 wgrads tile {tile}: {t:.1f} us {fl / t * 1e-6:.0f} TF, errs {errs}")
