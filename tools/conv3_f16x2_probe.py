#!/usr/bin/env python3
"""PROBE: row-reuse forward convolution with two fp16 pieces per operand (three piece products) against the product's six-product bf16
form: time and error against fp64.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared tools/probes/conv3_f16x2_probe.hip
-o tools/probes/libconv3_f16x2_probe.so"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import torch.nn.functional as F
from gemm_bench import report, timeit
from vbg import ops
lib = C.CDLL(os.path.join(ROOT, "tools", "probes", "libconv3_f16x2_probe.so"))
dev = torch.device("cuda")
P = lambda t: C.c_void_p(t.data_ptr())


def f16x2(x, w):
    B, H, W, Cs = x.shape
    y = torch.empty(B, H, W, w.shape[0], device=dev)
    rc = lib.probe_conv3x3_f16x2(P(x), P(w), P(y), B, H, W, Cs, w.shape[0], C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
    return y


for (B, H, W, Ci, Co, scale) in [(2, 16, 128, 64, 128, 1.0), (2, 16, 128, 64, 128, 1e-3), (8, 128, 128, 256, 256, 1.0)]:
    x = torch.randn(B, H, W, Ci, device=dev) * scale
    w = torch.randn(Co, 3, 3, Ci, device=dev) / (3 * Ci ** 0.5)
    y6 = ops.conv3x3(x, w)
    y3 = f16x2(x, w)
    if B * H * W <= 8192:
        ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), None, 1, 1).permute(0, 2, 3, 1)
        mag = F.conv2d(x.double().abs().permute(0, 3, 1, 2), w.double().abs().permute(0, 3, 1, 2), None, 1, 1).permute(0, 2, 3, 1)
        e6, e3 = float(((y6 - ref).abs() / mag).max()), float(((y3 - ref).abs() / mag).max())
        r6, r3 = float((y6 - ref).norm() / ref.norm()), float((y3 - ref).norm() / ref.norm())
        print(f"scale {scale}: max error / sum of magnitudes: bf16x3 {e6:.2e}  f16x2 {e3:.2e};  rel-L2: bf16x3 {r6:.2e}  f16x2 {r3:.2e}", flush=True)
    fl = 2.0 * B * H * W * Ci * Co * 9
    report(f"bf16x3 (6 products) B{B} {H}x{W} {Ci}->{Co}", fl, timeit(lambda: ops.conv3x3(x, w)))
    report(f"f16x2  (3 products) B{B} {H}x{W} {Ci}->{Co}", fl, timeit(lambda: f16x2(x, w)))
