#!/usr/bin/env python3
"""What the BatchNorm-statistics epilogue of the pre-split-filter forward convolutions costs: plain store / + bias / + statistics (per-column
sum and sum of squares, fp64 atomics into slot rows) on the cfg2 shapes.  python tools/conv3_epi_bench.py [reps]"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd"))
import torch
from vbg import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda")
torch.manual_seed(0)
def t(fn):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (B, H, W, C, N) in ((8, 128, 128, 256, 256), (8, 128, 128, 64, 64), (8, 64, 64, 128, 128), (8, 32, 32, 256, 256), (8, 16, 16, 512, 512), (1024, 7, 7, 256, 256)):
    x = torch.randn(B, H, W, C, device=dev)
    wd = (torch.randn(N, C, 3, 3, device=dev) / math.sqrt(9 * C)).contiguous(memory_format=torch.channels_last)
    w4 = wd.permute(0, 2, 3, 1)
    late = ops.conv3_late_choice(B, H, W, C, N)
    bn, nz = (late if late is not None else (0, None))
    wp = ops.conv3_planes(wd, w4, False, bn=bn)
    out = torch.empty(B, H, W, N, device=dev)
    st = torch.zeros(ops.bn_slots() * 2 * N, device=dev, dtype=torch.float64)
    am = ops.amax(x)
    kw = dict(f16x2=True, w_planes=wp, nsplit=nz, bn=bn)
    a = t(lambda: ops.conv3x3(x, w4, None, out, None, **kw))
    b = t(lambda: ops.conv3x3(x, w4, None, out, st, **kw))
    c = t(lambda: ops.conv3x3(x, w4, None, out, st, x_amax=am, **kw))
    print(f"B{B} {H}x{W} {C}->{N} bn={bn or 'auto'} nsplit={nz}:  plain {a:7.1f} us   + statistics {b:7.1f} us   + statistics + x_amax {c:7.1f} us", flush=True)
