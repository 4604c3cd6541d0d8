"""bn_apply / bn_bwd_apply with and without the amax output (device word with the bits of max |out|)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vibertgrid-pytorch_amd"))
from vbg import ops
dev = torch.device("cuda")
def t(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for (M, C) in [(131072, 256), (131072, 64), (32768, 128), (8192, 256), (2048, 512)]:
    x = torch.randn(M, C, device=dev); dy = torch.randn(M, C, device=dev) * 1e-6; y = torch.relu(x)
    st = ops.bn_stats(x); mean, invstd = ops.bn_finalize(st, C, ops.bn_slots(), M, 1e-5, 0.1, None, None)
    gam = torch.ones(C, device=dev); bet = torch.zeros(C, device=dev)
    sums = torch.zeros(2 * C, device=dev, dtype=torch.float64)
    slot = ops.amax_slot(dev)
    a0 = t(lambda: ops.bn_apply(x, None, mean, invstd, gam, bet, True))
    a1 = t(lambda: ops.bn_apply(x, None, mean, invstd, gam, bet, True, y_amax=slot))
    b0 = t(lambda: ops.bn_bwd_apply(dy, y, x, mean, invstd, gam, sums, M, True, False, None, None))
    b1 = t(lambda: ops.bn_bwd_apply(dy, y, x, mean, invstd, gam, sums, M, True, False, None, None, dx_amax=slot))
    b2 = t(lambda: ops.bn_bwd_apply(dy, y, x, mean, invstd, gam, sums, M, True, True, None, None, dx_amax=slot))
    print(f"M={M} C={C}: bn_apply {a0:.1f} -> {a1:.1f} us with amax | bn_bwd_apply {b0:.1f} -> {b1:.1f} us with amax ({4*M*C*4/b1/1e6:.2f} TB/s), with dres {b2:.1f}")
