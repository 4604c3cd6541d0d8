#!/usr/bin/env python3
"""LayerNorm backward of a BERT layer at the benchmark's shape (4128 x 768, dropout 0.1): fp32 + amax form and the bound-scaled
pair-plane form; time and algorithmic bytes / time.  python tools/ln_bwd_bench.py [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd"))
import torch
from vbg import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda")
torch.manual_seed(0)
rows, hid = 4128, 768
dy, xhat = torch.randn(rows, hid, device=dev) * 1e-4, torch.randn(rows, hid, device=dev)
rstd = torch.rand(rows, device=dev) + 0.5
gamma = torch.rand(hid, device=dev) + 0.5
dg, db, dbias = (torch.zeros(hid, device=dev) for _ in range(3))
def t(fn):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
mb = rows * hid * 4 * 4 / 1e6
s_dy = ops.amax(dy)
us = t(lambda: ops.dropout_add_ln_bwd_pair(dy, xhat, rstd, gamma, 0.1, 1, 2, dg, db, dbias, s_dy, ops.amax_slot(dev), ops.amax_slot(dev)))
print(f"pair planes by bound (incl. fold launch): {us:6.1f} us  {mb / us / 1e6 * 1e6:.2f} TB/s  checksum {float(dg.double().sum()):.6e}")
us = t(lambda: ops.dropout_add_ln_bwd(dy, xhat, rstd, gamma, 0.1, 1, 2, dg, db, dx_amax=ops.amax_slot(dev)))
print(f"fp32 dx + amax        (incl. fold launch): {us:6.1f} us  {mb / us / 1e6 * 1e6:.2f} TB/s")
