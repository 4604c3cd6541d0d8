#!/usr/bin/env python3
"""LayerNorm forward / backward kernels at the encoder's shape (4128 rows of 768; dropout 0.1): launch time and bytes over time.
VBG_LN_WROWS=n forces the rows per wave of the backward kernel."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from vbg import ops  # noqa: E402
from plane_gemm_bench import timed  # noqa: E402
dev = torch.device("cuda")
rows, hid = int(os.environ.get("ROWS", 4128)), 768
g = torch.Generator().manual_seed(0)
x, res = torch.randn(rows, hid, generator=g).to(dev), torch.randn(rows, hid, generator=g).to(dev)
gam, bet = torch.randn(hid, generator=g).to(dev), torch.randn(hid, generator=g).to(dev)
dy = (torch.randn(rows, hid, generator=g) * 1e-4).to(dev)
y, xhat, rstd = ops.dropout_add_ln_fwd(x, res, gam, bet, 1e-12, 0.1, 1, 2)[:3]
dg, db, dbias = (torch.zeros(hid, device=dev) for _ in range(3))
slot = ops.amax(dy)
s1, s2 = ops.amax_slot(dev), ops.amax_slot(dev)
MB = rows * hid * 4 / 1e6
tf = timed(lambda: ops.dropout_add_ln_fwd(x, res, gam, bet, 1e-12, 0.1, 1, 2, out_pair=ops.pair_empty(rows, hid, dev)))
tb = timed(lambda: ops.dropout_add_ln_bwd_pair(dy, xhat, rstd, gam, 0.1, 1, 2, dg, db, dbias, slot, s1, s2))
t0 = timed(lambda: ops.dropout_add_ln_bwd(dy, xhat, rstd, gam, 0.1, 1, 2, dg, db))
print(f"rows {rows} wrows {os.environ.get('VBG_LN_WROWS', 'auto')}: fwd (+pair planes) {tf:6.1f} us ({5 * MB / tf * 1e-3:4.2f} TB/s)   "
      f"bwd pair + fold {tb:6.1f} us ({4 * MB / tb * 1e-3:4.2f} TB/s)   bwd fp32 + fold {t0:6.1f} us ({4 * MB / t0 * 1e-3:4.2f} TB/s)")
