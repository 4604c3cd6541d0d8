#!/usr/bin/env python3
"""Time the three fused-attention kernels at the cfg2 shape (8 sequences of 512 + 8 of 4 tokens, 12 heads, dropout 0.1)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from vbg import ops  # noqa: E402
from vbg.lib import ATTN_DKV, ATTN_DQ, ATTN_FWD  # noqa: E402
from plane_gemm_bench import timed  # noqa: E402
from test_gpu_attention import _meta  # noqa: E402
dev = torch.device("cuda")
heads = 12
meta, *_ = _meta([512] * 8 + [4] * 8, heads)
hid, ntok = heads * 64, meta.ntok
g = torch.Generator().manual_seed(0)
xq, xdo = torch.randn(ntok, 3 * hid, generator=g).to(dev), (torch.randn(ntok, hid, generator=g) * 1e-6).to(dev)
for form, p in ((0, 0.1), (1, 0.1), (2, 0.1), (0, 0.0), (1, 0.0)):
    ops.set_amp(form == 2)
    slot = ops.amax(xdo) if form else None
    pq = ops.split_planes_pair(xq) if form else ops.split_planes(xq)
    pdo = ops.split_planes_pair(xdo, amax_slot_=slot) if form else ops.split_planes(xdo)
    masks = ops.attn_mask(meta, p, 1, 2) if p > 0 else None
    O, kbar = torch.zeros(ntok, hid, device=dev), torch.zeros(ntok, hid, device=dev)
    lse = torch.zeros(2, heads, meta.ntok_pad, device=dev)
    opl = ops.planes_empty(ntok, hid, dev)
    dqkv = torch.zeros(ntok, 3 * hid, device=dev)
    f = lambda: ops.attn(meta, ATTN_FWD, pq, None, O, lse, None, masks, 0.125, p, kbar=kbar, out_planes=opl)
    f()
    delta = torch.zeros_like(lse[0])
    dq = lambda: ops.attn(meta, ATTN_DQ, pq, pdo, dqkv, lse, delta, masks, 0.125, p, kbar=kbar, o=O, do_amax=slot)
    dkv = lambda: ops.attn(meta, ATTN_DKV, pq, pdo, dqkv, lse, delta, masks, 0.125, p, do_amax=slot)
    tf, tq, tk = timed(f), timed(dq), timed(dkv)
    fl = 8 * heads * 4.0 * 512 * 512 * 64
    print(f"form {form} ({('three bf16 pieces, six products', 'two fp16 pieces, three products', 'hi pieces, one product (amp)')[form]}) dropout {p}: FWD {tf:6.1f} us ({fl / tf * 1e-6:5.1f} TF)  DQ {tq:6.1f} us ({1.5 * fl / tq * 1e-6:5.1f} TF executed)  DKV {tk:6.1f} us ({2 * fl / tk * 1e-6:5.1f} TF algorithmic)")
ops.set_amp(False)
