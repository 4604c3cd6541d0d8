#!/usr/bin/env python3
"""Do the fused-attention launches lose time to whole-workgroup rounds?  The three kernels for 4 ... 12 sequences of 512 tokens at 12 heads:
48 ... 144 (sequence, head) pairs x 4 workgroups of 128 own rows = 192 ... 576 workgroups on 256 CUs x 2 resident."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from vbg import ops  # noqa: E402
from vbg.lib import ATTN_DKV, ATTN_DQ, ATTN_FWD  # noqa: E402
from plane_gemm_bench import timed  # noqa: E402
from test_gpu_attention import _meta  # noqa: E402
dev = torch.device("cuda")
heads, p = 12, 0.1
for nseq in (4, 5, 6, 7, 8, 9, 10, 11, 12):
    meta, *_ = _meta([512] * nseq, heads)
    hid, ntok = heads * 64, meta.ntok
    g = torch.Generator().manual_seed(0)
    pq = ops.split_planes(torch.randn(ntok, 3 * hid, generator=g).to(dev))
    pdo = ops.split_planes(torch.randn(ntok, hid, generator=g).to(dev))
    masks = ops.attn_mask(meta, p, 1, 2)
    O, kbar = torch.zeros(ntok, hid, device=dev), torch.zeros(ntok, hid, device=dev)
    lse = torch.zeros(2, heads, meta.ntok_pad, device=dev)
    opl = ops.planes_empty(ntok, hid, dev)
    dqkv = torch.zeros(ntok, 3 * hid, device=dev)
    f = lambda: ops.attn(meta, ATTN_FWD, pq, None, O, lse, None, masks, 0.125, p, kbar=kbar, out_planes=opl)
    f()
    delta = torch.zeros_like(lse[0])
    dq = lambda: ops.attn(meta, ATTN_DQ, pq, pdo, dqkv, lse, delta, masks, 0.125, p, kbar=kbar, o=O)
    dkv = lambda: ops.attn(meta, ATTN_DKV, pq, pdo, dqkv, lse, delta, masks, 0.125, p)
    tf, tq, tk = timed(f), timed(dq), timed(dkv)
    wg = nseq * heads * 4
    print(f"{nseq:2d} sequences = {wg:3d} workgroups ({wg / 256:4.2f} per CU):  FWD {tf:6.1f} us  DQ {tq:6.1f} us  DKV {tk:6.1f} us   per sequence {tf / nseq:5.2f} / {tq / nseq:5.2f} / {tk / nseq:5.2f}", flush=True)
