#!/usr/bin/env python3
"""What the epilogues of the two widest fp16-pair NT products of a BERT layer cost (M = 4128, N = 3072, K = 768, 256 x 128 tiles):
plain store / FFN1 forward (bias + GELU, h fp32 + gelu(h) pair planes) / FFN2 data gradient (x gelu'(h), bound-scaled pair planes +
column sums, no fp32 store).  python tools/plane_epi_bench.py [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd"))
import torch
from vbg import ops
from vbg.lib import EPI_GELU_DUAL, EPI_MUL_GELU_GRAD

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda")
torch.manual_seed(0)
M, N, K = 4128, 3072, 768
a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev) * 0.05
bias = torch.randn(N, device=dev)
pa, pb = ops.split_planes_pair(a), ops.split_planes_pair(b)
out, h = torch.empty(M, N, device=dev), torch.randn(M, N, device=dev)
pq = ops.pair_empty(M, N, dev)
s_in = ops.amax(a); l1 = ops.weight_col_l1max(b.t().contiguous()) if hasattr(ops, "weight_col_l1max") else None
cs = torch.zeros(N, device=dev)
def t(fn):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for tile in (256128, 128129):
    r = {}
    r["+ bias (first)"] = t(lambda: ops.plane_gemm(pa, pb, out, bias=bias, form=1, tile=tile))
    r["plain fp32 store"] = t(lambda: ops.plane_gemm(pa, pb, out, form=1, tile=tile))
    r["+ bias"] = t(lambda: ops.plane_gemm(pa, pb, out, bias=bias, form=1, tile=tile))
    r["plain fp32 store (again)"] = t(lambda: ops.plane_gemm(pa, pb, out, form=1, tile=tile))
    r["pair planes only (no fp32 store)"] = t(lambda: ops.plane_gemm(pa, pb, None, form=1, tile=tile, out_pair=pq))
    r["FFN1 fwd: bias + GELU dual, h fp32 + pair planes"] = t(lambda: ops.plane_gemm(pa, pb, out, bias=bias, epi=EPI_GELU_DUAL, out_pair=pq, form=1, tile=tile))
    r["FFN2 dgrad: x gelu'(h), measured-max pair path (fp32 store + c_amax)"] = t(lambda: ops.plane_gemm(pa, pb, out, epi=EPI_MUL_GELU_GRAD, C2=h, form=1, tile=tile, a_amax=s_in, c_amax=ops.amax_slot(dev)))
    r["FFN2 dgrad: x gelu'(h), bound-scaled pair planes + column sums"] = t(lambda: ops.plane_gemm(pa, pb, None, epi=EPI_MUL_GELU_GRAD, C2=h, form=1, tile=tile, a_amax=s_in, out_pair=pq,
                                                                                                  q_ref_in=s_in, q_l1=l1, q_mul=1.14, q_ref_out=ops.amax_slot(dev), colsum_out=cs))
    r["  ... without the column sums"] = t(lambda: ops.plane_gemm(pa, pb, None, epi=EPI_MUL_GELU_GRAD, C2=h, form=1, tile=tile, a_amax=s_in, out_pair=pq,
                                                                   q_ref_in=s_in, q_l1=l1, q_mul=1.14, q_ref_out=ops.amax_slot(dev)))
    for k, v in r.items():
        print(f"tile {tile}  {v:7.1f} us  {k}", flush=True)
