#!/usr/bin/env python3
"""Every generic-kernel product (vbg.ops.gemm_raw) of one cfg2 training step: phase, stream, shape, operand kinds, split -- which
launches the generic six-product kernel still takes, and on which stream."""
import contextlib, os, random, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd"))
import torch
import bench
from vbg import ops
from vbg.batch import PackedBatch
from vbg.optim import FusedAdamW, FusedSGD, split_parameters

dev = torch.device("cuda", 0)
with contextlib.redirect_stdout(sys.stderr):
    torch.manual_seed(42)
    net = bench.build_model(tempfile.mkdtemp()).to(dev).train()
cnn, bert = split_parameters(net)
opts = [FusedSGD(cnn, dev, lr=0.0), FusedAdamW(bert, dev, lr=0.0)]
batch = PackedBatch.pack(*bench.synthetic_batch(8, 512, 512, 512, 128, bench.NCLS, bench.VOCAB, 1234)).to(dev)
log, phase = [], ["warm"]
real = ops.gemm_raw
streams = {}
def rec(M, N, K, A, lda, a_kind, B, ldb, b_kind, Cout, ldc, **kw):
    sid = streams.setdefault(ops.raw_stream(dev), len(streams))
    log.append((phase[0], sid, int(M), int(N), int(K), a_kind, b_kind, kw.get("splitk", 1), bool(kw.get("accumulate", False)), "segs" if kw.get("segs") else "", "geo" if kw.get("geo") is not None else ""))
    return real(M, N, K, A, lda, a_kind, B, ldb, b_kind, Cout, ldc, **kw)
ops.gemm_raw = rec
import vbg.functions as Fn
for step in range(2):
    log.clear()
    for o in opts:
        o.zero_grad()
    phase[0] = "fwd"
    loss = net(*batch)
    phase[0] = "bwd"
    loss.backward()
    torch.cuda.synchronize()
kinds = {0: "DENSE_K", 1: "DENSE_R", 2: "CONV_K", 3: "CONV_R", 4: "WT_R"}
print("phase stream      M      N      K   A-kind   B-kind  splitk acc  GF")
for ph, sid, M, N, K, ak, bk, sk, acc, segs, geo in log:
    print(f"{ph:4s}  s{sid}  {M:7d} {N:6d} {K:7d}  {kinds.get(ak, ak):8s} {kinds.get(bk, bk):8s} {sk:3d}  {int(acc)}  {2.0 * M * N * K / 1e9:7.2f}  {segs} {geo}")
