#!/usr/bin/env python3
"""Per-call-site GEMM timing inside one bench step: HIP events around every vbg_gemm launch, aggregated by
(operand kinds, grouped, M, N, K, splitk).  python tools/step_gemm_profile.py"""
import os, sys, random, tempfile, collections, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd"))
import torch
import bench
from vbg import ops
from vbg.optim import FusedAdamW, FusedSGD, split_parameters

dev = torch.device("cuda")
with contextlib.redirect_stdout(sys.stderr):
    net = bench.build_model(tempfile.mkdtemp()).to(dev).train()
cnn, bert = split_parameters(net)
oc, ob = FusedSGD(cnn, dev, lr=0.005, momentum=0.9, weight_decay=0.005), FusedAdamW(bert, dev, lr=5e-5)
batch = bench.synthetic_batch(8, 512, 512, 512, 128, 5, 30522, 1234)
mv = lambda ts: tuple(t.to(dev) for t in ts)
db = (mv(batch[0]), mv(batch[1]), mv(batch[2]), mv(batch[3]), batch[4].to(dev), batch[5].to(dev))
def step():
    loss = net(*db); loss.item(); oc.zero_grad(); ob.zero_grad(); loss.backward(); oc.step(); ob.step()
for _ in range(2): step()
recs = []
orig = ops.gemm_raw
def wrapped(M, N, K, A, lda, a_kind, B, ldb, b_kind, Cout, ldc, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(M, N, K, A, lda, a_kind, B, ldb, b_kind, Cout, ldc, **kw); e1.record()
    g = kw.get("grp") is not None
    if g:
        M, N = kw["grp_max"]; K = -1
    recs.append(((a_kind, b_kind, g, M, N, K, kw.get("splitk", 1), kw.get("ngroups", 0), "seg" if kw.get("segs") else ""), e0, e1))
ops.gemm_raw = wrapped
import vbg.functions as F_
step()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, e0, e1 in recs:
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in agg.values())
print(f"total gemm ms/step (event time, serialised by events): {tot:.1f}")
names = {0: "DK", 1: "DR", 2: "CK", 3: "CR", 4: "WT"}
for key, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    a, b, g, M, N, K, sk, ng, seg = key
    fl = 2.0 * M * N * K * n if K > 0 else 0
    print(f"{ms:7.2f} ms  x{n:3d}  {names[a]}x{names[b]} {'grp'+str(ng) if g else '   '} {seg:3s} M{M} N{N} K{K} sk{sk}  {fl/ms/1e9 if ms and fl else 0:6.1f} TF/s")
