#!/usr/bin/env python3
"""Per-call-site timing of the plane products inside one bench step: HIP events around every vbg_plane_gemm launch, aggregated by
(form, trans, M, N, K, tile, epilogue extras).  python tools/step_plane_profile.py"""
import os, sys, tempfile, collections, contextlib
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
for _ in range(3): step()
recs = []
orig, orig_g = ops.plane_gemm, ops.plane_gemm_grouped
def wrapped(a, b, out=None, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig(a, b, out, **kw); e1.record()
    tr = bool(kw.get("trans"))
    M, N, K = (a.cols, b.cols, a.rows) if tr else (a.rows, b.rows, a.cols)
    extra = "+".join(k for k in ("bias", "C2", "out_planes", "out_pair", "colsum_out", "c_amax", "q_ref_in", "accumulate") if kw.get(k) is not None and kw.get(k) is not False)
    recs.append((("pair" if kw.get("form") else "bf16x3", "TN" if tr else "NT", M, N, K, kw.get("tile", 0), kw.get("epi", 0), extra), e0, e1, 2.0 * M * N * K))
    return r
def wrapped_g(problems, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig_g(problems, **kw); e1.record()
    fl = sum(2.0 * a.cols * b.cols * a.rows for a, b, _ in problems)
    recs.append((("pair" if kw.get("form") else "bf16x3", "TN grouped x%d" % len(problems), problems[0][0].cols, sum(b.cols for _, b, _ in problems), problems[0][0].rows, kw.get("tile", 0), 0, ""), e0, e1, fl))
    return r
ops.plane_gemm, ops.plane_gemm_grouped = wrapped, wrapped_g
step()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, e0, e1, fl in recs:
    a = agg.setdefault(key, [0, 0.0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += fl
tot = sum(v[1] for v in agg.values())
print(f"plane products: {len(recs)} launches, {tot:.2f} ms/step (event time)")
for key, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    form, kind, M, N, K, tile, epi, extra = key
    print(f"{ms:7.3f} ms  x{n:3d}  {ms / n * 1e3:7.1f} us  {fl / ms / 1e9:6.1f} TF/s  {form:6s} {kind:14s} M{M} N{N} K{K} tile{tile} epi{epi} {extra}")
