#!/usr/bin/env python3
"""Per-shape timing of the row-reuse convolution launches (csrc/conv3.hip) inside one bench step: HIP events around every
ops.conv3x3 / ops.conv3x3_wgrad call, aggregated by shape.   python tools/step_conv3_profile.py"""
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
    loss = net(*db); oc.zero_grad(); ob.zero_grad(); loss.backward(); oc.step(); ob.step()
for _ in range(3): step()
recs = []
o1, o2 = ops.conv3x3, ops.conv3x3_wgrad
def w1(x, w, *a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = o1(x, w, *a, **k); e1.record()
    B, H, W, C = x.shape
    kind = "dgrad" if (k.get("n_out") is not None or (len(a) > 3 and a[3])) else "fwd"          # (the PW input gradient passes n_out)
    recs.append((f"{kind:9s} B{B} {H}x{W} {C}->{w.shape[0]} f16={int(bool(k.get('f16x2')))}", 2.0 * B * H * W * C * w.shape[0] * 9, e0, e1))
    return r
def w2(dy, x, dw, *a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = o2(dy, x, dw, *a, **k); e1.record()
    B, H, W, C = x.shape
    recs.append((f"wgrad     B{B} {H}x{W} {C}->{dy.shape[3]} f16={int(bool(k.get('f16x2')))}", 2.0 * B * H * W * C * dy.shape[3] * 9, e0, e1))
    return r
ops.conv3x3, ops.conv3x3_wgrad = w1, w2
N = 3
for _ in range(N): step()
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for name, fl, e0, e1 in recs:
    a = agg[name]; a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += fl
tot = sum(a[1] for a in agg.values()) / N
print(f"row-reuse convolution launches: {tot:.2f} ms per step (event time)")
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {a[1] / N:6.3f} ms  x{a[0] / N:4.0f}  {a[1] / a[0] * 1e3:7.1f} us  {a[2] / a[1] / 1e9:6.1f} TF/s   {name}")
