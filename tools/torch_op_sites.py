#!/usr/bin/env python3
"""Which call sites launch the torch kernels of a step (aten::add / zeros / fill_ / copy_ ...): torch.profiler with stacks over one
bench step, grouped by (op, input shapes, innermost frame inside this repository).   python tools/torch_op_sites.py"""
import collections, os, random, sys, tempfile, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
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
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    dt = getattr(ev, "self_device_time_total", None)
    if dt is None:
        dt = getattr(ev, "self_cuda_time_total", 0)
    if not ev.name.startswith("aten::") or dt <= 0:
        continue
    site = "?"
    for fr in (ev.stack or []):
        if "vibertgrid-pytorch_amd" in fr or "bench.py" in fr:
            site = fr.split("vibertgrid-pytorch_amd/")[-1]
            break
    k = (ev.name, str(ev.input_shapes)[:70], site[:90])
    agg[k][0] += 1; agg[k][1] += dt
tot = sum(v[1] for v in agg.values())
print(f"torch kernels: {sum(v[0] for v in agg.values())} launches, {tot / 1e3:.2f} ms device time per step")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{v[1] / 1e3:7.3f} ms x{v[0]:3d}  {k[0]:22s} {k[1]:70s} {k[2]}")
