#!/usr/bin/env python3
"""N single-document inferences (for rocprofv3 --kernel-trace --stats): python tools/infer_one.py [n]"""
import contextlib, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd"))
import torch
import bench
dev = torch.device("cuda")
with contextlib.redirect_stdout(sys.stderr):
    net = bench.build_model(tempfile.mkdtemp()).to(dev).eval()
batch = bench.synthetic_batch(1, 512, 512, 512, 128, 5, 30522, 7)
mv = lambda ts: tuple(t.to(dev) for t in ts)
args = (mv(batch[0]), mv(batch[1]), mv(batch[3]), batch[4].to(dev), batch[5].to(dev))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
with torch.no_grad():
    for _ in range(n):
        net.inference(*args).cpu()
torch.cuda.synchronize()
