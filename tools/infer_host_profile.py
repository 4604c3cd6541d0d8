#!/usr/bin/env python3
"""cProfile of single-document ViBERTgridNet.inference (host-bound: ~6 ms per call for ~2.3 ms of device work)"""
import cProfile, contextlib, io, os, pstats, sys, tempfile
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
with torch.no_grad():
    for _ in range(5):
        net.inference(*args)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(50):
        net.inference(*args).cpu()
    pr.disable()
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
st.sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats("vibertgrid", 30)
print(s.getvalue()[:7000])
