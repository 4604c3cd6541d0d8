#!/usr/bin/env python3
"""Where the host time of one ViBERTgridNet.inference call (batch 1) goes: cProfile over 200 calls, top functions by own time, and the
wall time of the call with and without the final device wait."""
import contextlib, cProfile, os, pstats, sys, tempfile, time
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
    for _ in range(10):
        net.inference(*args)
    torch.cuda.synchronize()
    n = 100
    t0 = time.perf_counter()
    for _ in range(n):
        net.inference(*args)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host enqueue {1e3 * (t1 - t0) / n:.2f} ms per call; device behind by {1e3 * (t2 - t1):.2f} ms after {n} calls")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    t0 = time.perf_counter()
    for _ in range(n):
        p = net.inference(*args); p.cpu()
    print(f"with the caller's .cpu(): {1e3 * (time.perf_counter() - t0) / n:.2f} ms per call")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        p = net.inference(*args); p.cpu()
    pr.disable()
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats("tottime").print_stats(45)
