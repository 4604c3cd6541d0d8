#!/usr/bin/env python3
"""Where the HOST spends the step: cProfile of forward / backward enqueue at cfg2 (B=8) (tools/infer_host_profile.py does `inference()`).  The device is
drained before every phase, so what is timed is the Python + ctypes + dispatcher cost of enqueueing, not the kernels.

    python tools/host_profile.py [--steps 6] [--top 45]"""
import argparse
import contextlib
import cProfile
import io
import os
import pstats
import random
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd"))
import torch

import bench


def report(pr, title, top, steps):
    s = io.StringIO()
    st = pstats.Stats(pr, stream=s)
    st.sort_stats("tottime").print_stats(top)
    print(f"==== {title}: by own time ({steps} calls profiled) ====")
    print("\n".join(l[:200] for l in s.getvalue().splitlines()[4:]))
    s = io.StringIO()
    st = pstats.Stats(pr, stream=s)
    st.sort_stats("cumulative").print_stats(top)
    print(f"==== {title}: by cumulative time ====")
    print("\n".join(l[:200] for l in s.getvalue().splitlines()[4:]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--top", type=int, default=45)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    random.seed(42)
    with contextlib.redirect_stdout(sys.stderr):
        torch.manual_seed(42)
        model = bench.build_model(tempfile.mkdtemp(prefix="vbg_hp_")).to(dev).train()
    from vbg.optim import FusedAdamW, FusedSGD, split_parameters
    cnn, bert = split_parameters(model)
    oc = FusedSGD(cnn, dev, lr=0.005, momentum=0.9, weight_decay=0.005)
    ob = FusedAdamW(bert, dev, lr=5e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    batch = bench.synthetic_batch(8, 512, 512, 512, 128, bench.NCLS, bench.VOCAB, 1234)
    b = tuple(tuple(t.to(dev) for t in g) if isinstance(g, tuple) else g.to(dev) for g in batch)
    pf, pb = cProfile.Profile(), cProfile.Profile()
    tf = tb = 0.0
    for it in range(args.steps + 3):
        prof = it >= 3
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if prof:
            pf.enable()
        loss = model(*b)
        if prof:
            pf.disable()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        oc.zero_grad(); ob.zero_grad()
        t2 = time.perf_counter()
        if prof:
            pb.enable()
        loss.backward()
        if prof:
            pb.disable()
        t3 = time.perf_counter()
        torch.cuda.synchronize()
        oc.step(); ob.step()
        if prof:
            tf += t1 - t0; tb += t3 - t2
    print(f"host enqueue under cProfile: forward {tf / args.steps * 1e3:.2f} ms, backward {tb / args.steps * 1e3:.2f} ms per step")
    report(pf, "forward (training, cfg2 B=8)", args.top, args.steps)
    report(pb, "backward (training, cfg2 B=8)", args.top, args.steps)
    # the same without the profiler, for the true figure
    tf = tb = 0.0
    for it in range(args.steps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); loss = model(*b); t1 = time.perf_counter(); torch.cuda.synchronize()
        oc.zero_grad(); ob.zero_grad(); t2 = time.perf_counter(); loss.backward(); t3 = time.perf_counter(); torch.cuda.synchronize(); oc.step(); ob.step()
        tf += t1 - t0; tb += t3 - t2
    print(f"host enqueue without the profiler: forward {tf / args.steps * 1e3:.2f} ms, backward {tb / args.steps * 1e3:.2f} ms per step")


if __name__ == "__main__":
    main()
