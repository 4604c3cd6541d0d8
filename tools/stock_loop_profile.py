#!/usr/bin/env python3
"""Where the reference's loop (bench.py stock_loop_leg: pipeline/train_val_utils.py:248-287 verbatim around the drop-in model) spends its
step, phase by phase, against the same phases of the headline loop (vbg.optim + resident batch).  The stock loop synchronises at the end of
every phase by itself (`.item()`, `train_loss > 10`, the next step's pageable `.to(device)`), so wall-clock per phase is what each costs;
the headline loop gets a synchronize() per phase here, which it does not have in bench.py (its step is shorter than the sum printed).

    python tools/stock_loop_profile.py [--steps 10] [--optim torch|fused]"""
import argparse
import contextlib
import os
import random
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd"))
import torch

import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--optim", default="torch", choices=["torch", "fused"])
    ap.add_argument("--resident", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    random.seed(42)
    with contextlib.redirect_stdout(sys.stderr):
        torch.manual_seed(42)
        model = bench.build_model(tempfile.mkdtemp(prefix="vbg_slp_")).to(dev).train()
    batch = bench.synthetic_batch(8, 512, 512, 512, 128, bench.NCLS, bench.VOCAB, 1234)
    params_cnn = [p for n, p in model.named_parameters() if "bert_model" not in n and p.requires_grad]
    params_bert = [p for n, p in model.named_parameters() if "bert_model" in n and p.requires_grad]
    if args.optim == "torch":
        oc = torch.optim.SGD(params=params_cnn, lr=0.005, momentum=0.9, weight_decay=0.005)
        ob = torch.optim.AdamW(params=params_bert, lr=5e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    else:
        from vbg.optim import FusedAdamW, FusedSGD, split_parameters
        cnn, bert = split_parameters(model)
        oc = FusedSGD(cnn, dev, lr=0.005, momentum=0.9, weight_decay=0.005)
        ob = FusedAdamW(bert, dev, lr=5e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    on_dev = tuple(tuple(t.to(dev) for t in g) if isinstance(g, tuple) else g.to(dev) for g in batch)
    acc = {}

    def lap(name, t0):
        t1 = time.perf_counter()
        acc[name] = acc.get(name, 0.0) + (t1 - t0)
        return t1

    for it in range(args.steps + 3):
        if it == 3:
            acc.clear()
        torch.cuda.synchronize()
        t = time.perf_counter()
        if args.resident:
            b = on_dev
        else:
            b = (tuple(x.to(dev) for x in batch[0]), tuple(x.to(dev) for x in batch[1]), tuple(x.to(dev) for x in batch[2]),
                 tuple(x.to(dev) for x in batch[3]), batch[4].to(dev), batch[5].to(dev))
        t = lap("h2d (34 pageable .to(device))", t)
        loss = model(*b)
        t = lap("forward: host enqueue", t)
        v = loss.item()
        t = lap("forward: wait in .item()", t)
        oc.zero_grad()
        ob.zero_grad()
        t = lap("zero_grad x2", t)
        loss.backward()
        t = lap("backward: host enqueue", t)
        big = bool(loss > 10)
        t = lap("backward: wait in `loss > 10`", t)
        oc.step()
        t = lap("optimizer_cnn.step(): host", t)
        ob.step()
        t = lap("optimizer_bert.step(): host", t)
        torch.cuda.synchronize()
        t = lap("optimizers: wait for the device", t)
    tot = sum(acc.values())
    print(f"optim={args.optim} resident={args.resident}: {1e3 * tot / args.steps:.2f} ms per step, {8 * args.steps / tot:.1f} docs/s (phases synchronised)")
    for k, v in acc.items():
        print(f"  {k:42s} {1e3 * v / args.steps:7.2f} ms")


if __name__ == "__main__":
    main()
