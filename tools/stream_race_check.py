#!/usr/bin/env python3
"""Gradients of ONE cfg2 step (bench.py's model and batch, 8 documents) with the side streams on against the same step on one stream,
from identical state (same dropout seed, same host RNG): every flat gradient buffer, relative L2 and max-norm difference, several
repetitions each -- the single-stream repetitions give the noise floor of the float atomics.  A race shows as a difference far above it.

    python tools/stream_race_check.py [--reps 4]"""
import argparse, contextlib, os, random, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd"))
import torch
import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=4)
    args = ap.parse_args()
    from vbg import ops
    from vbg.batch import PackedBatch
    from vbg.optim import FusedAdamW, FusedSGD, split_parameters
    dev = torch.device("cuda", 0)
    with contextlib.redirect_stdout(sys.stderr):
        torch.manual_seed(42)
        net = bench.build_model(tempfile.mkdtemp(prefix="vbg_race_")).to(dev).train()
    cnn, bert = split_parameters(net)
    opts = [FusedSGD(cnn, dev, lr=0.0), FusedAdamW(bert, dev, lr=0.0)]
    batch = PackedBatch.pack(*bench.synthetic_batch(8, 512, 512, 512, 128, bench.NCLS, bench.VOCAB, 1234)).to(dev)
    gen = net.BERTgrid_generator

    def one(overlap, cw, bw):
        ops.set_overlap(overlap); ops._CONV_WGRAD_STREAM[0] = 2 if cw else 0; ops.set_wgrad_stream(bw)
        for o in opts:
            o.zero_grad()
        gen._step_seed = 0x5EED
        random.seed(7)
        loss = net(*batch)
        loss.backward()
        out = [o.group.gflat.clone() for o in opts]          # (enqueued right behind backward(): the join must cover it)
        torch.cuda.synchronize()
        return float(loss), out

    one(False, False, False)                                  # warm-up (flat storage, plane images)
    l0, g0 = one(False, False, False)
    names = ("cnn group", "bert group")

    def cmp(tag, l, g):
        parts = []
        for n, a, b in zip(names, g0, g):
            parts.append(f"{n}: rel-L2 {float((a - b).norm() / a.norm()):.2e} max {float((a - b).abs().max() / a.abs().max()):.2e}")
        print(f"{tag:44s} loss diff {abs(l - l0):.1e}   " + "   ".join(parts), flush=True)

    for r in range(args.reps):
        cmp(f"one stream again (noise floor) #{r}", *one(False, False, False))
    for r in range(args.reps):
        cmp(f"encoder on the side stream #{r}", *one(True, False, False))
    for r in range(args.reps):
        cmp(f"+ conv weight gradients on their stream #{r}", *one(True, True, False))
    for r in range(args.reps):
        cmp(f"+ encoder weight gradients on theirs #{r}", *one(True, True, True))


if __name__ == "__main__":
    main()
