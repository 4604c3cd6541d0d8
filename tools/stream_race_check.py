#!/usr/bin/env python3
"""Gradients of ONE cfg2 step (bench.py's model and batch, 8 documents) with the side streams on against the same step on one stream,
from identical state (same dropout seed, same host RNG): every flat gradient buffer, relative L2 and max-norm difference, and the
worst single PARAMETER (a stale operand hits one layer: a group norm can hide it), several repetitions each -- the single-stream
repetitions give the noise floor of the float atomics.  A race shows as a difference far above it.

    python tools/stream_race_check.py [--reps 4] [--amax-pool 16] [--no-reserve] [--only-default]

--amax-pool N   slots per amax pool (default 256): small pools turn over several times inside one backward, which is what a
                use-after-free of a pool needs to show
--no-reserve    the round-5 behaviour: amax slots read by the weight-gradient stream are NOT record_stream-ed (the A/B that names
                the root cause of profiles/r05_stream_race.txt line 10)
--only-default  skip the intermediate configurations: one stream vs the default streams only"""
import argparse, contextlib, os, random, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd"))
import torch
import bench


def group_report(groups, g0, g):
    """[(group rel-L2, group max, worst parameter's rel-L2, its name)] of g against g0"""
    out = []
    for grp, a, b in zip(groups, g0, g):
        worst, wname = 0.0, ""
        for n, p, off in zip(grp.names, grp.params, grp.offsets):
            x, y = a[off:off + p.numel()], b[off:off + p.numel()]
            nx = float(x.norm())
            if nx == 0.0:
                continue
            r = float((x - y).norm()) / nx
            if r > worst:
                worst, wname = r, n
        out.append((float((a - b).norm() / a.norm()), float((a - b).abs().max() / a.abs().max()), worst, wname))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--amax-pool", type=int, default=0)
    ap.add_argument("--no-reserve", action="store_true")
    ap.add_argument("--only-default", action="store_true")
    ap.add_argument("--no-encoder-stream", action="store_true", help="the conv weight-gradient stream WITHOUT the encoder's side stream")
    ap.add_argument("--cw-level", type=int, default=2, help="VBG_CONV_WGRAD_STREAM level of the streams-on runs (1: conv + BatchNorm nodes only)")
    ap.add_argument("--jitter", action="store_true", help="ONE stream throughout, but a different allocation pattern every step (random blocks held across the step, "
                    "filled with NaN and freed): a kernel that reads memory nobody wrote shows without any second stream")
    ap.add_argument("--trace", action="store_true", help="capture the intermediates of the 64-channel conv + BatchNorm backward nodes (incoming gradient, "
                    "BatchNorm slot sums, dz, its amax slot, the input gradient) in every run and name the FIRST one that leaves the one-stream run's")
    ap.add_argument("--amp", action="store_true", help="the steps inside torch.autocast(float16): the one-product forms")
    ap.add_argument("--shape", default="cfg2", choices=["cfg2", "cfg3", "cfg4", "cfg5"], help="bench.py's other shapes (cfg4: 512 segments per document -> ~4 000 RoIs on the heads' stream; cfg5: 16 x 1024 x 1024)")
    ap.add_argument("--offenders", type=float, default=0.0, help="for every run whose cnn group differs by more than this: list every parameter above it, in flat-buffer order")
    args = ap.parse_args()
    from vbg import ops
    from vbg.batch import PackedBatch
    from vbg.optim import FusedAdamW, FusedSGD, split_parameters
    if args.amax_pool:
        ops._AMAX_POOL_SLOTS[0] = args.amax_pool
    ops._RESERVE_AMAX[0] = not args.no_reserve
    dev = torch.device("cuda", 0)
    with contextlib.redirect_stdout(sys.stderr):
        torch.manual_seed(42)
        shp = {"cfg2": dict(img=512, S=128, ncls=bench.NCLS, vocab=bench.VOCAB, backbone="resnet_34_fpn_pretrained", batch=8, roberta=False),
               "cfg3": dict(img=512, S=128, ncls=4, vocab=bench.VOCAB, backbone="resnet_34_fpn", batch=8, roberta=False),
               "cfg4": dict(img=512, S=512, ncls=12, vocab=21128, backbone="resnet_34_fpn", batch=8, roberta=False),
               "cfg5": dict(img=1024, S=128, ncls=bench.NCLS, vocab=50265, backbone="resnet_34_fpn", batch=16, roberta=True)}[args.shape]
        net = bench.build_model(tempfile.mkdtemp(prefix="vbg_race_"), backbone=shp["backbone"], vocab=shp["vocab"], img=shp["img"], ncls=shp["ncls"],
                                roberta=shp["roberta"]).to(dev).train()
    cnn, bert = split_parameters(net)
    opts = [FusedSGD(cnn, dev, lr=0.0), FusedAdamW(bert, dev, lr=0.0)]
    groups = [o.group for o in opts]
    batch = PackedBatch.pack(*bench.synthetic_batch(shp["batch"], shp["img"], shp["img"], 512, shp["S"], shp["ncls"], shp["vocab"], 1234)).to(dev)
    gen = net.BERTgrid_generator

    def one(overlap, cw, bw):
        ops.set_overlap(overlap and not args.no_encoder_stream); ops._CONV_WGRAD_STREAM[0] = args.cw_level if cw else 0; ops.set_wgrad_stream(bw)
        for o in opts:
            o.zero_grad()
        gen._step_seed = 0x5EED
        random.seed(7)
        with torch.autocast("cuda", dtype=torch.float16, enabled=args.amp):
            loss = net(*batch)
        loss.backward()
        out = [o.group.gflat.clone() for o in opts]          # (enqueued right behind backward(): the join must cover it)
        torch.cuda.synchronize()
        return float(loss), out

    one(False, False, False)                                  # warm-up (flat storage, plane images)
    TR = {"on": False, "rec": []}
    if args.trace:
        _red, _app, _dg = ops.bn_bwd_reduce, ops.bn_bwd_apply_fold, ops.conv2d_dgrad

        def red(dy, y, x, mean, invstd, relu, sums=None):
            out = _red(dy, y, x, mean, invstd, relu, sums)
            if TR["on"] and x.shape[-1] == 64:
                TR["rec"] += [("bn_bwd_reduce: incoming dy", dy.clone()), ("bn_bwd_reduce: slot sums", out.clone())]
            return out

        def app(dy, y, x, mean, invstd, gamma, slots, count, relu, want_dres, dgamma, dbeta, dx_amax=None):
            dx, dres = _app(dy, y, x, mean, invstd, gamma, slots, count, relu, want_dres, dgamma, dbeta, dx_amax=dx_amax)
            if TR["on"] and x.shape[-1] == 64:
                TR["rec"] += [("bn_bwd_apply_fold: dz", dx.clone())] + ([("bn_bwd_apply_fold: dz amax slot", dx_amax.clone())] if dx_amax is not None else []) \
                    + ([("bn_bwd_apply_fold: dres", dres.clone())] if dres is not None else [])
            return dx, dres

        def dg(dy, w_ohwi, x_shape, stride, pad, out=None, accumulate=False, dy_amax=None, w_owner=None):
            o = _dg(dy, w_ohwi, x_shape, stride, pad, out=out, accumulate=accumulate, dy_amax=dy_amax, w_owner=w_owner)
            if TR["on"] and dy.shape[-1] == 64 and x_shape[-1] == 64:
                TR["rec"] += [("conv2d_dgrad: dx", o.clone())] + ([("conv2d_dgrad: amax slot as read", dy_amax.clone())] if dy_amax is not None else [])
            return o
        ops.bn_bwd_reduce, ops.bn_bwd_apply_fold, ops.conv2d_dgrad = red, app, dg

    def traced(*a):
        TR["on"], TR["rec"] = True, []
        try:
            l, g = one(*a)
        finally:
            TR["on"] = False
        rec, TR["rec"] = TR["rec"], []
        return l, g, rec

    def first_deviation(ref, rec):
        if len(ref) != len(rec):
            return f"trace lengths differ: {len(ref)} vs {len(rec)}"
        lines = []
        for i, ((n0, a), (n1, b)) in enumerate(zip(ref, rec)):
            a, b = a.double().flatten(), b.double().flatten()
            d = float((a - b).norm() / (a.norm() + 1e-300))
            if d > 1e-4 or n0 != n1:
                bad = (a - b).abs() > 1e-4 * a.abs().max()
                idx = bad.nonzero().flatten()
                lines.append(f"      trace #{i:3d} {n1:40s} rel-L2 {d:.2e}; {int(bad.sum())} of {a.numel()} elements off by > 1e-4 of the max, "
                             f"first at {int(idx[0]) if idx.numel() else -1}, last at {int(idx[-1]) if idx.numel() else -1}")
                if len(lines) >= 6:
                    break
        return "\n".join(lines) if lines else "      (no captured intermediate deviates)"

    if args.trace:
        l0, g0, ref_trace = traced(False, False, False)
    else:
        l0, g0 = one(False, False, False)
    names = ("cnn", "bert")
    print(f"amax pool slots {ops._AMAX_POOL_SLOTS[0]}, amax slots reserved for the side streams: {ops._RESERVE_AMAX[0]}", flush=True)
    worst_seen = {}

    def cmp(tag, l, g):
        parts = []
        for n, (rl2, mx, wp, wn) in zip(names, group_report(groups, g0, g)):
            parts.append(f"{n}: rel-L2 {rl2:.2e} max {mx:.2e} worst-param {wp:.2e} ({wn})")
            key = tag.rsplit("#", 1)[0]
            worst_seen[(key, n)] = max(worst_seen.get((key, n), (0.0, 0.0)), (rl2, wp))
        print(f"{tag:44s} loss diff {abs(l - l0):.1e}   " + "   ".join(parts), flush=True)
        if args.offenders > 0:
            for grp, a, b in zip(groups, g0, g):
                if float((a - b).norm() / a.norm()) <= args.offenders:
                    continue
                for n, p, off in zip(grp.names, grp.params, grp.offsets):
                    x, y = a[off:off + p.numel()], b[off:off + p.numel()]
                    nx = float(x.norm())
                    r = float((x - y).norm()) / nx if nx > 0 else 0.0
                    if r > args.offenders:
                        print(f"      offender @{off:>9d} {n:60s} rel-L2 {r:.2e}  |g| {nx:.3e}  share of the group's error {float((x - y).norm() / (a - b).norm()):.3f}", flush=True)

    for r in range(args.reps):
        cmp(f"one stream again (noise floor) #{r}", *one(False, False, False))
    if args.jitter:
        rng = random.Random(99)
        for r in range(args.reps):
            # fresh NaN-filled blocks of odd sizes, some freed right away (they are what the step's torch.empty calls get next), some held
            # across the step (they push the step's tensors to other addresses)
            junk = [torch.full((rng.randrange(1 << 18, 1 << 25),), float("nan"), device=dev) for _ in range(rng.randrange(4, 16))]
            held = junk[::2]
            del junk
            cmp(f"one stream, shifted allocations #{r}", *one(False, False, False))
            del held
        print("worst over the repetitions (group rel-L2, parameter rel-L2):")
        for (key, n), (a, b) in worst_seen.items():
            print(f"  {key:44s} {n:5s} {a:.2e} {b:.2e}")
        return
    if not args.only_default:
        for r in range(args.reps):
            cmp(f"encoder on the side stream #{r}", *one(True, False, False))
    for r in range(args.reps):
        if args.trace:
            l, g, rec = traced(True, True, False)
            cmp(f"+ conv weight gradients on their stream #{r}", l, g)
            if float((g0[0] - g[0]).norm() / g0[0].norm()) > 2e-5:
                print(first_deviation(ref_trace, rec), flush=True)
            del rec
        else:
            cmp(f"+ conv weight gradients on their stream #{r}", *one(True, True, False))
    if not args.only_default:
        for r in range(args.reps):
            cmp(f"+ encoder weight gradients on theirs #{r}", *one(True, True, True))
    print("worst over the repetitions (group rel-L2, parameter rel-L2):")
    for (key, n), (a, b) in worst_seen.items():
        print(f"  {key:44s} {n:5s} {a:.2e} {b:.2e}")


if __name__ == "__main__":
    main()
