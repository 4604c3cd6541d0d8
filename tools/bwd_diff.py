#!/usr/bin/env python3
"""Which backward node is not reproducible?  Two identical training steps (same state, same seeds, lr = 0); every node of the autograd graph
gets a hook that fingerprints the gradients arriving at it and the gradients it produces (fp64 sum and sum of magnitudes, plus a
strided bit hash).  A node whose INCOMING gradients agree between the two runs bit for bit while what it PRODUCES differs is where the run-to-run
difference is made (float atomics inside a kernel show here as differences of ~1e-7 of the norm; a race or an unwritten operand as much more).

    AMP=1 MODEL=e2e|cfg2 python tools/bwd_diff.py"""
import contextlib, os, random, sys, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vibertgrid-pytorch_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import torch
from vbg import ops
from vbg.optim import FusedAdamW, FusedSGD, split_parameters

amp = os.environ.get("AMP", "1") != "0"
KEEP = os.environ.get("KEEP", "0") != "0"          # keep every gradient (small models only): relative L2 differences per node
which = os.environ.get("MODEL", "e2e")
dev = torch.device("cuda")
with contextlib.redirect_stdout(sys.stderr):
    if which == "e2e":
        from test_gpu_model import build_product, load_synth, to_dev
        from test_oracle_golden import _e2e_inputs, e2e_cfg
        cfg = e2e_cfg("resnet_18_fpn")
        net = build_product(pathlib.Path(tempfile.mkdtemp()), "resnet_18_fpn", cfg, layers=2, dropout=0.0)
        load_synth(net, cfg, 1200)
        net = net.to(dev).train()
        g = np.load(os.path.join(ROOT, "tests", "golden", "e2e.npz"))
        batch = to_dev(_e2e_inputs(g), dev)
        if os.environ.get("FORCE_PAIR", "1") != "0":
            ops.set_pair(True, force=True)
    else:
        import bench
        from vbg.batch import PackedBatch
        torch.manual_seed(42)
        net = bench.build_model(tempfile.mkdtemp()).to(dev).train()
        batch = PackedBatch.pack(*bench.synthetic_batch(8, 512, 512, 512, 128, bench.NCLS, bench.VOCAB, 1234)).to(dev)
cnn, bert = split_parameters(net)
opts = [FusedSGD(cnn, dev, lr=0.0), FusedAdamW(bert, dev, lr=0.0)]
ops.set_overlap(False); ops._CONV_WGRAD_STREAM[0] = 0; ops.set_wgrad_stream(False)          # one stream


def fp(t):
    if t is None or not torch.is_tensor(t) or not t.is_floating_point():
        return None
    d = t.detach().double()
    flat = t.detach().reshape(-1).view(torch.int32) if t.dtype == torch.float32 and t.is_contiguous() else None
    h = int(flat[::max(1, flat.numel() // 65536)].long().sum()) if flat is not None else 0
    return (tuple(t.shape), float(d.sum()), float(d.abs().sum()), h)


def one():
    for o in opts:
        o.zero_grad()
    net.BERTgrid_generator._step_seed = 41
    random.seed(7)
    with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
        loss = net(*batch)
    # every node reachable from the loss, in a fixed traversal order
    nodes, seen, stack = [], set(), [loss.grad_fn]
    while stack:
        n = stack.pop()
        if n is None or id(n) in seen:
            continue
        seen.add(id(n)); nodes.append(n)
        for nx, _ in n.next_functions:
            stack.append(nx)
    rec, order = {}, []
    for i, n in enumerate(nodes):
        def hook(gin, gout, i=i, n=n):
            rec[i] = (type(n).__name__, [fp(t) for t in gout], [fp(t) for t in gin],
                      [t.detach().clone() if (KEEP and torch.is_tensor(t) and t.is_floating_point()) else None for t in gout],
                      [t.detach().clone() if (KEEP and torch.is_tensor(t) and t.is_floating_point()) else None for t in gin])
            order.append(i)
        n.register_hook(hook)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss.detach()), rec, order, [o.group.gflat.clone() for o in opts]


one()
a = one()
b = one()
print(f"model {which}, amp {amp}: losses {a[0]!r} {b[0]!r}; flat gradients rel-L2 " + " ".join(f"{float((x - y).norm() / x.norm()):.1e}" for x, y in zip(a[3], b[3])))
sa, sb = [a[1][i] for i in a[2]], [b[1][i] for i in b[2]]
if KEEP:
    def rel(xs, ys):
        w = 0.0
        for x, y in zip(xs, ys):
            if x is not None and y is not None and x.shape == y.shape and float(x.norm()) > 0:
                w = max(w, float((x.double() - y.double()).norm() / x.double().norm()))
        return w
    print("execution index, node, rel-L2 difference of the incoming gradients, of the produced gradients, produced shapes")
    for k, (ra, rb) in enumerate(zip(sa, sb)):
        if ra[0] == rb[0] and ra[0] != "AccumulateGrad":
            ri, ro = rel(ra[3], rb[3]), rel(ra[4], rb[4])
            if ri > 0 or ro > 0:
                print(f"   {k:5d} {ra[0]:34s} in {ri:.1e}  out {ro:.1e}  {[tuple(t.shape) for t in ra[4] if t is not None]}")
sa, sb = [r[:3] for r in sa], [r[:3] for r in sb]
ka, kb = [(r[0], [x and x[0] for x in r[2]]) for r in sa], [(r[0], [x and x[0] for x in r[2]]) for r in sb]
if ka != kb:
    d = next((i for i in range(min(len(ka), len(kb))) if ka[i] != kb[i]), min(len(ka), len(kb)))
    print("tail of the longer:", (ka if len(ka) > len(kb) else kb)[min(len(ka), len(kb)):][:6])
    print(f"the two runs executed different node sequences ({len(ka)} vs {len(kb)} nodes); first difference at position {d}:")
    for i in range(max(0, d - 4), min(len(ka), len(kb), d + 8)):
        print(f"   {i:5d}  {str(ka[i])[:100]:100s} | {str(kb[i])[:100]}")
    # match what can be matched: by (name, shapes) multiset order
    from collections import defaultdict
    pos = defaultdict(list)
    for r in sb:
        pos[(r[0], str([x and x[0] for x in r[2]]), str([x and x[0] for x in r[1]]))].append(r)
    sb2 = []
    for r in sa:
        lst = pos[(r[0], str([x and x[0] for x in r[2]]), str([x and x[0] for x in r[1]]))]
        sb2.append(lst.pop(0) if lst else (r[0], None, None))
    sb = sb2
made, first = [], None
for k, ((na, ia, pa), (nb, ib, pb)) in enumerate(zip(sa, sb)):
    if ib is None:
        continue
    same_in, same_out = ia == ib, pa == pb
    if not same_out and first is None:
        first = k
    if same_in and not same_out:
        rel = max((abs(x[2] - y[2]) / max(abs(x[2]), 1e-30)) for x, y in zip(pa, pb) if x is not None and y is not None and x != y)
        made.append((k, na, rel, [x[0] for x in pa if x is not None]))
print(f"{len(a[2])} nodes ran; the first whose produced gradients differ is number {first} in execution order")
print("nodes with identical incoming gradients and different produced gradients (execution index, node, relative difference of sum |g|, shapes):")
for k, n, rel, shapes in made[:40]:
    print(f"   {k:5d}  {n:36s} {rel:.1e}  {shapes}")
