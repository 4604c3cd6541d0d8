#!/usr/bin/env python3
"""Are the gradients of an `amp: True` step reproducible run to run?  The e2e fixture model (resnet-18 FPN, 2 encoder layers), N identical
steps (lr = 0) inside torch.autocast; pairwise relative L2 differences of the flat gradients per group, and the tensors that differ."""
import os, random, sys, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
from test_gpu_model import build_product, load_synth, to_dev
from test_oracle_golden import _e2e_inputs, e2e_cfg
from vbg import ops
from vbg.optim import FusedAdamW, FusedSGD, split_parameters

amp = os.environ.get("AMP", "1") != "0"
dev = torch.device("cuda")
cfg = e2e_cfg("resnet_18_fpn")
net = build_product(pathlib.Path(tempfile.mkdtemp()), "resnet_18_fpn", cfg, layers=2, dropout=0.0)
load_synth(net, cfg, 1200)
net = net.to(dev).train()
cnn, bert = split_parameters(net)
opts = [FusedSGD(cnn, dev, lr=0.0), FusedAdamW(bert, dev, lr=0.0)]
g = np.load(os.path.join(ROOT, "tests", "golden", "e2e.npz"))
dbatch = to_dev(_e2e_inputs(g), dev)
ops.set_overlap(os.environ.get("OVERLAP", "0") != "0")
if os.environ.get("ONE_STREAM", "0") != "0":
    ops._CONV_WGRAD_STREAM[0] = 0; ops.set_wgrad_stream(False)
ops.set_pair(True, force=True)
runs = []
for r in range(5):
    for o in opts:
        o.zero_grad()
    net.BERTgrid_generator._step_seed = 41
    random.seed(7)
    with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
        loss = net(*dbatch)
    loss.backward()
    torch.cuda.synchronize()
    runs.append((float(loss.detach()), [o.group.gflat.clone() for o in opts]))
print("losses", [r[0] for r in runs])
for gi, name in enumerate(("cnn", "bert")):
    print(name, "pairwise rel-L2:")
    for i in range(5):
        print("   ", " ".join(f"{float((runs[i][1][gi] - runs[j][1][gi]).norm() / runs[i][1][gi].norm()):.1e}" for j in range(5)))
for gi, name in enumerate(("cnn", "bert")):
    grp = opts[gi].group
    a, b = runs[0][1][gi], runs[1][1][gi]
    bad = []
    for n, p, off in zip(grp.names, grp.params, grp.offsets):
        x, y = a[off:off + p.numel()], b[off:off + p.numel()]
        nx = float(x.norm())
        if nx > 0:
            bad.append((float((x - y).norm()) / nx, n, tuple(p.shape)))
    print(f"{name}: {sum(1 for r, _, _ in bad if r > 1e-5)} of {len(bad)} tensors differ by more than 1e-5 between run 0 and run 1; in layout order:")
    for r, n, sh in bad:
        if r > 1e-5 and ("bias" in n or "head" in n or "fusion" in n or "layer.1." in n):
            print(f"   {r:.2e}  {n} {sh}")
