#!/usr/bin/env python3
"""When, inside the backward pass of the cfg2 step, does every gradient bucket of vbg.optim.FlatReducer become complete?
One process, dry-run reducer (same buckets / hooks / launch sequence, a timestamp instead of the collective).
    python tools/bucket_timeline.py [bucket_mb=32]
Prints per bucket: size, the time since backward started at which its last gradient was enqueued (GPU timeline), the bytes
outstanding from then on, and the xGMI time budget that is left under the rest of backward."""
import os, sys, random, tempfile, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd"))
import torch
import bench as B
from vbg.batch import PackedBatch
from vbg.optim import FlatReducer, FusedAdamW, FusedSGD, split_parameters

mb = float(sys.argv[1]) if len(sys.argv) > 1 else 32.0
dev = torch.device("cuda", 0)
torch.manual_seed(42); random.seed(42)
with contextlib.redirect_stdout(sys.stderr):
    net = B.build_model(tempfile.mkdtemp(prefix="vbg_tl_")).to(dev).train()
cnn, bert = split_parameters(net)
opts = [FusedSGD(cnn, dev, lr=0.005, momentum=0.9, weight_decay=0.005), FusedAdamW(bert, dev, lr=5e-5, weight_decay=0.01)]
red = FlatReducer(opts, bucket_mb=mb, dry_run=True)
dbatch = PackedBatch.pack(*B.synthetic_batch(8, 512, 512, 512, 128, B.NCLS, B.VOCAB, 1234)).to(dev)
for step in range(6):
    loss = net(*dbatch)
    for o in opts:
        o.zero_grad()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    loss.backward()
    red.finish()
    e1.record()
    for o in opts:
        o.step()
    torch.cuda.synchronize()
    tl = red.timeline(e0)
total = sum(m for _, m, _ in tl)
bwd = e0.elapsed_time(e1)
print(f"backward {bwd:.2f} ms on the GPU timeline; {len(tl)} buckets, {total:.1f} MB of gradients ({mb:.0f} MB buckets, reverse-forward layout)")
print("bucket   MB    complete at   gradients still to come   ring all-reduce of everything complete so far could have used")
done = 0.0
for i, m, t in tl:
    done += m
    print(f"  {i:3d} {m:6.1f}   {t:7.2f} ms   {total - done:8.1f} MB            {done:7.1f} MB in {t:6.2f} ms")
last = tl[-1][2]
# 8 GPUs, ring all-reduce: 2 (N-1)/N x bytes over the slowest link direction; per-link ~153 GB/s peak, ~70 % achievable
for bw in (153.0, 107.0):
    tail = 0.0
    t_free = 0.0
    for i, m, t in tl:
        start = max(t, t_free)
        t_free = start + 2 * 7 / 8 * m * 1e6 / (bw * 1e9) * 1e3
    print(f"ring over one {bw:.0f} GB/s link direction: buckets sent as they complete, the last one leaves at {t_free:.2f} ms "
          f"-> {max(0.0, t_free - bwd):.2f} ms exposed behind a {bwd:.2f} ms backward")
