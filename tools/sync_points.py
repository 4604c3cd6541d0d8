#!/usr/bin/env python3
"""Which calls of one training step synchronise the host with the device?  torch.cuda.set_sync_debug_mode("warn") over two steps of the
headline loop (vbg.optim + resident batch); every warning is printed once with the innermost frame of this repository that caused it.
    python tools/sync_points.py [--forced]      (--forced: one-rank RCCL group, FlatReducer + SyncBatchNorm as for N > 1)"""
import collections
import contextlib
import os
import random
import sys
import tempfile
import traceback
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd"))
import torch

import bench

forced = "--forced" in sys.argv
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
if forced:
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from vbg.batch import AsyncScalar, PackedBatch
from vbg.optim import FlatReducer, FusedAdamW, FusedSGD, split_parameters

random.seed(1)
with contextlib.redirect_stdout(sys.stderr):
    torch.manual_seed(42)
    net = bench.build_model(tempfile.mkdtemp(prefix="vbg_sync_"))
if forced:
    net = torch.nn.SyncBatchNorm.convert_sync_batchnorm(net)
net = net.to(dev).train()
cnn, bert = split_parameters(net)
oc = FusedSGD(cnn, dev, lr=0.005, momentum=0.9, weight_decay=0.005)
ob = FusedAdamW(bert, dev, lr=5e-5, weight_decay=0.01)
red = FlatReducer([oc, ob], static_graph=True, force_enable=forced)
batch = bench.synthetic_batch(8, 512, 512, 512, 128, bench.NCLS, bench.VOCAB, 1234)
dbatch = PackedBatch.pack(*batch).to(dev)


def step():
    loss = net(*dbatch)
    val = AsyncScalar(loss)
    oc.zero_grad(); ob.zero_grad()
    loss.backward()
    red.finish()
    v = val.get()
    oc.step(); ob.step()
    return v


for _ in range(3):
    step()
torch.cuda.synchronize()
seen = collections.Counter()
where = {}


def hook(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" not in str(message).lower():
        return
    st = [f for f in traceback.extract_stack() if ROOT in f.filename and "sync_points" not in f.filename]
    key = f"{os.path.relpath(st[-1].filename, ROOT)}:{st[-1].lineno} {st[-1].line}" if st else f"{filename}:{lineno}"
    seen[key] += 1
    where[key] = str(message)[:120]


warnings.showwarning = hook
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
for _ in range(2):
    step()
torch.cuda.set_sync_debug_mode("default")
print(f"synchronising calls in 2 steps ({'one-rank RCCL reducer + SyncBatchNorm' if forced else 'one process'}):")
for k, n in seen.most_common():
    print(f"  {n:4d} x  {k}\n          {where[k]}")
if not seen:
    print("  none reported")
