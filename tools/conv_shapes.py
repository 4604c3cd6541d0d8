#!/usr/bin/env python3
"""every convolution call of one cfg2 training step with the kernel it takes: python tools/conv_shapes.py"""
import collections
import contextlib
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd"))
sys.path.insert(0, ROOT)
import torch

import bench
from vbg import ops

dev = torch.device("cuda")
with contextlib.redirect_stdout(sys.stderr):
    net = bench.build_model(tempfile.mkdtemp(), "resnet_34_fpn_pretrained").to(dev).train()
batch = bench.synthetic_batch(8, 512, 512, 512, 128, bench.NCLS, bench.VOCAB, 1234)
from vbg.batch import PackedBatch
db = PackedBatch.pack(*batch).to(dev)
log = collections.Counter()
f0, d0, w0 = ops.conv2d_fwd, ops.conv2d_dgrad, ops.conv2d_wgrad


def fwd(x, w, stride, pad, *a, **k):
    B, H, W, Ci = x.shape
    Co, kh, kw, _ = w.shape
    log[("fwd", H, W, Ci, Co, kh, stride, "conv3" if ops.conv3_ok(B, H, W, Ci, Co, kh, kw, stride, pad) else "generic")] += 1
    return f0(x, w, stride, pad, *a, **k)


def dgrad(dy, w, xs, stride, pad, *a, **k):
    B, H, W, Ci = xs
    Co, kh, kw, _ = w.shape
    log[("dgrad", H, W, Ci, Co, kh, stride, "conv3" if ops.conv3_ok(B, H, W, Co, Ci, kh, kw, stride, pad) else "generic")] += 1
    return d0(dy, w, xs, stride, pad, *a, **k)


def wgrad(dy, x, dw, stride, pad, *a, **k):
    B, H, W, Ci = x.shape
    Co, kh, kw, _ = dw.shape
    log[("wgrad", H, W, Ci, Co, kh, stride, "conv3" if ops.conv3w_ok(B, H, W, Ci, Co, kh, kw, stride, pad) else "generic")] += 1
    return w0(dy, x, dw, stride, pad, *a, **k)


ops.conv2d_fwd, ops.conv2d_dgrad, ops.conv2d_wgrad = fwd, dgrad, wgrad
net(*db).backward()
torch.cuda.synchronize()
for k, v in sorted(log.items()):
    print(v, k)
