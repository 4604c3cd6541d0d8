#!/usr/bin/env python3
"""Host cost of the encoder layer forward, per-launch path vs the one-call entry (batch-1 inference shapes): wall time of the Python
wrapper, of the ctypes call alone, and of a bare small launch."""
import contextlib, ctypes as C, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd"))
import torch
import bench
from vbg import ops, functions as Fn
from vbg.lib import lib

dev = torch.device("cuda")
with contextlib.redirect_stdout(sys.stderr):
    net = bench.build_model(tempfile.mkdtemp()).to(dev).eval()
B = int(os.environ.get("B", "1"))
batch = bench.synthetic_batch(B, 512, 512, 512, 128, 5, 30522, 7)
mv = lambda ts: tuple(t.to(dev) for t in ts)
args = (mv(batch[0]), mv(batch[1]), mv(batch[3]), batch[4].to(dev), batch[5].to(dev))

# a bare launch
x = torch.ones(64, device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000):
    lib.vbg_scale_inplace(C.c_void_p(x.data_ptr()), 64, 1.0, ops._stream())
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"bare ctypes launch (vbg_scale_inplace, 64 elements): {1e6 * (t1 - t0) / 2000:.2f} us per call")

acc = {}
def timed(obj, name, key):
    f = getattr(obj, name)
    def w(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        e = acc.setdefault(key, [0, 0.0]); e[0] += 1; e[1] += time.perf_counter() - t
        return r
    setattr(obj, name, w)
    return f

for entry in (0, 1):
    ops.set_layer_entry(bool(entry))
    with torch.no_grad():
        for _ in range(10):
            net.inference(*args)
        torch.cuda.synchronize()
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            net.inference(*args).cpu()
        dt = (time.perf_counter() - t0) / n
    print(f"entry={entry}: inference batch {B}: {1e3 * dt:.3f} ms per call")

# inside: the layer function, the wrapper, the C call
orig_apply = Fn.BertLayerFn.forward
for entry in (0, 1):
    ops.set_layer_entry(bool(entry))
    acc.clear()
    o1 = timed(ops, "bert_layer_fwd", "ops.bert_layer_fwd (wrapper + C call)")
    o2 = timed(ops, "plane_gemm", "ops.plane_gemm")
    o3 = timed(ops, "attn", "ops.attn")
    o4 = timed(ops, "dropout_add_ln_fwd", "ops.dropout_add_ln_fwd")
    o5 = timed(ops, "weight_planes", "ops.weight_planes")
    o6 = timed(ops, "stacked_qkv", "ops.stacked_qkv")
    real = lib.vbg_bert_layer_fwd
    def cwrap(*a):
        t = time.perf_counter(); r = real(*a); e = acc.setdefault("lib.vbg_bert_layer_fwd (C call alone)", [0, 0.0]); e[0] += 1; e[1] += time.perf_counter() - t; return r
    lib.vbg_bert_layer_fwd = cwrap
    f0 = Fn.BertLayerFn.forward
    def fwd(*a, **k):
        t = time.perf_counter(); r = f0(*a, **k); e = acc.setdefault("BertLayerFn.forward", [0, 0.0]); e[0] += 1; e[1] += time.perf_counter() - t; return r
    Fn.BertLayerFn.forward = staticmethod(fwd)
    with torch.no_grad():
        for _ in range(100):
            net.inference(*args).cpu()
    Fn.BertLayerFn.forward = staticmethod(f0)
    lib.vbg_bert_layer_fwd = real
    ops.bert_layer_fwd, ops.plane_gemm, ops.attn, ops.dropout_add_ln_fwd, ops.weight_planes, ops.stacked_qkv = o1, o2, o3, o4, o5, o6
    print(f"entry={entry}:")
    for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f"   {k:48s} {c / 100:6.1f} calls per inference, {1e6 * t / c:7.2f} us each, {1e3 * t / 100:6.3f} ms per inference")
