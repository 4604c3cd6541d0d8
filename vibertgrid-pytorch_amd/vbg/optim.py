"""Flat parameter / gradient storage, fused optimizers and the data-parallel gradient reducer.

Mirrors what the reference wires up in train_SROIE.py:206-235 (DDP(find_unused_parameters=True), SGD for the
CNN/head parameters, AdamW for every parameter whose name contains "bert_model") and steps at
pipeline/train_val_utils.py:272-284, designed for MI355X instead of translated:

* every trainable parameter of a group lives in ONE flat fp32 buffer (so do its gradient and optimizer
  state); `param.data` / `param.grad` are views, conv weights keep their channels_last (OHWI) memory;
* `zero_grad` is one memset, each optimizer step is ONE HIP launch over the flat range
  (libvbg `vbg_sgd_step` / `vbg_adamw_step`, 20 / 28 B per parameter);
* data parallel: the flat gradient buffer is cut into large contiguous buckets; a bucket is all-reduced
  (RCCL over xGMI, async, its own stream) as soon as autograd has accumulated its last gradient, so the
  exchange overlaps the rest of backward; no bucket copies (gradients ARE the bucket); the 1/world
  averaging is folded into the optimizer kernels.  Parameters that never receive a gradient
  (`bert_model.pooler.*`, `backbone.resnet.fc.*`) are kept out of the buffers, which is what
  `find_unused_parameters=True` + "skip params with grad None" amounts to in the reference.
"""
from typing import Dict, List, Tuple

import torch
import torch.distributed as dist

from . import functions as Fn
from . import ops

STATIC_UNUSED = ("pooler.", "resnet.fc.")


def _phys_view(flat: torch.Tensor, off: int, p: torch.Tensor) -> torch.Tensor:
    """view of flat[off:off+numel] with p's logical shape and p's memory layout (channels_last kept)"""
    n = p.numel()
    chunk = flat[off:off + n]
    if p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last) and not p.is_contiguous():
        O, I, H, W = p.shape
        return chunk.view(O, H, W, I).permute(0, 3, 1, 2)
    return chunk.view(p.shape)


def _fusion_order(named):
    """Same parameters, ordered so that each attention block's query/key/value weights (and then their biases) sit back to
    back in the flat buffers: the three projections then run as ONE [3*hidden, hidden] GEMM forward and backward
    (vbg/functions.BertLayerFn).  The optimizers are element-wise, so the order is otherwise irrelevant."""
    by_name = dict(named)
    out, used = [], set()
    for name, p in named:
        if name in used:
            continue
        if name.endswith("attention.self.query.weight"):
            stem = name[:-len("query.weight")]
            block = [stem + k for k in ("query.weight", "key.weight", "value.weight", "query.bias", "key.bias", "value.bias")]
            if all(b in by_name for b in block):
                for b in block:
                    out.append((b, by_name[b]))
                    used.add(b)
                continue
        out.append((name, p))
        used.add(name)
    return out


class FlatGroup:
    def __init__(self, named: List[Tuple[str, torch.nn.Parameter]], device):
        named = _fusion_order(named)
        self.names = [n for n, _ in named]
        self.params = [p for _, p in named]
        sizes = [p.numel() for p in self.params]
        # 16-byte aligned slots so float4 kernels and vector GEMM loads stay legal on every view
        self.offsets, o = [], 0
        for s in sizes:
            self.offsets.append(o)
            o += (s + 3) // 4 * 4
        self.total = o
        self.pflat = torch.zeros((self.total,), device=device, dtype=torch.float32)
        self.gflat = torch.zeros((self.total,), device=device, dtype=torch.float32)
        for p, off in zip(self.params, self.offsets):
            v = _phys_view(self.pflat, off, p.data)
            v.copy_(p.data)
            p.data = v
            p.grad = _phys_view(self.gflat, off, p.data)
            p._vbg_sunk = True               # weight-gradient GEMMs accumulate straight into this view

    def zero_grad(self):
        self.gflat.zero_()
        for p, off in zip(self.params, self.offsets):       # re-attach if something set grads to None
            if p.grad is None or p.grad.data_ptr() != self.gflat.data_ptr() + 4 * off:
                p.grad = _phys_view(self.gflat, off, p.data)


def split_parameters(model: torch.nn.Module):
    """(cnn_named, bert_named) exactly like train_SROIE.py:215-221, minus the statically unused tensors."""
    cnn, bert = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad or any(u in name for u in STATIC_UNUSED):
            continue
        (bert if "bert_model" in name else cnn).append((name, p))
    return cnn, bert


class _FlatOptimizer:
    def __init__(self, named, device, defaults: Dict):
        self.group = FlatGroup(named, device)
        self.param_groups = [dict(defaults, params=self.group.params)]      # same knobs the reference's loop writes (lr, weight_decay)
        self.grad_scale = 1.0
        self.steps = 0

    def zero_grad(self, set_to_none: bool = False):
        self.group.zero_grad()

    def state_dict(self):
        return {"steps": self.steps, "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups],
                "state": {k: v for k, v in self.__dict__.items() if isinstance(v, torch.Tensor)}}


class FusedSGD(_FlatOptimizer):
    """torch.optim.SGD(momentum, weight_decay) semantics (dampening 0, no nesterov) over a flat buffer."""

    def __init__(self, named, device, lr, momentum=0.0, weight_decay=0.0):
        super().__init__(named, device, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self.mom = torch.zeros_like(self.group.pflat)

    def step(self):
        g = self.param_groups[0]
        ops.sgd_step(self.group.pflat, self.group.gflat, self.mom, float(g["lr"]), float(g["momentum"]), float(g["weight_decay"]),
                     self.steps == 0, self.grad_scale)
        self.steps += 1


class FusedAdamW(_FlatOptimizer):
    """torch.optim.AdamW semantics (decoupled weight decay, bias correction, amsgrad off)."""

    def __init__(self, named, device, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        super().__init__(named, device, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.m = torch.zeros_like(self.group.pflat)
        self.v = torch.zeros_like(self.group.pflat)

    def step(self):
        g = self.param_groups[0]
        self.steps += 1
        ops.adamw_step(self.group.pflat, self.group.gflat, self.m, self.v, float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                       float(g["eps"]), float(g["weight_decay"]), self.steps, self.grad_scale)


def clip_grad_norm_(optimizers, max_norm: float, norm_scale: float = 1.0) -> float:
    """torch.nn.utils.clip_grad_norm (pipeline/train_val_utils.py:281-282) over the flat gradient buffers."""
    dev = optimizers[0].group.gflat.device
    acc = torch.zeros((1,), device=dev, dtype=torch.float32)
    for o in optimizers:
        ops.sumsq(o.group.gflat, acc)
    total = float(acc.item()) ** 0.5 * norm_scale
    coef = max_norm / (total + 1e-6)
    if coef < 1.0:
        for o in optimizers:
            ops.scale_(o.group.gflat, coef)
    return total


class FlatReducer:
    """Bucketed, overlapped gradient all-reduce over the flat buffers (replaces DistributedDataParallel's reducer)."""

    def __init__(self, optimizers, bucket_mb: float = 64.0, group=None):
        self.enabled = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.pg = group
        self.optimizers = optimizers
        self.world = dist.get_world_size(group) if self.enabled else 1
        self.buckets = []          # (tensor view, pending count)
        self.handles = []
        self.bucket_of = {}        # id(param) -> bucket index
        self._reported = set()     # sunk parameters already counted this step
        if not self.enabled:
            return
        Fn.GRAD_READY[0] = self._param_ready      # sunk gradients (written by the wgrad kernels) report here
        cap = int(bucket_mb * (1 << 20) / 4)
        for o in optimizers:
            o.grad_scale = 1.0 / self.world
            g = o.group
            start, members = 0, []
            for i, (p, off) in enumerate(zip(g.params, g.offsets)):
                members.append(p)
                end = g.offsets[i + 1] if i + 1 < len(g.params) else g.total
                if end - start >= cap or i + 1 == len(g.params):
                    self._add_bucket(g.gflat[start:end], members)
                    start, members = end, []

    def _add_bucket(self, view, members):
        idx = len(self.buckets)
        self.buckets.append([view, len(members), len(members)])
        for p in members:
            self.bucket_of[id(p)] = idx
            p.register_post_accumulate_grad_hook(lambda _p, idx=idx: self._ready(idx))       # autograd-accumulated gradients

    def _param_ready(self, p):
        """a sunk gradient is complete (each sunk parameter feeds exactly one autograd node per step in this model;
        a repeated report for the same parameter is ignored rather than double-counted)"""
        idx = self.bucket_of.get(id(p))
        if idx is not None and id(p) not in self._reported:
            self._reported.add(id(p))
            self._ready(idx)

    def _ready(self, idx):
        b = self.buckets[idx]
        b[2] -= 1
        if b[2] == 0:
            self.handles.append(dist.all_reduce(b[0], group=self.pg, async_op=True))

    def finish(self):
        """wait for every bucket (call after backward, before the optimizer steps); re-arms the counters"""
        if not self.enabled:
            return
        for b in self.buckets:
            if b[2] != 0:          # a parameter got no gradient this step (zero rows): reduce what is there
                self.handles.append(dist.all_reduce(b[0], group=self.pg, async_op=True))
            b[2] = b[1]
        for h in self.handles:
            h.wait()
        self.handles = []
        self._reported.clear()
