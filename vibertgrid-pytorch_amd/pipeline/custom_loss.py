"""Sampled / OHEM cross-entropy losses, device-driven (reference: pipeline/custom_loss.py
`CrossEntropyLossRandomSample` :9-101 and `CrossEntropyLossOHEM` :104-201).

Semantics kept from the reference, quirks included:
* categories are order-preserving compactions of the per-element losses (`ce_loss[mask]`);
* the random subsets come from python's global `random.sample(range(n), k)` on the HOST, consumed in
  the same order as the reference (so a pinned `random.seed` selects the same elements);
* OHEM indexes the SORTED losses with the ORIGINAL positions of the top-k
  (`sorted_loss[sorted_index[:k]]`, :175-176) — reproduced literally;
* RandomSample returns float64 shape [1], OHEM a 0-dim float32.
Differences: the sort is a stable descending radix sort on the GPU (the reference's `torch.sort` is
unstable; with exactly tied losses its result depends on the tie order — DESIGN.md "OHEM ties");
per-element CE is computed straight from low-resolution logits (`up_shift`), so the seg head never
materialises the x4-upsampled activation; only the selected elements carry gradient, as in the
reference.

A "plan" holds everything that depends on LABELS only (category compactions, counts, host random
draws).  Plans for several losses are resolved with ONE device->host copy (`resolve_plans`).
"""
import random
from typing import List, Optional, Sequence

import numpy as np
import torch

from vbg import functions as Fn
from vbg import ops


class _Cat:
    __slots__ = ("idx", "cnt_dev", "n", "elem")


class RandomSamplePlan:
    """label-only part of CrossEntropyLossRandomSample."""

    def __init__(self, labels_i32: torch.Tensor, ncls_logits: int, sample_list: Optional[Sequence[int]]):
        self.labels = labels_i32
        self.cats: List[_Cat] = []
        self.plain = sample_list is None          # reference :36-44: plain (weighted) mean CE over every element
        self.num_keep_total = 0
        if self.plain:
            self.sample_list = None
            return
        self.sample_list = list(sample_list)
        ncat = len(sample_list)
        if ncat == 2 and ncls_logits >= 2:
            spec = [(0, True), (0, False)]
        else:
            assert ncat == ncls_logits, f"shape mismatch, number of elements in sample_list must be 2 or equals dimensions of input, {ncat} and {ncls_logits} given"
            spec = [(c, True) for c in range(ncat)]
        for value, eq in spec:
            c = _Cat()
            c.idx, c.cnt_dev = ops.compact(labels_i32, value, eq)
            self.cats.append(c)

    def count_tensors(self):
        return [c.cnt_dev for c in self.cats]

    def draw(self, counts: Sequence[int]):
        """host half of the selection: the random draws in the reference's order -> [(category, positions or None)]"""
        self.num_keep_total = 0
        out = []
        if self.plain:
            return out
        for c, n, k in zip(self.cats, counts, self.sample_list):
            c.n = int(n)
            keep = min(k, c.n)
            self.num_keep_total += keep
            out.append((c, random.sample(range(c.n), keep) if keep == k else None))
        return out

    def resolve(self, counts: Sequence[int]):
        _apply_draws(self.draw(counts), self.labels.device)


class OhemPlan:
    """label-only part of CrossEntropyLossOHEM (positives = label != 0).  `keyed`: labels are 1 (positive) / 0 (negative) /
    anything else = not part of the loss (BCE-OHEM over a predicted-positive subset, semantic_segmentation_head.py:216-226)."""

    def __init__(self, labels_i32: torch.Tensor, num_pos: int, num_neg: int, rand: bool, keyed: bool = False):
        self.labels, self.num_pos, self.num_neg, self.rand = labels_i32, num_pos, num_neg, rand
        self.plain = (num_pos == -1 and num_neg == -1)
        self.cats: List[_Cat] = []
        if not self.plain:
            for value, eq in (((1, True), (0, True)) if keyed else ((0, False), (0, True))):          # positives first, like the reference
                c = _Cat()
                c.idx, c.cnt_dev = ops.compact(labels_i32, value, eq)
                self.cats.append(c)

    def count_tensors(self):
        return [c.cnt_dev for c in self.cats]

    def draw(self, counts: Sequence[int]):
        out = []
        for c, n, k in zip(self.cats, counts, (self.num_pos, self.num_neg)):
            c.n = int(n)
            out.append((c, random.sample(range(c.n), 2 * k) if self.rand and 2 * k < c.n else None))
        return out

    def resolve(self, counts: Sequence[int]):
        _apply_draws(self.draw(counts), self.labels.device)


def _apply_draws(draws, dev):
    """device half: ALL drawn positions travel in one pinned, asynchronous upload (a pageable copy per category made the host wait for
    the stream six times per step), then one gather per category"""
    sel = [p for _, p in draws if p is not None]
    d = ops.h2d(np.asarray([x for p in sel for x in p], dtype=np.int32), dev) if sel else None
    o = 0
    for c, p in draws:
        if p is None:
            c.elem = c.idx[:c.n]
        else:
            c.elem = ops.gather_i32(c.idx, d[o:o + len(p)])
            o += len(p)


class PendingCounts:
    """the category counts of several plans on their way to the host: one asynchronous D2H copy into pinned memory + an event, so
    the caller can enqueue more device work (the whole BERT / CNN trunk) before it needs the numbers"""

    def __init__(self, plans):
        self.plans = plans
        tens = [t for p in plans for t in p.count_tensors()]
        self.host = None
        if tens:
            dev = torch.cat(tens)
            self.host = torch.empty(dev.shape, dtype=dev.dtype, pin_memory=True)
            self.host.copy_(dev, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record()
            self._keep = dev

    def finish(self):
        """wait for the copy only (not for work enqueued after it), then the host-side random draws in the reference's order"""
        counts = []
        if self.host is not None:
            self.event.synchronize()
            counts = self.host.tolist()
        o = 0
        draws, dev = [], None
        for p in self.plans:
            k = len(p.count_tensors())
            draws += p.draw(counts[o:o + k])
            dev = p.labels.device
            o += k
        if draws:
            _apply_draws(draws, dev)


def resolve_plans(plans):
    """One D2H copy for the category counts of all plans, then the host-side random draws in order."""
    PendingCounts(plans).finish()


def plain_mean_ce(logits2d: torch.Tensor, labels_i32: torch.Tensor, weight: Optional[torch.Tensor], up_shift: int, H: int,
                  W: int) -> torch.Tensor:
    """`F.cross_entropy(input, target, weight, reduction="mean")` over EVERY element (the reference's path for
    `sample_list=None` :36-44 and `num_hard_* == -1` :127-136): sum_i w[t_i] CE_i / sum_i w[t_i]; 0-dim fp32."""
    n = labels_i32.numel()
    elem = torch.arange(n, dtype=torch.int32, device=logits2d.device)
    if weight is None:
        return Fn.SelectedCEFn.apply(logits2d, elem, labels_i32, None, 1.0 / n, up_shift, H, W)
    w = weight.to(device=logits2d.device, dtype=torch.float32)
    denom = (torch.bincount(labels_i32.long(), minlength=w.numel()).to(torch.float32) * w).sum()
    return Fn.SelectedCEFn.apply(logits2d, elem, labels_i32, w, 1.0, up_shift, H, W) / denom


class CrossEntropyLossRandomSample(torch.nn.Module):
    def __init__(self, sample_list: Optional[List], weight: Optional[torch.Tensor] = None, reduction: str = "mean") -> None:
        super().__init__()
        assert reduction == "mean", "only the reduction the model uses is implemented"
        self.sample_list = sample_list
        if sample_list is not None:
            assert len(sample_list) >= 2, f"sample list must contains at least two elements, {len(sample_list)} given"
        self.register_buffer("weight", weight)

    def plan(self, labels_i32, ncls_logits):
        return RandomSamplePlan(labels_i32, ncls_logits, self.sample_list)

    def forward(self, logits2d: torch.Tensor, labels_i32: torch.Tensor, plan: RandomSamplePlan = None, up_shift: int = 0,
                H: int = 0, W: int = 0) -> torch.Tensor:
        """logits2d [rows, C] fp32 (rows = low-res pixels when H > 0); labels int32 flat (full resolution)."""
        if plan is None:
            plan = self.plan(labels_i32, logits2d.shape[1])
            resolve_plans([plan])
        if plan.plain:
            return plain_mean_ce(logits2d, labels_i32, self.weight, up_shift, H, W)
        # (round 6: ONE selected-CE node over the categories' concatenated element lists instead of one per category -- the same sum, the
        #  same per-element gradient scale, a third of the small launches in the forward's tail and the backward's head)
        elems = [c.elem for c in plan.cats if c.elem.numel()]
        if not elems:
            return torch.zeros((1,), dtype=torch.float64, device=logits2d.device) / plan.num_keep_total
        e = elems[0] if len(elems) == 1 else torch.cat(elems)
        total = Fn.SelectedCEFn.apply(logits2d, e.contiguous(), labels_i32, self.weight, 1.0, up_shift, H, W).double().reshape(1)
        return total / plan.num_keep_total


class CrossEntropyLossOHEM(torch.nn.Module):
    def __init__(self, num_hard_positive: int = -1, num_hard_negative: int = -1, weight: Optional[torch.Tensor] = None,
                 reduction: str = "mean", random: bool = False) -> None:
        super().__init__()
        assert reduction == "mean", "only the reduction the model uses is implemented"
        self.num_hard_positive, self.num_hard_negative, self.random = num_hard_positive, num_hard_negative, random
        self.register_buffer("weight", weight)

    def plan(self, labels_i32, keyed: bool = False):
        return OhemPlan(labels_i32, self.num_hard_positive, self.num_hard_negative, self.random, keyed)

    def forward(self, logits2d: torch.Tensor, labels_i32: torch.Tensor, plan: OhemPlan = None, up_shift: int = 0, H: int = 0,
                W: int = 0) -> torch.Tensor:
        if plan is None:
            plan = self.plan(labels_i32)
            resolve_plans([plan])
        if plan.plain:
            return plain_mean_ce(logits2d, labels_i32, self.weight, up_shift, H, W)
        elems, keeps = [], []
        for c, k in zip(plan.cats, (self.num_hard_positive, self.num_hard_negative)):
            m = int(c.elem.numel())
            keep = min(m, k)
            keeps.append(keep)
            if 0 < keep < m:
                v = ops.ce_fwd(logits2d.detach(), c.elem.contiguous(), labels_i32, m, self.weight, up_shift, H, W)   # losses of this category only
                _, si = ops.sort_desc(v)                       # si[r] = position (in c.elem) of rank r
                ranks = ops.gather_i32(si, si[:keep].contiguous())   # the reference's sorted_loss[sorted_index[:k]]
                elems.append(ops.gather_i32(c.elem, ranks))
            else:
                elems.append(c.elem)
        denom = keeps[0] + keeps[1]
        sel = [e for e in elems if e.numel()]
        if not sel:
            return logits2d.sum() * 0.0
        # (one node over positives + negatives: both carry the scale 1 / (k_pos + k_neg))
        e = sel[0] if len(sel) == 1 else torch.cat(sel)
        return Fn.SelectedCEFn.apply(logits2d, e.contiguous(), labels_i32, self.weight, 1.0 / denom, up_shift, H, W)


# ----------------------------------------------------------------------------------------------
# binary variants (classifier_mode full): reference `BCELossRandomSample` :204-290, `BCELossOHEM` :293-382
# ----------------------------------------------------------------------------------------------
_BCE_WEIGHT_MSG = ("the binary (BCE) losses of classifier_mode 'full' / 'crf' take no `weight`: the reference hands its per-CLASS "
                   "`loss_weights` to F.binary_cross_entropy_with_logits as a per-ELEMENT weight (pipeline/custom_loss.py:228-241, "
                   ":312-333), which only broadcasts when the number of elements happens to equal num_classes; "
                   "use loss_weights=None with these modes, or classifier_mode='simp' for class-weighted losses")


def _two_column(logit: torch.Tensor) -> torch.Tensor:
    """BCE-with-logits(x, t) is exactly the 2-class cross entropy of the logit pair (0, x) with target t
    (logsumexp(0, x) - t*x = max(x, 0) + log1p(exp(-|x|)) - t*x), so the CE kernels serve both."""
    x = logit.reshape(-1, 1).to(torch.float32)
    return torch.cat([torch.zeros_like(x), x], dim=1)


class BCELossRandomSample(torch.nn.Module):
    """categories are split by the SIGN OF THE PREDICTION (`mask = input > 0`, :250): category 0 = input <= 0 sampled with
    sample_list[0], category 1 = input > 0 with sample_list[1]; float64 [1] like the reference."""

    def __init__(self, sample_list: List, weight: Optional[torch.Tensor] = None, reduction: str = "mean") -> None:
        super().__init__()
        assert reduction == "mean", "only the reduction the model uses is implemented"
        if weight is not None:
            raise NotImplementedError(_BCE_WEIGHT_MSG)
        assert sample_list is not None and len(sample_list) == 2, "sample list must contain two elements"
        self.sample_list = sample_list

    def forward(self, input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        assert input.dim() == 1 or (input.dim() == 2 and input.shape[1] == 1), "invalid shape"
        x = input.reshape(-1)
        keys = (x.detach() > 0).to(torch.int32)
        labels = (target.reshape(-1) != 0).to(torch.int32)
        plan = RandomSamplePlan(keys, 2, self.sample_list)
        resolve_plans([plan])
        logits2 = _two_column(x)
        total = torch.zeros((1,), dtype=torch.float64, device=x.device)
        for c in plan.cats:
            if c.elem.numel():
                total = total + Fn.SelectedCEFn.apply(logits2, c.elem.contiguous(), labels, None, 1.0, 0, 0, 0).double()
        return total / plan.num_keep_total


class BCELossOHEM(torch.nn.Module):
    """positives = target != 0; everything else (random pre-sample, descending sort, the `sorted[sorted_index[:k]]` quirk,
    mean over the keep counts) is the CE version on the logit pair (0, x)."""

    def __init__(self, num_hard_positive: int = -1, num_hard_negative: int = -1, weight: Optional[torch.Tensor] = None,
                 reduction: str = "mean", random: bool = False) -> None:
        super().__init__()
        if weight is not None:
            raise NotImplementedError(_BCE_WEIGHT_MSG)
        self.ce = CrossEntropyLossOHEM(num_hard_positive, num_hard_negative, None, reduction, random)

    def forward(self, input: torch.Tensor, target: torch.Tensor = None, keyed_labels: torch.Tensor = None, up_shift: int = 0,
                H: int = 0, W: int = 0) -> torch.Tensor:
        """input: logits [n] / [n,1] (or low-resolution rows when H > 0); either `target` (0/1 per element) or
        `keyed_labels` int32 over the full-resolution elements: 1 positive, 0 negative, other = outside the loss."""
        logits2 = _two_column(input)
        if keyed_labels is None:
            labels = (target.reshape(-1) != 0).to(torch.int32)
            return self.ce(logits2, labels)
        plan = self.ce.plan(keyed_labels, keyed=True)
        resolve_plans([plan])
        return self.ce(logits2, keyed_labels, plan, up_shift, H, W)
