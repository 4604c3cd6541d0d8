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
  averaging is folded into the optimizer kernels.  The flat buffers hold the parameters in REVERSE forward order (heads first,
  stem / word embeddings last), so contiguous buckets complete roughly in the order autograd produces them; buckets are launched
  strictly in ONE sequence that is identical on every rank (the completion order rank 0 observed in its first step, broadcast
  once), so data-dependent graphs (classifier_mode full / crf: a per-class net may get no gradient on one rank) can never make two
  ranks issue collectives in different orders.  SyncBatchNorm statistics share the reducer's communicator by default (one
  communicator, one rank-invariant program order: the boring configuration for graphs that are the same on every rank); an own
  communicator for them, concurrent with the buckets, is the opt-in (`sync_bn_group="new"`, FlatReducer.__init__).  Parameters that never receive a gradient
  (`bert_model.pooler.*`, `backbone.resnet.fc.*`) are kept out of the buffers, which is what
  `find_unused_parameters=True` + "skip params with grad None" amounts to in the reference.
"""
import os
import time
from typing import Dict, List, Sequence, Tuple

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
    """Same parameters in the order of the flat buffers:
    * each attention block's query/key/value weights (and then their biases) sit back to back, so the three projections run as
      ONE [3*hidden, hidden] GEMM forward and backward (vbg/functions.BertLayerFn);
    * the units (such a block, or a single parameter) are laid out in REVERSE registration (= forward) order, so the
      contiguous gradient buckets of FlatReducer fill up in roughly the order backward produces them: heads, FPN, trunk
      top-down, stem last; encoder layer 11 ... 0, the embedding tables last.
    The optimizers are element-wise, so the order is otherwise irrelevant."""
    by_name = dict(named)
    units, used = [], set()
    for name, p in named:
        if name in used:
            continue
        if name.endswith("attention.self.query.weight"):
            stem = name[:-len("query.weight")]
            block = [stem + k for k in ("query.weight", "key.weight", "value.weight", "query.bias", "key.bias", "value.bias")]
            if all(b in by_name for b in block):
                units.append([(b, by_name[b]) for b in block])
                used.update(block)
                continue
        units.append([(name, p)])
        used.add(name)
    return [np_ for u in reversed(units) for np_ in u]


class FlatGroup:
    def __init__(self, named: List[Tuple[str, torch.nn.Parameter]], device):
        self.ref_names = [n for n, _ in named]            # the caller's (= reference's) parameter order, for state_dict indices
        named = _fusion_order(named)
        self.names = [n for n, _ in named]
        self.params = [p for _, p in named]
        sizes = [p.numel() for p in self.params]
        # 32-byte aligned slots: float4 kernels / vector GEMM loads stay legal on every view, and the same offsets address the
        # bf16 plane image of the buffer 16-byte aligned (LDS-DMA rows)
        self.offsets, o = [], 0
        for s in sizes:
            self.offsets.append(o)
            o += (s + 7) // 8 * 8
        self.total = (o + 31) // 32 * 32
        self.pflat = torch.zeros((self.total,), device=device, dtype=torch.float32)
        self.gflat = torch.zeros((self.total,), device=device, dtype=torch.float32)
        self.gviews = []                     # the persistent gradient views (re-attached when something set .grad to None)
        self._ptrs = []                      # device address every parameter must still have for the group to be its home
        for p, off in zip(self.params, self.offsets):
            v = _phys_view(self.pflat, off, p.data)
            v.copy_(p.data)
            p.data = v
            gv = _phys_view(self.gflat, off, p.data)
            p.grad = gv
            p._vbg_sunk = True               # weight-gradient GEMMs accumulate straight into this view
            p._vbg_flat = (self, off)
            self.gviews.append(gv)
            self._ptrs.append(self.pflat.data_ptr() + 4 * off)
        self._index = {id(p): i for i, p in enumerate(self.params)}
        # bf16 planes of the whole buffer (csrc/gemm_planes.hip operands), refreshed at most once per parameter version: the plain
        # image [3][total] in ONE elementwise launch, the transposed images of the matrices that asked for one in ONE batched launch
        self._planes = None
        self._planes_tag = None
        self._plain_keys, self._plain_tags = set(), {}
        self._pair = None                    # fp16-pair image [2][total] (operands of the form-1 forward products)
        self._pair_tag = None
        # transposed images: bf16 planes (`_tp`) and fp16-pair planes (`_tq`); jobs: (offset, rows, cols) -> (slot offset, ld)
        self._tp = {"jobs": {}, "buf": None, "tbl": None, "tag": None, "tiles": 0}
        self._tq = {"jobs": {}, "buf": None, "tbl": None, "tag": None, "tiles": 0}
        self._ver = {}
        self._owners = {}
        self._armed = []                     # indices whose gradient view arm() attached in the current backward

    def zero_grad(self):
        self.gflat.zero_()
        for p, gv in zip(self.params, self.gviews):          # re-attach if something set grads to None
            if p.grad is not gv:
                p.grad = gv

    def valid(self) -> bool:
        """every parameter still lives where the group put it (Module.to / .half / a fresh `p.data = ...` moves it away)"""
        return all(p.data_ptr() == a for p, a in zip(self.params, self._ptrs))

    def arm(self):
        """start of a backward pass (vbg.functions.StepRootFn): the reference's loop calls `optimizer.zero_grad()` between forward and
        backward (pipeline/train_val_utils.py:272-273) and torch.optim's default is set_to_none -- every `.grad` is None then.  The
        gradient views come back, over ONE memset of the flat buffer when all of them were dropped (the usual case), slice by slice
        otherwise (a caller that keeps some gradients to accumulate into).  A gradient the caller left in place is accumulated into,
        as torch would."""
        # (a parameter frozen after it was homed keeps `.grad = None`: torch.optim skips it, as it would without flat storage)
        missing, foreign = [], []
        for i, (p, gv) in enumerate(zip(self.params, self.gviews)):
            g = p.grad
            if g is None:
                if p.requires_grad:
                    missing.append(i)
            elif g is not gv:
                foreign.append(i)
        self._armed = missing                # (ModelHome.drop_untouched only hands None back for gradients attached HERE, in this backward)
        if len(missing) == len(self.params):
            self.gflat.zero_()
        else:
            # one memset per contiguous RUN of missing parameters (one frozen parameter used to turn the single memset into
            # ~400 per-parameter launches)
            k = 0
            while k < len(missing):
                j = k
                while j + 1 < len(missing) and missing[j + 1] == missing[j] + 1:
                    j += 1
                i0, i1 = missing[k], missing[j]
                end = self.offsets[i1 + 1] if i1 + 1 < len(self.params) else self.total
                self.gflat[self.offsets[i0]:end].zero_()
                k = j + 1
        for i in missing:
            self.params[i].grad = self.gviews[i]
        # a `.grad` that is somebody else's tensor (DistributedDataParallel's finalize_backward leaves one on a locally-unused parameter):
        # its contents move into the flat view -- the weight-gradient kernels accumulate into the VIEW's memory (wgrad_dest), and
        # torch.optim must see what they wrote
        for i in foreign:
            g = self.params[i].grad
            gv = self.gviews[i]
            if g.shape == gv.shape and g.device == gv.device:
                with torch.no_grad():
                    gv.copy_(g)
                self.params[i].grad = gv

    def attach(self, p):
        """gradient view of one parameter whose `.grad` is None outside an armed backward (a Function used on its own): zeroed, attached"""
        gv = self.gviews[self._index[id(p)]]
        gv.zero_()
        p.grad = gv
        return gv

    def invalidate(self):
        """forget every cached plane image of the buffer.  The caches follow the optimizer kernels (weight epoch), torch in-place
        updates (`_version` of the buffer and of the parameter asked for) and load_state_dict; a write through `param.data`
        (`p.data.copy_(...)`, `p.data.mul_(...)`: EMA / weight-tying code) moves none of these counters -- call this, or
        vbg.ops.bump_weight_epoch(), after such a write"""
        self._planes_tag = self._pair_tag = self._tp["tag"] = self._tq["tag"] = None
        self._plain_tags.clear()
        self._ver.clear()

    def _tag(self):
        return (ops._W_EPOCH[0], self.pflat._version)

    def _stale(self, which: str, off: int, owners) -> bool:
        """refresh of the whole-buffer image `which` needed for the matrix at `off`?  The optimizer kernels of this library bump the epoch
        (`_tag`); torch's own in-place updates (torch.optim on the flat views, load_state_dict, tests) only move the version counters of
        the parameters they touch.  `owners`: the parameter(s) the matrix consists of.  The image is split from the WHOLE buffer, so at a
        refresh the versions of every matrix registered for it are noted -- a matrix asked for later in the same step is then known to be
        current, and one whose counter has moved since (the next step) is known to be stale.  (Round 5: until then a matrix that had not
        been asked for since the last refresh counted as current -- the transposed images, which register their matrices one refresh at a
        time, served the previous step's weights to the data-gradient products under torch.optim.)"""
        reg = self._owners.setdefault(which, {})
        reg[off] = owners
        ver = tuple(t._version for t in owners)
        seen = self._ver.setdefault(which, {})
        if {"p": self._planes_tag, "q": self._pair_tag, "t": self._tp["tag"], "u": self._tq["tag"]}[which] == self._tag() and seen.get(off) == ver:
            return False
        seen.clear()
        for o2, ow in reg.items():
            seen[o2] = tuple(t._version for t in ow)
        return True

    def planes_of(self, off: int, rows: int, cols: int, owners=()):
        """plane operand of the matrix [rows, cols] stored at element offset `off` of the parameter buffer (None if its layout does not
        allow it)"""
        if cols % 32 or off % 8:
            return None
        if self._planes is None:
            self._planes = torch.empty((3, self.total), device=self.pflat.device, dtype=torch.int16)
        key = (off, rows, cols)
        self._plain_keys.add(key)
        if len(self._plain_keys) <= 16:
            # few matrices want the bf16 image (with the fp16-pair forms on: the 12 attention-output projections of bert-base, 7 MB of a
            # 435 MB buffer): each is split on its own, once per parameter version -- not the whole buffer (0.2 ms per step)
            tag = (self._tag(), tuple(t._version for t in owners))
            if self._plain_tags.get(key) != tag:
                with torch.no_grad():
                    ops.split_planes(self.pflat.detach()[off:off + rows * cols].view(rows, cols),
                                     out=ops.Planes(self._planes[:, off:off + rows * cols].view(3, rows, cols), rows, cols, cols))
                self._plain_tags[key] = tag
            return ops.Planes(self._planes[:, off:off + rows * cols].view(3, rows, cols), rows, cols, cols)
        if self._stale("p", off, owners):
            with torch.no_grad():
                ops.split_planes(self.pflat.detach().view(-1, 32), out=ops.Planes(self._planes.view(3, -1, 32), self.total // 32, 32, 32))
            self._planes_tag = self._tag()
        return ops.Planes(self._planes[:, off:off + rows * cols].view(3, rows, cols), rows, cols, cols)

    def pair_of(self, off: int, rows: int, cols: int, owners=()):
        """the same matrix as an fp16-pair plane operand (csrc/gemm_planes.hip FORM 1): one launch per optimizer step for the whole buffer"""
        if cols % 32 or off % 8:
            return None
        if self._pair is None:
            self._pair = torch.empty((2, self.total), device=self.pflat.device, dtype=torch.int16)
        if self._stale("q", off, owners):
            with torch.no_grad():
                ops.split_planes_pair(self.pflat.detach().view(-1, 32), out=ops.Planes(self._pair.view(2, -1, 32), self.total // 32, 32, 32))
            self._pair_tag = self._tag()
        return ops.Planes(self._pair[:, off:off + rows * cols].view(2, rows, cols), rows, cols, cols)

    def planes_t_of(self, off: int, rows: int, cols: int, owners=(), pair=False):
        """plane operand of the TRANSPOSE of that matrix ([cols, rows], reduction over rows); pair: as two fp16 planes (form 1)"""
        if off % 8:
            return None
        st = self._tq if pair else self._tp
        key = (off, rows, cols)
        if key not in st["jobs"]:
            ld = (rows + 31) // 32 * 32
            slot = sum(c * l for (_, _, c), (_, l) in st["jobs"].items())
            st["jobs"][key] = (slot, ld)
            st["tbl"] = None
        if st["tbl"] is None:
            tot = sum(c * l for (_, _, c), (_, l) in st["jobs"].items())
            st["buf"] = torch.empty((2 if pair else 3, (tot + 7) // 8 * 8), device=self.pflat.device, dtype=torch.int16)
            rows_, first = [], 0
            for (o, r, c), (slot, ld) in st["jobs"].items():
                rows_.append([o, r, c, slot, ld, first])
                first += ((ld + 63) // 64) * ((c + 63) // 64)
            st["tiles"] = first
            st["tbl"] = torch.tensor(rows_, dtype=torch.int64).to(self.pflat.device)
            st["tag"] = None
        which = "u" if pair else "t"
        if self._stale(which, off, owners):
            (ops.split_planes_pair_t_batched if pair else ops.split_planes_t_batched)(self.pflat, st["buf"], st["tbl"], len(st["jobs"]), st["tiles"])
            st["tag"] = self._tag()
        slot, ld = st["jobs"][key]
        return ops.Planes(st["buf"][:, slot:slot + cols * ld].view(2 if pair else 3, cols, ld), cols, rows, ld)

    def view(self, flat: torch.Tensor, i: int) -> torch.Tensor:
        """parameter i's slice of another flat buffer of this layout (optimizer state), shaped / laid out like the parameter"""
        return _phys_view(flat, self.offsets[i], self.params[i].data)


class NamedParams(list):
    """[(name, parameter)] that remembers the reference's FULL parameter list of the group (`ref_names`): torch optimizer checkpoints
    key their state by index into that list (train_SROIE.py:215-221 keeps the never-used tensors in it), so the indices written and
    read by _FlatOptimizer.state_dict / load_state_dict stay those of the reference's optimizer even though the unused tensors are
    kept out of the flat buffers"""
    ref_names: List[str] = None


def split_parameters(model: torch.nn.Module, unused: Sequence[str] = STATIC_UNUSED):
    """(cnn_named, bert_named) exactly like train_SROIE.py:215-221, minus the tensors that never receive a gradient
    (`unused`: name fragments; the reference keeps them in its optimizers, where `grad is None` makes every step skip them -- and
    where they still occupy an index: see NamedParams)."""
    cnn, bert = NamedParams(), NamedParams()
    cnn.ref_names, bert.ref_names = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        dst = bert if "bert_model" in name else cnn
        dst.ref_names.append(name)
        if any(u in name for u in unused):
            continue
        dst.append((name, p))
    return cnn, bert


def home_parameters(model: torch.nn.Module, device=None) -> List[FlatGroup]:
    """Flat parameter / gradient storage owned by the MODEL (called by ViBERTgridNet at its first training forward): the trainable
    parameters move into one flat buffer per optimizer group of the reference (train_SROIE.py:215-221: "bert_model" in name -> AdamW,
    the rest -> SGD), their `.grad` become views of a flat gradient buffer that the weight-gradient kernels accumulate into, and the
    plane / filter images the matrix kernels read are refreshed once per weight version for the whole buffer.  None of this depends on
    the optimizer class the caller constructs: torch.optim.SGD / AdamW update the views in place (their version counters tell the
    caches), FusedSGD / FusedAdamW adopt the groups they find.  Groups that are already in place (and still valid) are kept."""
    cnn, bert = split_parameters(model)
    groups = []
    for named in (cnn, bert):
        if not named:
            continue
        found = {}
        for _, p in named:
            g = getattr(p, "_vbg_flat", (None,))[0]
            found[id(g)] = g
        if None not in found.values() and all(g.valid() for g in found.values()):
            # already homed (an optimizer of vbg.optim built before the first forward, possibly over a different split): kept as is
            groups.extend(g for g in found.values() if all(g is not h for h in groups))
            continue
        dev = device if device is not None else named[0][1].device
        g = FlatGroup(list(named), dev)
        g.ref_names = list(named.ref_names)
        groups.append(g)
    return groups


class ModelHome:
    """What ViBERTgridNet keeps about the flat storage its parameters live in: the groups, and -- for graphs that depend on the data --
    which parameters took part in the current backward (post-accumulate hooks fire for every parameter whose autograd node ran, also
    when a Function wrote the gradient itself and returned None), so that the others get `.grad = None` back when the backward ends,
    as torch.optim expects of parameters that received no gradient (it skips them: no weight decay, no momentum step)."""

    def __init__(self, model, track_unused: bool):
        self.groups = home_parameters(model)
        self.touched = set() if track_unused else None
        if track_unused:
            for g in self.groups:
                for p in g.params:
                    if not getattr(p, "_vbg_touch_hook", False):
                        p.register_post_accumulate_grad_hook(self._touch)
                        p._vbg_touch_hook = True
                    p._vbg_home = self

    @staticmethod
    def _touch(p):
        h = getattr(p, "_vbg_home", None)
        if h is not None and h.touched is not None:
            h.touched.add(id(p))

    def valid(self) -> bool:
        return all(g.valid() for g in self.groups)

    def drop_untouched(self):
        """end of a backward (classifier_mode full): a parameter that took no part in it gets `.grad = None` back -- but only where arm()
        attached the view in THIS backward.  A gradient that already existed when the backward started (gradient accumulation, DDP
        `no_sync` micro-steps) stays, as torch would have left it."""
        t = self.touched
        for g in self.groups:
            for i in g._armed:
                p = g.params[i]
                if id(p) not in t:
                    p.grad = None
            g._armed = []


def _torch_defaults(cls, **kw):
    """the param_group keys of the torch optimizer this one stands in for (so a checkpoint written here loads into it)"""
    return dict(cls([torch.nn.Parameter(torch.zeros(1))], **kw).defaults)


class _FlatOptimizer(torch.optim.Optimizer):
    """A torch.optim.Optimizer whose parameters, gradients and state live in flat buffers and whose step is one HIP launch.
    Being an Optimizer, it plugs into the reference's loop unchanged: `StepLR(optimizer=...)` (train_SROIE.py:247) and
    `scaler.step(optimizer)` / `scaler.unscale_` (pipeline/train_val_utils.py:274-278) operate on `param_groups` and on the
    `.grad` views; `state_dict()` / `load_state_dict()` use torch's own optimizer checkpoint format (per-parameter state keyed by
    the parameter's index in the caller's list), so optimizer checkpoints interchange with torch.optim.SGD / AdamW
    (train_SROIE.py:377-416 saves them, resume loads them).
    One deliberate difference: the step is element-wise over the whole flat range, so a parameter that received NO gradient in a
    step still gets weight decay / momentum applied (torch skips `grad is None` parameters).  On this model that only concerns the
    per-class nets of classifier_mode full in steps where no segment was predicted positive."""

    _state_names: Tuple[str, ...] = ()

    def __init__(self, named, device, defaults: Dict):
        ref = getattr(named, "ref_names", None)
        named = list(named)
        # a group the model (or an earlier optimizer) already homed exactly these parameters in is adopted, not rebuilt
        homes = {id(getattr(p, "_vbg_flat", (None,))[0]): getattr(p, "_vbg_flat", (None,))[0] for _, p in named}
        g = next(iter(homes.values())) if len(homes) == 1 else None
        if (g is not None and len(g.params) == len(named) and set(g._index) == {id(p) for _, p in named} and g.valid()
                and g.pflat.device.type == torch.device(device).type and torch.device(device).index in (None, g.pflat.device.index)):
            self.group = g
            g.zero_grad()
        else:
            self.group = FlatGroup(named, device)
        if ref is not None:                 # indices of the reference optimizer's full parameter list (unused tensors included)
            assert set(n for n, _ in named) <= set(ref)
            self.group.ref_names = list(ref)
        super().__init__([p for _, p in named], defaults)
        self.grad_scale = 1.0
        self.steps = 0
        import weakref
        self.group._vbg_optimizer = weakref.ref(self)          # (ViBERTgridNet._home: a group a live optimizer steps is never silently replaced)

    def zero_grad(self, set_to_none: bool = False):
        self.group.zero_grad()

    # ---- checkpoint format of torch.optim ---------------------------------------------------------------------------
    def _flat_state(self) -> Dict[str, torch.Tensor]:
        raise NotImplementedError

    def state_dict(self):
        g = self.group
        pos = {n: i for i, n in enumerate(g.names)}
        state = {}
        if self.steps > 0:
            for ref_i, n in enumerate(g.ref_names):
                if n not in pos:             # a tensor that never receives a gradient: torch keeps no state for it either
                    continue
                i = pos[n]
                st = {k: g.view(flat, i).clone() for k, flat in self._flat_state().items()}
                if "exp_avg" in st:
                    st = {"step": torch.tensor(float(self.steps)), **st}
                state[ref_i] = st
        groups = [dict({k: v for k, v in pg.items() if k != "params"}, params=list(range(len(g.ref_names)))) for pg in self.param_groups]
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        g = self.group
        pos = {n: i for i, n in enumerate(g.names)}
        for pg, src in zip(self.param_groups, sd["param_groups"]):
            pg.update({k: v for k, v in src.items() if k != "params"})
        flats = self._flat_state()
        for f in flats.values():
            f.zero_()
        self.steps = 0
        n_ref = len(sd["param_groups"][0]["params"]) if sd.get("param_groups") else len(g.ref_names)
        if n_ref != len(g.ref_names):
            raise ValueError(f"optimizer checkpoint covers {n_ref} parameters, this optimizer's reference list has {len(g.ref_names)}: "
                             "build the optimizer from split_parameters(model) so that the indices are the reference's")
        for ref_i, st in sd["state"].items():
            n = g.ref_names[int(ref_i)]
            if n not in pos:                 # state of a tensor kept out of the flat buffers (it never receives a gradient)
                continue
            i = pos[n]
            for k, flat in flats.items():
                if k in st and st[k] is not None:
                    g.view(flat, i).copy_(st[k])
            if "step" in st:
                self.steps = max(self.steps, int(float(st["step"])))
            elif flats:
                self.steps = max(self.steps, 1)


class FusedSGD(_FlatOptimizer):
    """torch.optim.SGD(momentum, weight_decay) semantics (dampening 0, no nesterov) over a flat buffer."""

    def __init__(self, named, device, lr, momentum=0.0, weight_decay=0.0):
        super().__init__(named, device, _torch_defaults(torch.optim.SGD, lr=lr, momentum=momentum, weight_decay=weight_decay))
        self.mom = torch.zeros_like(self.group.pflat)

    def _flat_state(self):
        return {"momentum_buffer": self.mom}

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        ops.sgd_step(self.group.pflat, self.group.gflat, self.mom, float(g["lr"]), float(g["momentum"]), float(g["weight_decay"]),
                     self.steps == 0, self.grad_scale)
        self.steps += 1
        ops.bump_weight_epoch()


class FusedAdamW(_FlatOptimizer):
    """torch.optim.AdamW semantics (decoupled weight decay, bias correction, amsgrad off)."""

    def __init__(self, named, device, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        super().__init__(named, device, _torch_defaults(torch.optim.AdamW, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.m = torch.zeros_like(self.group.pflat)
        self.v = torch.zeros_like(self.group.pflat)

    def _flat_state(self):
        return {"exp_avg": self.m, "exp_avg_sq": self.v}

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        self.steps += 1
        ops.adamw_step(self.group.pflat, self.group.gflat, self.m, self.v, float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                       float(g["eps"]), float(g["weight_decay"]), self.steps, self.grad_scale)
        ops.bump_weight_epoch()


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
    """Bucketed, overlapped gradient all-reduce over the flat buffers (replaces DistributedDataParallel's reducer,
    train_SROIE.py:206-210).

    Ordering contract (what keeps N ranks from ever pairing different buffers): bucket collectives are issued in ONE fixed
    sequence `self.order`, bucket `order[k]` only after `order[0..k-1]` have been issued; whatever has not completed by the end
    of backward is flushed by `finish()` in the same sequence.  Step 1 launches nothing during backward: it records the order
    in which this rank's buckets completed; `finish()` then takes rank 0's record as the sequence for every rank (one
    broadcast) -- from step 2 on the buckets fire during backward.  A rank whose graph skips a sub-module in some step simply
    issues the affected bucket (and everything behind it) from `finish()`; the sequence is unchanged."""

    def __init__(self, optimizers, bucket_mb: float = 32.0, group=None, sync_bn_group="auto", serialize_syncbn=None, overlap=None,
                 dry_run=False, force_enable=None, static_graph=None):
        """force_enable (VBG_FORCE_REDUCER=1): run the whole machinery -- hooks, rank-agreed sequence, collectives issued from the staging
        stream, work.wait(), SyncBatchNorm statistics on the communicator -- on a process group of ONE rank as well (every collective is an
        identity there): the one-GPU check of the real backend (ProcessGroupNCCL = RCCL) before a multi-GPU node sees this code.
        static_graph: the caller's word that every rank runs the same autograd graph every step (classifier_mode simp / crf with every
        sub-module used).  With SyncBatchNorm statistics on the buckets' communicator AND overlap, a bucket that one rank issues from
        finish() while the others issue it inside backward sits at a different place among the statistics collectives -- mismatched
        collectives on one communicator.  So on a shared communicator the buckets only leave from inside backward when
        static_graph=True; without that word every bucket is issued from finish(), a rank-invariant point for ANY graph (the overlap is
        what is given up; `sync_bn_group="new"` keeps it for rank-varying graphs on a second communicator).  A rank that asserted
        static_graph=True and then has to flush a bucket from finish() after step 1 raises instead of risking a silent mismatch.
        dry_run (one process): same buckets, same hooks and the same launch sequence, but a bucket's "collective" is a timestamp on
        the stream it would be issued from -- `timeline()` then tells when, inside backward, every bucket could have left
        (tools/bucket_timeline.py)"""
        self.dry = bool(dry_run)
        if force_enable is None:
            force_enable = os.environ.get("VBG_FORCE_REDUCER", "0") != "0"
        self.forced = bool(force_enable) and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) == 1
        self.enabled = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or self.forced)
        if static_graph is None and os.environ.get("VBG_STATIC_GRAPH") is not None:
            static_graph = os.environ["VBG_STATIC_GRAPH"] != "0"
        self.static_graph = static_graph
        self.pg = group
        self.optimizers = optimizers
        self.world = dist.get_world_size(group) if self.enabled else 1
        self.buckets = []          # [tensor view, member count, pending count]
        self.handles = []          # (bucket index, work) of the collectives in flight
        self.bucket_of = {}        # id(param) -> bucket index
        self._reported = set()     # sunk parameters already counted this step
        self.order = None          # the agreed launch sequence (bucket indices); None until the first finish()
        self._observed = []        # completion order seen in the current step
        self._complete = set()
        self._next = 0             # position in `order` of the next bucket to issue
        self._stage = None         # stream the collectives are issued from
        self._streams = {}         # raw handle -> torch stream: compute streams gradients were reported from in this step
        self._events = []          # dry run: (bucket, event) in issue order
        self.steps_done = 0
        self.sync_bn_mode = "none"
        self._phase = "idle"       # what the host is doing, for the watchdog's report
        self._beat = time.monotonic()
        self._watch = None
        self.overlap = (os.environ.get("VBG_DDP_OVERLAP", "1") != "0") if overlap is None else bool(overlap)
        if self.dry:
            self.enabled = False
        elif not self.enabled:
            return
        # SyncBatchNorm statistics and gradient buckets.  The buckets follow ONE rank-agreed sequence, the SyncBatchNorm collectives
        # the program order of forward / backward.  Where the bucket launches fall BETWEEN the SyncBatchNorm collectives is rank-invariant
        # exactly when every rank runs the same autograd graph (classifier_mode simp / crf, every sub-module used every step):
        #   sync_bn_group="default" ("auto" without VBG_SYNCBN_GROUP: THE DEFAULT): statistics and buckets share the reducer's
        #       communicator -- one communicator, one rank-invariant program order of collectives, nothing concurrent.  A statistics
        #       all-reduce queues behind the buckets in flight (a 32 MB bucket: a fraction of a millisecond over xGMI); the encoder's
        #       backward, where most of the gradient bytes leave, contains no BatchNorm.  overlap=False (VBG_DDP_OVERLAP=0) launches every
        #       bucket from finish() instead of from inside backward: the order is then rank-invariant for ANY graph, at the price of
        #       the overlap -- the setting for data-dependent graphs (classifier_mode full) on one communicator;
        #   sync_bn_group="new" (VBG_SYNCBN_GROUP=new; only with group=None -- dist.new_group is a collective over the DEFAULT group,
        #       every rank must construct its reducer): an own communicator for the statistics, collectives of the two communicators
        #       concurrently in flight.  Tolerates rank-varying graphs WITH overlap, but torch documents concurrent collectives on two
        #       NCCL communicators as unsafe when their kernels cannot co-reside: the opt-in A/B, not the default.
        #       serialize_syncbn=True (VBG_SERIALIZE_SYNCBN=1) makes the compute stream wait for the buckets in flight (work.wait() of
        #       every outstanding handle: the collectives run on the backend's own stream) before each statistics collective --
        #       meaningful only with rank-invariant graphs, where the one-communicator default does the same thing for free;
        #   sync_bn_group=<ProcessGroup>: the caller's communicator (required with a sub-group for "new"-style separation).
        if self.dry:
            sync_bn_group = None
        if sync_bn_group == "auto":
            # round 6: the shared communicator is the default again.  The direct communicator has only ever run on a
            # one-rank group, and a second ncclComm_t with collectives in flight beside ProcessGroupNCCL's is the configuration a first
            # N > 1 run should not meet unasked: VBG_SYNCBN_GROUP=direct / sync_bn_group="direct" / bench.py --syncbn-comm direct opt in.
            sync_bn_group = os.environ.get("VBG_SYNCBN_GROUP", "default")
        if Fn.SyncCtx.direct is not None:          # a reducer is being rebuilt: the previous one's communicator goes first
            try:
                Fn.SyncCtx.direct.destroy()
            except Exception:
                pass
        Fn.SyncCtx.direct = None
        if sync_bn_group == "direct":
            # OPT-IN: the statistics on a communicator of this library's own, enqueued on the compute stream (vbg/rccl.py): no event
            # hand-overs to and from ProcessGroupNCCL's stream (measured on one rank: 34.8 vs 35.8 ms per step), and -- the point at
            # N > 1 -- never queued behind a 32 MB gradient bucket on that stream.  Two communicators then have kernels in flight during
            # backward.  That is deadlock-free when every rank ENQUEUES them in the same relative order (streams beyond the device's
            # hardware queues share one, and a collective's kernel parked at the head of a queue holds up whatever sits behind it): the
            # same program on every rank issues the same sequence, which is what static_graph=True asserts -- without it the buckets
            # leave from finish(), behind every statistics collective of the step, as on the shared communicator.
            from . import rccl
            if group is not None or not rccl.available():
                raise ValueError("sync_bn_group='direct' needs backend 'nccl' (RCCL) and the default process group")
            dev = optimizers[0].group.pflat.device
            Fn.SyncCtx.group = group
            comm, err = None, None
            try:
                comm = rccl.DirectComm(dev)
            except Exception as e:
                err = e
            # every rank must end up on the SAME communicator: the ranks agree on success over the default group (a rank-local
            # fallback would leave some ranks on the direct communicator and some on torch.distributed's -- mismatched collectives)
            ok = torch.tensor([0 if comm is None else 1], device=dev, dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 1:
                Fn.SyncCtx.direct = comm
                self.sync_bn_mode = "direct RCCL communicator on the compute stream"
            else:
                import sys
                if comm is not None:
                    comm.destroy()
                print(f"[vbg reducer] no direct RCCL communicator on every rank ({'this rank: ' + type(err).__name__ + ': ' + str(err) if err else 'another rank failed'}); "
                      "SyncBatchNorm statistics through torch.distributed on every rank", file=sys.stderr, flush=True)
                sync_bn_group = "default"
        if sync_bn_group == "direct":
            pass
        elif sync_bn_group == "new":
            if group is not None:
                raise ValueError("FlatReducer(group=<sub-group>): pass sync_bn_group=<ProcessGroup> or 'default' -- dist.new_group is a "
                                 "collective over the default group and would hang when only the sub-group's ranks call it")
            Fn.SyncCtx.group = dist.new_group(ranks=list(range(dist.get_world_size())))
            self.sync_bn_mode = "own communicator"
        elif sync_bn_group == "default":
            Fn.SyncCtx.group = group
            self.sync_bn_mode = "shared communicator"
        elif sync_bn_group is not None:
            Fn.SyncCtx.group = sync_bn_group
            self.sync_bn_mode = "caller's communicator"
        if serialize_syncbn is None:
            serialize_syncbn = os.environ.get("VBG_SERIALIZE_SYNCBN", "0") != "0"
        Fn.SyncCtx.before = self._wait_for_buckets if (serialize_syncbn and not self.dry) else None
        Fn.SyncCtx.seq = 0
        Fn.SyncCtx.force = self.forced
        if self.sync_bn_mode in ("shared communicator", "direct RCCL communicator on the compute stream") and self.overlap and not self.static_graph:
            self.overlap = False       # graphs not known to be rank-invariant: every bucket from finish(), a rank-invariant point
        Fn.GRAD_READY[0] = self._param_ready      # sunk gradients (written by the wgrad kernels) report here
        import weakref
        for o in optimizers:                      # (ViBERTgridNet._overlap_safe: these groups' gradients are collected by a reducer that
            o.group._vbg_reducer = weakref.ref(self)      #  knows about the side streams)
        cap = int(bucket_mb * (1 << 20) / 4)
        for o in optimizers:
            if not self.dry:
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
            p.register_post_accumulate_grad_hook(lambda _p, idx=idx: self._hook_ready(_p, idx))       # autograd-accumulated gradients

    def _hook_ready(self, p, idx):
        """post-accumulate hook of parameter p.  torch runs these hooks for a parameter whose autograd node handed it no gradient
        as well (every sunk parameter: its Function wrote the gradient into the flat buffer itself, reported it through
        `_param_ready` and returned None), so a parameter is counted ONCE per step whichever way it reports first -- counting both
        let a bucket reach zero when only half of its members were complete, and the bucket's all-reduce then raced the kernels
        that were still writing the rest (the last bucket lost: ranks drifted apart in the stem's parameters)."""
        if id(p) in self._reported:
            return
        self._reported.add(id(p))
        self._ready(idx)

    def _param_ready(self, p):
        """a sunk gradient is complete (each sunk parameter feeds exactly one autograd node per step in this model;
        a repeated report for the same parameter is ignored rather than double-counted)"""
        idx = self.bucket_of.get(id(p))
        if idx is not None and id(p) not in self._reported:
            self._reported.add(id(p))
            self._ready(idx)

    def _note_stream(self, dev):
        """remember the stream a gradient was reported from (hundreds of reports per step: the raw handle is two C calls, the torch
        stream object is only built for a handle not seen yet in this step)"""
        raw = ops.raw_stream(dev)
        if raw not in self._streams:
            self._streams[raw] = torch.cuda.current_stream(dev)

    def _issue(self, idx):
        buf = self.buckets[idx][0]
        if self.dry:               # the moment this bucket's collective could start: everything reported so far has been enqueued
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(buf.device))
            self._events.append((idx, ev))
            return
        if not buf.is_cuda:
            self.handles.append((idx, dist.all_reduce(buf, group=self.pg, async_op=True)))
            return
        # gradients are written on more than one compute stream (the encoder's backward runs on vbg.ops.side_stream): the
        # collective is issued from a staging stream that waits for every stream a gradient was reported from and for the side
        # streams -- the reporting stream itself is not held up
        dev = buf.device
        if self._stage is None:
            self._stage = torch.cuda.Stream(device=dev)
        self._note_stream(dev)
        for s in list(self._streams.values()) + [t for t in ops.side_streams() if t.device == dev]:
            self._stage.wait_stream(s)
        with torch.cuda.stream(self._stage):
            self.handles.append((idx, dist.all_reduce(buf, group=self.pg, async_op=True)))

    def timeline(self, start_event):
        """dry run: [(bucket, MB, ms since start_event)] of the last step, in issue order; clears the record (call after a sync)"""
        out = [(i, self.buckets[i][0].numel() * 4 / 1e6, start_event.elapsed_time(ev)) for i, ev in self._events]
        self._events = []
        return out

    def _wait_for_buckets(self):
        """compute stream waits for the bucket collectives issued so far (serialize_syncbn).  The collectives were issued with
        async_op=True: they run on the BACKEND's stream (ProcessGroupNCCL's own), which the staging stream only joins at work.wait() --
        so the current stream has to wait for the works themselves (waiting for the staging stream alone left the buckets
        free to overlap the statistics collective).  work.wait() on NCCL blocks the current STREAM, not the host; on gloo the host.
        The handles stay listed for finish()."""
        for _, h in self.handles:
            h.wait()

    def _ready(self, idx):
        b = self.buckets[idx]
        if b[0].is_cuda:
            self._note_stream(b[0].device)
        b[2] -= 1
        assert b[2] >= 0, "a gradient was reported twice: the bucket would be reduced before it is complete"
        if b[2] != 0:
            return
        self._observed.append(idx)
        self._complete.add(idx)
        if self.order is None or not self.overlap:
            return
        while self._next < len(self.order) and self.order[self._next] in self._complete:
            self._issue(self.order[self._next])
            self._next += 1

    def finish(self):
        """issue what backward left over (in sequence), wait for every bucket (call after backward, before the optimizer
        steps), re-arm the counters"""
        if self.dry:
            if self.order is None:
                self.order = self._observed + [i for i in range(len(self.buckets)) if i not in self._complete]
            while self._next < len(self.order):
                self._issue(self.order[self._next])
                self._next += 1
            for b in self.buckets:
                b[2] = b[1]
            self._reported.clear()
            self._streams.clear()
            self._observed, self._complete, self._next = [], set(), 0
            return
        if not self.enabled:
            return
        self._phase, self._beat = "finish", time.monotonic()
        if self.order is None:
            seen = self._observed + [i for i in range(len(self.buckets)) if i not in self._complete]
            t = torch.tensor(seen, dtype=torch.int64, device=self.buckets[0][0].device)
            dist.broadcast(t, src=dist.get_global_rank(self.pg, 0) if self.pg is not None else 0, group=self.pg)
            self.order = [int(i) for i in t.tolist()]
            assert sorted(self.order) == list(range(len(self.buckets)))
        late = 0
        while self._next < len(self.order):        # a parameter got no gradient this step (zero rows): reduce what is there
            self._issue(self.order[self._next])
            self._next += 1
            late += 1
        for _, h in self.handles:
            h.wait()
        if (late and self.overlap and self.static_graph and self.sync_bn_mode != "own communicator" and self.sync_bn_mode != "none" and self.steps_done >= 1
                and Fn.SyncCtx.seq > 0 and not self.forced):
            raise RuntimeError(f"vbg.optim.FlatReducer(static_graph=True): {late} gradient bucket(s) had to be issued from finish() in step "
                               f"{self.steps_done + 1} -- this rank's autograd graph skipped a sub-module, so its collectives did not interleave "
                               "with the SyncBatchNorm statistics the way the other ranks' did.  Build the reducer with static_graph=False "
                               "(buckets after backward) or sync_bn_group='new' (own communicator for the statistics).")
        self.handles = []
        for b in self.buckets:
            b[2] = b[1]
        self._reported.clear()
        self._streams.clear()
        self._observed, self._complete, self._next = [], set(), 0
        self.steps_done += 1
        self._phase, self._beat = "between steps", time.monotonic()

    # ------------------------------------------------------------------------------------------------------------------------
    def describe_pending(self) -> str:
        """one line for a hang report: which bucket / which SyncBatchNorm collective this rank is at"""
        rank = dist.get_rank() if dist.is_initialized() else 0
        pend = []
        for idx, h in self.handles:
            try:
                done = h.is_completed()
            except Exception:
                done = None
            if not done:
                pend.append(idx)
        nxt = self.order[self._next] if (self.order is not None and self._next < len(self.order)) else None
        waiting = [i for i in range(len(self.buckets)) if self.buckets[i][2] > 0]
        return (f"[vbg reducer] rank {rank}: step {self.steps_done} ({self._phase}), {self.sync_bn_mode if self.enabled else 'disabled'}, "
                f"overlap {'on' if self.overlap else 'off'}; buckets issued {self._next}/{len(self.buckets)}, in flight {pend}, next in sequence {nxt}, "
                f"still collecting gradients {waiting[:8]}{'...' if len(waiting) > 8 else ''}; SyncBatchNorm collectives issued so far "
                f"{getattr(Fn.SyncCtx, 'seq', 0)} (last: {getattr(Fn.SyncCtx, 'last', None)})")

    def start_watchdog(self, seconds: float, exit_code=None):
        """daemon thread: when no step completes for `seconds`, print `describe_pending()` to stderr (once per stall) so that a hung
        collective leaves a diagnostic instead of an empty record; exit_code != None then ends the process"""
        import sys
        import threading

        def run():
            reported = -1
            while True:
                time.sleep(min(5.0, seconds / 4))
                if time.monotonic() - self._beat > seconds and reported != self.steps_done:
                    reported = self.steps_done
                    print(self.describe_pending() + f" -- no progress for {seconds:.0f} s", file=sys.stderr, flush=True)
                    if exit_code is not None:
                        os._exit(exit_code)

        self._watch = threading.Thread(target=run, daemon=True, name="vbg-reducer-watchdog")
        self._watch.start()
