"""CPU oracle for the ViBERTgrid joint forward/backward step.

TEST INFRASTRUCTURE ONLY.  This file is a from-scratch CPU restatement (plain
torch-CPU + numpy) of the algorithm on the hot path of ZeningLin/ViBERTgrid-PyTorch
(`model/ViBERTgrid_net.py::ViBERTgridNet.forward` and everything below it).  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it;
the product path (`vibertgrid-pytorch_amd/`) never does and fails loudly when the HIP
library is missing.

Pinning: the reference ships no tests/golden vectors.  The oracle is pinned
against outputs of the reference itself, imported in the build container by
`tests/golden/make_golden.py` (fixtures under `tests/golden/*.npz`).  One piece stays
"parity unpinned" by the reference: `torchvision.ops.RoIAlign` (torchvision 0.14.1 is not
installed anywhere here) — `roi_align` below follows the published Mask R-CNN / torchvision
`roi_align` definition and is pinned by hand-computable known-answer tests
(tests/test_oracle_roi_align.py).

Everything is functional: parameters come from a flat dict `sd` with the reference's
state_dict key names, so the same weights drive the reference, the oracle
and the HIP product.

All `file:line` citations are into /root/reference.
"""
from __future__ import annotations

import math
import random as _pyrandom
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------
@dataclass
class BertCfg:
    """The slice of a HF BertConfig / RobertaConfig the encoder arithmetic needs."""
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    intermediate: int = 3072
    ln_eps: float = 1e-12
    roberta: bool = False          # RoBERTa position-id rule (padding_idx = 1)
    dropout: float = 0.1


@dataclass
class NetCfg:
    num_classes: int = 5
    image_mean: Sequence[float] = (0.9248, 0.9224, 0.9215)
    image_std: Sequence[float] = (0.1532, 0.1545, 0.1536)
    image_min_size: Sequence[int] = (512,)
    image_max_size: int = 512
    test_image_min_size: int = 512
    backbone: str = "resnet_34_fpn"         # resnet_{18,34}_fpn[_pretrained]
    grid_mode: str = "mean"
    stride: int = 8                          # early_fusion_downsampling_ratio
    roi_shape: int = 7
    p_fuse_stride: int = 4                   # p_fuse_downsampling_ratio
    loss_weights: Optional[Sequence[float]] = None
    num_hard_positive_main_1: int = 16
    num_hard_negative_main_1: int = 16
    num_hard_positive_main_2: int = 32
    num_hard_negative_main_2: int = 32
    loss_aux_sample_list: Optional[Sequence[int]] = (256, 512, 256)
    num_hard_positive_aux: int = 256
    num_hard_negative_aux: int = 256
    ohem_random: bool = True
    loss_control_lambda: float = 1.0
    add_pos_neg: bool = True
    classifier_mode: str = "simp"            # simp | full | crf
    layer_mode: str = "multi"                # BinaryClassifier / CRF emission net: single | multi
    bert: BertCfg = field(default_factory=BertCfg)


# --------------------------------------------------------------------------------------
# a1. transform  (pipeline/transform.py:104-171, 225-312)
# --------------------------------------------------------------------------------------
def resize_scale(h: int, w: int, min_size: float, max_size: float) -> float:
    """Scale factor of `_resize_image` (pipeline/transform.py:133-157)."""
    mn, mx = float(min(h, w)), float(max(h, w))
    s = min_size / mn
    if mx * s > max_size:
        s = max_size / mx
    return s


def resized_shape(h: int, w: int, scale: float) -> Tuple[int, int]:
    """Output size of F.interpolate(scale_factor=s, recompute_scale_factor=True):
    floor(in * s) computed in double precision."""
    return int(math.floor(float(h) * scale)), int(math.floor(float(w) * scale))


def rescale_boxes(coor: Tensor, orig: Tuple[int, int], new: Tuple[int, int]) -> Tensor:
    """`_resize_labels` (pipeline/transform.py:159-171).  Columns 0/2 (x) are scaled by the
    HEIGHT ratio and 1/3 (y) by the WIDTH ratio (the reference's swap), in fp32, then
    truncated toward zero to int32."""
    rh = new[0] / orig[0]
    rw = new[1] / orig[1]
    c = coor.to(torch.float32).clone()
    c[:, [0, 2]] *= rh
    c[:, [1, 3]] *= rw
    return c.to(torch.int32)


def transform(images: Sequence[Tensor], coors: Sequence[Tensor], cfg: NetCfg, training: bool,
              min_sizes: Optional[Sequence[int]] = None):
    """`GeneralizedViBERTgridTransform.forward` (pipeline/transform.py:273-312).

    `min_sizes`: the per-image min side already chosen (the reference draws it with torch's
    global CPU RNG in `torch_choice`, :124-131); None -> drawn here the same way in training,
    `test_image_min_size` in eval.  Returns (batch [B,3,H,W] fp32 zero padded to /32,
    list of int32 boxes, list of resized (h, w))."""
    mean = torch.tensor(list(cfg.image_mean), dtype=torch.float32)[:, None, None]
    std = torch.tensor(list(cfg.image_std), dtype=torch.float32)[:, None, None]
    out_imgs, out_coors, sizes = [], [], []
    for i, (img, coor) in enumerate(zip(images, coors)):
        if img.dim() != 3:
            raise ValueError("images is expected to be a list of 3d tensors of shape [C, H, W], got {}".format(img.shape))
        x = (img - mean) / std
        if min_sizes is not None:
            size = float(min_sizes[i])
        elif training:
            k = list(cfg.image_min_size)
            size = float(k[int(torch.empty(1).uniform_(0.0, float(len(k))).item())])
        else:
            size = float(cfg.test_image_min_size)
        h, w = x.shape[-2:]
        s = resize_scale(h, w, size, float(cfg.image_max_size))
        x = F.interpolate(x[None], scale_factor=s, mode="bilinear", recompute_scale_factor=True,
                          align_corners=False)[0]
        nh, nw = x.shape[-2:]
        out_imgs.append(x)
        out_coors.append(rescale_boxes(coor, (h, w), (nh, nw)))
        sizes.append((nh, nw))
    H = int(math.ceil(max(s[0] for s in sizes) / 32.0) * 32)
    W = int(math.ceil(max(s[1] for s in sizes) / 32.0) * 32)
    batch = torch.zeros((len(out_imgs), 3, H, W), dtype=torch.float32)
    for b, x in enumerate(out_imgs):
        batch[b, :, : x.shape[1], : x.shape[2]] = x
    return batch, out_coors, sizes


# --------------------------------------------------------------------------------------
# a2. sliding windows  (model/BERTgrid_generator.py:78-146)
# --------------------------------------------------------------------------------------
def bert_windows(corpus: Tensor, mask: Tensor):
    """Build the reference's [CLS] .. [SEP] [PAD].. windows.

    Returns a list of (input_ids int64 [B,Lw], attention_mask int64 [B,Lw], curr_len).
    Lw is 512 for every window (the last one is padded with id 0 / mask 0 to 512); the [SEP]
    sits at column 1+curr_len, i.e. after the *batch-max* length, so shorter documents have
    `real.., 0(mask 0).., SEP(mask 1)`.  CLS/SEP/PAD ids are the hard-coded 101/102/0."""
    B, T = corpus.shape
    nwin = T // 510 + 1
    wins = []
    start = 0
    cls = torch.full((B, 1), 101, dtype=torch.long)
    sep = torch.full((B, 1), 102, dtype=torch.long)
    one = torch.ones((B, 1), dtype=torch.long)
    for c in range(nwin):
        end = (c + 1) * 510
        if end > T:
            seq = corpus[:, start:].long()
            msk = mask[:, start:].long()
            cur = seq.shape[1]
            pad = torch.zeros((B, end - T), dtype=torch.long)
            ids = torch.cat([cls, seq, sep, pad], 1)
            am = torch.cat([one, msk, one, pad], 1)
        else:
            seq = corpus[:, start:end].long()
            msk = mask[:, start:end].long()
            cur = seq.shape[1]
            ids = torch.cat([cls, seq, sep], 1)
            am = torch.cat([one, msk, one], 1)
        wins.append((ids, am, cur))
        start = end
    return wins


# --------------------------------------------------------------------------------------
# a3. BERT / RoBERTa encoder (transformers 4.36.0 BertModel / RobertaModel, third party:
#     modeling_bert.py BertEmbeddings / BertSelfAttention / BertSelfOutput / BertIntermediate /
#     BertOutput; called at model/BERTgrid_generator.py:134)
# --------------------------------------------------------------------------------------
def bert_position_ids(input_ids: Tensor, bc: BertCfg) -> Tensor:
    L = input_ids.shape[1]
    if not bc.roberta:
        return torch.arange(L, dtype=torch.long)[None].expand_as(input_ids)
    # RobertaEmbeddings.create_position_ids_from_input_ids, padding_idx = 1
    m = (input_ids != 1).long()
    return torch.cumsum(m, 1) * m + 1


def bert_forward(sd: Dict[str, Tensor], prefix: str, input_ids: Tensor, attn_mask: Tensor,
                 bc: BertCfg, train: bool = False) -> Tensor:
    """last_hidden_state [B,L,hidden] of a HF BertModel/RobertaModel (pooler unused)."""
    p = prefix
    pos = bert_position_ids(input_ids, bc)
    x = (sd[p + "embeddings.word_embeddings.weight"][input_ids]
         + sd[p + "embeddings.token_type_embeddings.weight"][torch.zeros_like(input_ids)]
         + sd[p + "embeddings.position_embeddings.weight"][pos])
    x = F.layer_norm(x, (bc.hidden,), sd[p + "embeddings.LayerNorm.weight"],
                     sd[p + "embeddings.LayerNorm.bias"], bc.ln_eps)
    x = F.dropout(x, bc.dropout, train)
    B, L, _ = x.shape
    dh = bc.hidden // bc.heads
    ext = (1.0 - attn_mask[:, None, None, :].to(x.dtype)) * torch.finfo(x.dtype).min
    for i in range(bc.layers):
        lp = f"{p}encoder.layer.{i}."
        def lin(t, name):
            return F.linear(t, sd[lp + name + ".weight"], sd[lp + name + ".bias"])
        q = lin(x, "attention.self.query").view(B, L, bc.heads, dh).transpose(1, 2)
        k = lin(x, "attention.self.key").view(B, L, bc.heads, dh).transpose(1, 2)
        v = lin(x, "attention.self.value").view(B, L, bc.heads, dh).transpose(1, 2)
        s = q @ k.transpose(-1, -2) / math.sqrt(dh) + ext
        pr = F.dropout(torch.softmax(s, -1), bc.dropout, train)
        ctx = (pr @ v).transpose(1, 2).reshape(B, L, bc.hidden)
        a = F.dropout(lin(ctx, "attention.output.dense"), bc.dropout, train)
        x = F.layer_norm(a + x, (bc.hidden,), sd[lp + "attention.output.LayerNorm.weight"],
                         sd[lp + "attention.output.LayerNorm.bias"], bc.ln_eps)
        h = F.gelu(lin(x, "intermediate.dense"))
        o = F.dropout(lin(h, "output.dense"), bc.dropout, train)
        x = F.layer_norm(o + x, (bc.hidden,), sd[lp + "output.LayerNorm.weight"],
                         sd[lp + "output.LayerNorm.bias"], bc.ln_eps)
    return x


def bert_embedding(sd, prefix, corpus: Tensor, mask: Tensor, bc: BertCfg, train=False) -> Tensor:
    """Token embeddings [B,T,hidden]: windows -> encoder -> strip specials -> concat
    (model/BERTgrid_generator.py:99-146)."""
    outs = []
    for ids, am, cur in bert_windows(corpus, mask):
        h = bert_forward(sd, prefix, ids, am, bc, train)
        outs.append(h[:, 1:1 + cur])
    return torch.cat(outs, 1)


# --------------------------------------------------------------------------------------
# a4. token -> segment aggregation  (model/BERTgrid_generator.py:148-191)
# --------------------------------------------------------------------------------------
def seg_runs(seg_indices: Tensor) -> Tuple[np.ndarray, np.ndarray]:
    """Maximal runs of equal consecutive seg_indices: (start offsets, lengths)."""
    s = seg_indices.cpu().numpy().astype(np.int64)
    if s.size == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    brk = np.flatnonzero(np.diff(s) != 0) + 1
    starts = np.concatenate([[0], brk])
    lens = np.diff(np.concatenate([starts, [s.size]]))
    return starts, lens


def seg_aggregate(emb: Tensor, mask_row: Tensor, seg_indices: Tensor, mode: str = "mean") -> Tensor:
    """One document: rows with mask==1 -> [S,hidden].  `mean` is the sequential fp sum in token
    order followed by a division by the count (bit-exact to the reference's in-place `+=`,
    `/=` chain: ((e0+e1)+e2)/3); `first` takes the first token of each run."""
    e = emb[mask_row == 1]
    assert e.shape[0] == seg_indices.shape[0]
    starts, lens = seg_runs(seg_indices)
    rows = []
    for st, ln in zip(starts, lens):
        if mode == "first":
            rows.append(e[st])
        else:
            acc = e[st]
            for j in range(1, ln):
                acc = acc + e[st + j]
            rows.append(acc / float(ln))
    return torch.stack(rows, 0)


# --------------------------------------------------------------------------------------
# a5. bbox -> grid scatter  (model/BERTgrid_generator.py:193-245) and the owner map shared
#     with the seg-head label rasterisation (model/semantic_segmentation_head.py:326-341)
# --------------------------------------------------------------------------------------
def _slice_bounds(lo: int, hi: int, n: int) -> Tuple[int, int]:
    """Python slice clipping `a[lo:hi]` on an axis of length n (negative indices wrap once)."""
    st, en, _ = slice(lo, hi).indices(n)
    return st, max(st, en)


def _tdiv(v: int, d: int) -> int:
    """int(v / d) for python ints: true division then truncation toward zero."""
    return int(v / d)


def owner_map(boxes: np.ndarray, gh: int, gw: int, stride: int) -> np.ndarray:
    """int32 [gh,gw]: index of the LAST segment (in order) whose rectangle
    rows int(y1/stride):int(y2/stride), cols int(x1/stride):int(x2/stride) covers the cell,
    -1 where none does.  Last-writer-wins == the reference's sequential slice assignment."""
    own = np.full((gh, gw), -1, np.int32)
    for s in range(boxes.shape[0]):
        x1, y1, x2, y2 = (int(v) for v in boxes[s])
        r0, r1 = _slice_bounds(_tdiv(y1, stride), _tdiv(y2, stride), gh)
        c0, c1 = _slice_bounds(_tdiv(x1, stride), _tdiv(x2, stride), gw)
        own[r0:r1, c0:c1] = s
    return own


def grid_scatter(embs: Sequence[Tensor], coors: Sequence[Tensor], H: int, W: int, stride: int = 8) -> Tensor:
    """BERTgrid fp32 [B,C,int(H/stride),int(W/stride)] (always fp32, :220-228)."""
    B = len(embs)
    C = embs[0].shape[-1]
    gh, gw = int(H / stride), int(W / stride)
    grid = torch.zeros((B, C, gh, gw), dtype=torch.float32)
    for b in range(B):
        assert embs[b].shape[0] == coors[b].shape[0]
        if embs[b].shape[0] == 0:
            continue
        own = torch.from_numpy(owner_map(coors[b].cpu().numpy(), gh, gw, stride)).long()
        covered = own >= 0
        idx = own.clamp(min=0)
        g = embs[b].to(torch.float32)[idx]                      # [gh,gw,C] (autograd: index -> sums)
        g = g * covered[..., None].to(g.dtype)
        grid[b] = g.permute(2, 0, 1)
    return grid


def label_raster(seg_classes: Sequence[Tensor], coors: Sequence[Tensor], H: int, W: int):
    """pos_neg (0 bg / 1 key / 2 non-key) and class labels, int64 [B,H,W], stride-1
    last-writer-wins raster (model/semantic_segmentation_head.py:314-341)."""
    B = len(coors)
    pos_neg = torch.zeros((B, H, W), dtype=torch.long)
    cls = torch.zeros((B, H, W), dtype=torch.long)
    for b in range(B):
        own = torch.from_numpy(owner_map(coors[b].cpu().numpy(), H, W, 1)).long()
        cov = own >= 0
        c = seg_classes[b].long()[own.clamp(min=0)]
        cls[b] = torch.where(cov, c, torch.zeros_like(c))
        pos_neg[b] = torch.where(cov, torch.where(c > 0, 1, 2), torch.zeros_like(c))
    return pos_neg, cls


# --------------------------------------------------------------------------------------
# a6-a8. ResNet-FPN backbone with early fusion  (model/ResNetFPN_ViBERTgrid.py)
# --------------------------------------------------------------------------------------
BN_FROZEN = False      # True: every BatchNorm uses its running statistics even in a training step (modules put in eval() mode)


def _bn(sd, name, x, train, momentum=0.1, eps=1e-5):
    train = train and not BN_FROZEN
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"],
                        sd[name + ".bias"], train, momentum, eps)


def _basic_block(sd, p, x, train, stride, names):
    """BasicBlock (:106-184) / DBlock (:187-269) / torchvision BasicBlock.  `names` maps logical -> key suffix;
    the ResNet-D block's projection shortcut is AvgPool2d(2,2) -> 1x1 conv (stride 1) -> BN (:222-236)."""
    c1, b1, c2, b2, sc_conv, sc_bn = names
    y = F.conv2d(x, sd[p + c1 + ".weight"], None, stride, 1)
    y = F.relu(_bn(sd, p + b1, y, train))
    y = F.conv2d(y, sd[p + c2 + ".weight"], None, 1, 1)
    y = _bn(sd, p + b2, y, train)
    if (p + sc_conv + ".weight") in sd:
        if names is _OWN_D:
            s = F.conv2d(F.avg_pool2d(x, 2, 2), sd[p + sc_conv + ".weight"], None, 1, 0)
        else:
            s = F.conv2d(x, sd[p + sc_conv + ".weight"], None, stride, 0)
        s = _bn(sd, p + sc_bn, s, train)
    else:
        s = x
    return F.relu(y + s)


_OWN = ("conv_1", "bn_1", "conv_2", "bn_2", "conv_shortcut.0", "conv_shortcut.1")
_OWN_D = ("conv_1", "bn_1", "conv_2", "bn_2", "conv_shortcut.1", "conv_shortcut.2")      # Sequential(AvgPool2d, Conv2d, BatchNorm2d)
_TV = ("conv1", "bn1", "conv2", "bn2", "downsample.0", "downsample.1")


def backbone_forward(sd, x: Tensor, grid: Tensor, kind: str, train: bool, prefix="backbone.") -> Tensor:
    """P_fuse [B,256,H/4,W/4].  kind in resnet_{18,34}_fpn[_pretrained] | resnet_{18,34}_D_fpn."""
    _OWN = _OWN_D if "_D_" in kind else globals()["_OWN"]
    sizes = [2, 2, 2, 2] if "18" in kind else [3, 4, 6, 3]
    p = prefix
    if kind.endswith("_pretrained"):
        r = p + "resnet."
        x1 = F.conv2d(x, sd[r + "conv1.weight"], None, 2, 3)
        x1 = F.relu(_bn(sd, r + "bn1", x1, train))
        x1 = F.max_pool2d(x1, 3, 2, 1)
        for i in range(sizes[0]):
            x1 = _basic_block(sd, f"{r}layer1.{i}.", x1, train, 1, _TV)
        x2 = _basic_block(sd, f"{r}layer2.0.", x1, train, 2, _TV)
        x2 = F.conv2d(torch.cat((x2, grid), 1), sd[p + "early_fusion.weight"], None)   # bias=False (:529-535)
        for i in range(1, sizes[1]):
            x2 = _basic_block(sd, f"{r}layer2.{i}.", x2, train, 1, _TV)
        x3 = x2
        for i in range(sizes[2]):
            x3 = _basic_block(sd, f"{r}layer3.{i}.", x3, train, 2 if i == 0 else 1, _TV)
        x4 = x3
        for i in range(sizes[3]):
            x4 = _basic_block(sd, f"{r}layer4.{i}.", x4, train, 2 if i == 0 else 1, _TV)
    else:
        x1 = F.conv2d(x, sd[p + "conv_1.0.weight"], None, 2, 3)
        x1 = F.relu(_bn(sd, p + "conv_1.1", x1, train))
        x1 = F.max_pool2d(x1, 3, 2, 1)
        for i in range(sizes[0]):
            x1 = _basic_block(sd, f"{p}conv_2_x.{i}.", x1, train, 1, _OWN)
        x2 = _basic_block(sd, p + "conv_3_x.block_1.", x1, train, 2, _OWN)
        x2 = F.conv2d(torch.cat((x2, grid), 1), sd[p + "conv_3_x.early_fusion.weight"],
                      sd[p + "conv_3_x.early_fusion.bias"])                              # bias=True (:305-309)
        for i in range(sizes[1] - 1):
            x2 = _basic_block(sd, f"{p}conv_3_x.layers.{i}.", x2, train, 1, _OWN)
        x3 = x2
        for i in range(sizes[2]):
            x3 = _basic_block(sd, f"{p}conv_4_x.{i}.", x3, train, 2 if i == 0 else 1, _OWN)
        x4 = x3
        for i in range(sizes[3]):
            x4 = _basic_block(sd, f"{p}conv_5_x.{i}.", x4, train, 2 if i == 0 else 1, _OWN)
    # FPN top-down (:488-508 / :628-648); all convs bias-free, nearest upsampling
    up = lambda t, s: F.interpolate(t, scale_factor=s, mode="nearest")
    x4 = F.conv2d(x4, sd[p + "conv_6_x.weight"])
    x5 = F.conv2d(up(x4, 2) + F.conv2d(x3, sd[p + "skip_1.weight"]), sd[p + "merge_1.weight"], None, 1, 1)
    x6 = F.conv2d(up(x5, 2) + F.conv2d(x2, sd[p + "skip_2.weight"]), sd[p + "merge_2.weight"], None, 1, 1)
    x7 = F.conv2d(up(x6, 2) + F.conv2d(x1, sd[p + "skip_3.weight"]), sd[p + "merge_3.weight"], None, 1, 1)
    cat = torch.cat([up(x4, 8), up(x5, 4), up(x6, 2), x7], 1)
    return F.conv2d(cat, sd[p + "fuse.weight"])


# --------------------------------------------------------------------------------------
# a13. losses  (pipeline/custom_loss.py)
# --------------------------------------------------------------------------------------
def ce_random_sample(logits: Tensor, target: Tensor, sample_list: Optional[Sequence[int]],
                     weight: Optional[Tensor] = None) -> Tensor:
    """CrossEntropyLossRandomSample, reduction='mean' (pipeline/custom_loss.py:35-101).
    Result is float64 shape [1].  Consumes the GLOBAL python `random` state like the
    reference (random.sample per category that has >= k elements, in category order)."""
    if sample_list is None:
        return F.cross_entropy(logits.float(), target, weight=weight)
    ce = F.cross_entropy(logits.float(), target, weight=weight, reduction="none")
    ncat = len(sample_list)
    if ncat == 2 and logits.shape[1] >= 2:
        masks = [target == 0, target != 0]
    else:
        assert ncat == logits.shape[1]
        masks = [target == c for c in range(ncat)]
    total = torch.zeros((1,), dtype=torch.float64)
    nkeep_total = 0
    for k, m in zip(sample_list, masks):
        cur = ce[m]
        nkeep = min(k, cur.shape[0])
        nkeep_total += nkeep
        if nkeep == k:
            idx = torch.tensor(_pyrandom.sample(range(int(cur.shape[0])), nkeep), dtype=torch.long)
            cur = cur[idx]
        total = total + cur.sum()
    return total / nkeep_total


# `torch.sort(..., descending=True)` in the reference is NOT stable, and with exactly tied losses
# (pervasive in the seg head: the x4 nearest upsampling replicates logits over 4x4 pixels) the
# quirk below makes the loss VALUE depend on the tie order.  False reproduces the reference's CPU
# behaviour (what the golden fixtures pin); the HIP product sorts stably (LSD radix sort) and is
# compared with the oracle under OHEM_STABLE_SORT = True.  See DESIGN.md "OHEM ties".
OHEM_STABLE_SORT = False


def ce_ohem(logits: Tensor, target: Tensor, npos: int, nneg: int, weight: Optional[Tensor] = None,
            rand: bool = False) -> Tensor:
    """CrossEntropyLossOHEM, reduction='mean' (pipeline/custom_loss.py:127-201), including the
    reference's quirk of indexing the SORTED losses with the ORIGINAL positions of the top-k
    (`sorted_loss[sorted_index[:k]]`, :175-176, :185-186).  0-dim fp32 result."""
    if npos == -1 and nneg == -1:
        return F.cross_entropy(logits.float(), target, weight=weight)
    ce = F.cross_entropy(logits.float(), target, weight=weight, reduction="none")
    m = target == 0
    pos, neg = ce[~m], ce[m]
    if rand:
        if 2 * npos < pos.shape[0]:
            pos = pos[torch.tensor(_pyrandom.sample(range(int(pos.shape[0])), 2 * npos), dtype=torch.long)]
        if 2 * nneg < neg.shape[0]:
            neg = neg[torch.tensor(_pyrandom.sample(range(int(neg.shape[0])), 2 * nneg), dtype=torch.long)]

    def pick(v, k):
        sv, si = torch.sort(v, descending=True, stable=OHEM_STABLE_SORT)
        keep = min(sv.shape[0], k)
        if 0 < keep < sv.shape[0]:
            sv = sv[si[:keep]]
        return sv, keep

    sp, kp = pick(pos, npos)
    sn, kn = pick(neg, nneg)
    return (sp.sum() + sn.sum()) / (kp + kn)


def bce_random_sample(logit: Tensor, target: Tensor, sample_list: Optional[Sequence[int]]) -> Tensor:
    """BCELossRandomSample, reduction='mean' (pipeline/custom_loss.py:204-290).  The two categories are split by the SIGN OF
    THE PREDICTION (`mask = input > 0`, :250), category 0 = input <= 0 sampled with sample_list[0], category 1 = input > 0
    with sample_list[1]; python `random.sample` only when the category holds >= k elements.  float64 [1]."""
    if logit.dim() == 2:
        logit = logit.squeeze(1)
    if sample_list is None:
        return F.binary_cross_entropy_with_logits(logit.float(), target)
    ce = F.binary_cross_entropy_with_logits(logit.float(), target, reduction="none")
    m = logit > 0
    total = torch.zeros((1,), dtype=torch.float64)
    kept = 0
    for idx in range(2):
        cur = ce[~m] if idx == 0 else ce[m]
        k = sample_list[idx]
        keep = min(k, cur.shape[0])
        kept += keep
        if keep == k:
            cur = cur[torch.tensor(_pyrandom.sample(range(int(cur.shape[0])), keep), dtype=torch.long)]
        total = total + cur.sum()
    return total / kept


def bce_ohem(logit: Tensor, target: Tensor, npos: int, nneg: int, rand: bool = False) -> Tensor:
    """BCELossOHEM, reduction='mean' (pipeline/custom_loss.py:293-382): positives = target != 0; optional random pre-sample of
    2k; descending sort; the same `sorted_loss[sorted_index[:k]]` quirk as the CE version (:344-346, :354-356); when the keep
    count is <= 0 the whole category stays in the sum while the denominator adds that count.  0-dim fp32."""
    if npos == -1 and nneg == -1:
        return F.binary_cross_entropy_with_logits(logit.float(), target)
    ce = F.binary_cross_entropy_with_logits(logit.float(), target, reduction="none")
    m = target == 0
    pos, neg = ce[~m], ce[m]
    if rand:
        if 2 * npos < pos.shape[0]:
            pos = pos[torch.tensor(_pyrandom.sample(range(int(pos.shape[0])), 2 * npos), dtype=torch.long)]
        if 2 * nneg < neg.shape[0]:
            neg = neg[torch.tensor(_pyrandom.sample(range(int(neg.shape[0])), 2 * nneg), dtype=torch.long)]

    def pick(v, k):
        sv, si = torch.sort(v, descending=True, stable=OHEM_STABLE_SORT)
        keep = min(sv.shape[0], k)
        if 0 < keep < sv.shape[0]:
            sv = sv[si[:keep]]
        return sv, keep

    sp, kp = pick(pos, npos)
    sn, kn = pick(neg, nneg)
    return (sp.sum() + sn.sum()) / (kp + kn)


# --------------------------------------------------------------------------------------
# a9. auxiliary semantic segmentation head (simp)  (model/semantic_segmentation_head.py:66-78,
#     288-352)
# --------------------------------------------------------------------------------------
def seg_head_logits(sd, p_fuse: Tensor, train: bool, prefix="semantic_segmentation_head.semantic_segmentation_encoder."):
    p = prefix
    x = F.relu(_bn(sd, p + "bn_1", F.conv2d(p_fuse, sd[p + "conv_1.weight"], None, 1, 1), train))
    x = F.relu(_bn(sd, p + "bn_2", F.conv2d(x, sd[p + "conv_2.weight"], None, 1, 1), train))
    x = F.interpolate(x, scale_factor=4, mode="nearest")
    x1 = F.conv2d(x, sd[p + "conv_3_1.weight"], sd[p + "conv_3_1.bias"])
    x2 = F.conv2d(x, sd[p + "conv_3_2.weight"], sd[p + "conv_3_2.bias"])
    return x1, x2


def seg_head(sd, p_fuse, seg_classes, coors, cfg: NetCfg, train: bool):
    x1, x2 = seg_head_logits(sd, p_fuse, train)
    H, W = x1.shape[-2:]
    pos_neg, cls = label_raster(seg_classes, coors, H, W)
    w = None if cfg.loss_weights is None else torch.tensor(list(cfg.loss_weights), dtype=torch.float32)
    l1 = ce_random_sample(x1, pos_neg, cfg.loss_aux_sample_list)        # aux_loss_1 never weighted (:279-283)
    l2 = ce_ohem(x2, cls, cfg.num_hard_positive_aux, cfg.num_hard_negative_aux, w, rand=False)
    return l1 + l2, x1, x2


def seg_head_full(sd, p_fuse, seg_classes, coors, cfg: NetCfg, train: bool, prefix="semantic_segmentation_head."):
    """SemanticSegmentationClassifier, the two-stage variant used by classifier_mode full / crf
    (model/semantic_segmentation_head.py:100-233): aux_loss_1 as in the simplified head; then, on the pixels PREDICTED
    positive (`softmax(x_out_1).argmax(1) == 1`), one 1x1 conv (ncls -> 1) per foreground class on x_out_2 with a BCE-OHEM loss
    (never the random pre-sample, :138-153) against `class_label == idx + 1`."""
    x1, x2 = seg_head_logits(sd, p_fuse, train, prefix + "ss_encoder.")
    H, W = x1.shape[-2:]
    pos_neg, cls = label_raster(seg_classes, coors, H, W)
    l1 = ce_random_sample(x1, pos_neg, cfg.loss_aux_sample_list)
    pos_mask = x1.softmax(dim=1).argmax(dim=1) == 1
    l2 = torch.zeros((1,))
    if int(pos_mask.int().sum()) != 0:
        for ci in range(cfg.num_classes - 1):
            q = f"{prefix}ss_binary_classifier_{ci}.conv1."
            pred = F.conv2d(x2, sd[q + "weight"], sd[q + "bias"])[pos_mask.unsqueeze(1)]
            lab = (cls[pos_mask] == (ci + 1)).float()
            l2 = l2 + bce_ohem(pred, lab, cfg.num_hard_positive_aux, cfg.num_hard_negative_aux, False)
    return l1 + l2, x1, x2


# --------------------------------------------------------------------------------------
# a10. RoIAlign  (torchvision 0.14.1 `torchvision.ops.RoIAlign(7, 1/4, sampling_ratio=-1)`,
#      aligned=False; call site model/grid_roi_align.py:37-41,81) -- restated from the
#      published algorithm; parity unpinned by the reference, pinned by KATs.
# --------------------------------------------------------------------------------------
def roi_align(feat: Tensor, boxes: Sequence[Tensor], out_size: int = 7, spatial_scale: float = 0.25) -> Tensor:
    """feat [B,C,H,W]; boxes: per-image float [S,4] (x1,y1,x2,y2 image coords).
    -> [sum S, C, out, out].  Differentiable w.r.t. feat (index/gather ops)."""
    B, C, H, W = feat.shape
    outs = []
    ph = pw = out_size
    for b, bx in enumerate(boxes):
        bx = bx.to(torch.float32)
        for r in range(bx.shape[0]):
            x1, y1, x2, y2 = (np.float32(float(bx[r, j])) * np.float32(spatial_scale) for j in range(4))
            roi_w = max(np.float32(x2 - x1), np.float32(1.0))
            roi_h = max(np.float32(y2 - y1), np.float32(1.0))
            bin_h = np.float32(roi_h / np.float32(ph))
            bin_w = np.float32(roi_w / np.float32(pw))
            gh = int(math.ceil(float(roi_h) / ph))
            gw = int(math.ceil(float(roi_w) / pw))
            cnt = max(gh * gw, 1)
            # sample coordinates, fp32 like the kernel
            iy = np.arange(gh, dtype=np.float32)
            ix = np.arange(gw, dtype=np.float32)
            ys = (np.float32(y1) + np.arange(ph, dtype=np.float32)[:, None] * bin_h
                  + (iy[None, :] + np.float32(0.5)) * bin_h / np.float32(gh)).astype(np.float32)      # [ph,gh]
            xs = (np.float32(x1) + np.arange(pw, dtype=np.float32)[:, None] * bin_w
                  + (ix[None, :] + np.float32(0.5)) * bin_w / np.float32(gw)).astype(np.float32)      # [pw,gw]
            wy0, wy1, y_lo, y_hi, vy = _bilinear_axis(ys.reshape(-1), H)
            wx0, wx1, x_lo, x_hi, vx = _bilinear_axis(xs.reshape(-1), W)
            f = feat[b]                                                       # [C,H,W]
            # gather rows then cols
            def t(a, dt=torch.float32):
                return torch.from_numpy(np.ascontiguousarray(a)).to(dt)
            fy = (f[:, t(y_lo, torch.long), :] * t(wy0 * vy)[None, :, None]
                  + f[:, t(y_hi, torch.long), :] * t(wy1 * vy)[None, :, None])           # [C, ph*gh, W]
            fxy = (fy[:, :, t(x_lo, torch.long)] * t(wx0 * vx)[None, None, :]
                   + fy[:, :, t(x_hi, torch.long)] * t(wx1 * vx)[None, None, :])         # [C, ph*gh, pw*gw]
            fxy = fxy.view(C, ph, gh, pw, gw).sum((2, 4)) / float(cnt)
            outs.append(fxy)
    if not outs:
        return feat.new_zeros((0, C, ph, pw))
    return torch.stack(outs, 0)


def _bilinear_axis(c: np.ndarray, size: int):
    """Per-axis bilinear taps of torchvision's `bilinear_interpolate`: coordinate < -1 or > size
    -> contributes 0; clamp to >= 0; low >= size-1 -> low = high = size-1, frac 0."""
    c = c.astype(np.float32)
    valid = ~((c < -1.0) | (c > float(size)))
    cc = np.maximum(c, np.float32(0.0))
    lo = np.floor(cc).astype(np.int64)
    edge = lo >= size - 1
    lo = np.where(edge, size - 1, lo)
    hi = np.where(edge, size - 1, lo + 1)
    cc = np.where(edge, lo.astype(np.float32), cc)
    l = (cc - lo.astype(np.float32)).astype(np.float32)
    h = (np.float32(1.0) - l).astype(np.float32)
    lo = np.where(valid, lo, 0)
    hi = np.where(valid, hi, 0)
    return h, l, lo, hi, valid.astype(np.float32)


# --------------------------------------------------------------------------------------
# a11. late fusion  (model/field_type_classification_head.py:64-75, 164-190)
# --------------------------------------------------------------------------------------
def late_fusion(sd, roi: Tensor, bert_embs: Sequence[Tensor], train: bool, prefix="late_fusion_net."):
    p = prefix + "ROI_embedding_net."
    x = F.relu(_bn(sd, p + "bn_1", F.conv2d(roi, sd[p + "conv_1.weight"], None, 1, 1), train))
    x = F.relu(_bn(sd, p + "bn_2", F.conv2d(x, sd[p + "conv_2.weight"], None, 1, 1), train))
    x = F.linear(x.flatten(1), sd[p + "linear.weight"], sd[p + "linear.bias"])
    be = torch.cat(list(bert_embs), 0)
    assert x.shape[0] == be.shape[0]
    f = torch.cat((x, be), 1)
    q = prefix + "fuse_embedding_net.linear."
    return F.linear(f, sd[q + "weight"], sd[q + "bias"])


# --------------------------------------------------------------------------------------
# a12. simplified field-type classification head  (:530-588; both nets are always the 2-layer
#      MLP because of the "sigle" typo at :474)
# --------------------------------------------------------------------------------------
def _mlp(sd, p, x):
    h = F.relu(F.linear(x, sd[p + "linear_1.weight"], sd[p + "linear_1.bias"]))
    return F.linear(h, sd[p + "linear_2.weight"], sd[p + "linear_2.bias"])


def simp_head(sd, fuse: Tensor, seg_classes: Sequence[Tensor], cfg: NetCfg,
              prefix="field_type_classification_head."):
    label = torch.cat(list(seg_classes), 0).long()
    label_pn = (label > 0).long()
    fuse = fuse.reshape(-1, fuse.shape[-1])
    assert fuse.shape[0] == label.shape[0]
    w = None if cfg.loss_weights is None else torch.tensor(list(cfg.loss_weights), dtype=torch.float32)
    pn = _mlp(sd, prefix + "pos_neg_classification_net.", fuse)
    l_pn = ce_ohem(pn, label_pn, cfg.num_hard_positive_main_1, cfg.num_hard_negative_main_1, None, cfg.ohem_random)
    pc = _mlp(sd, prefix + "category_classification_net.", fuse)
    l_c = ce_ohem(pc, label, cfg.num_hard_positive_main_2, cfg.num_hard_negative_main_2, w, cfg.ohem_random)
    loss = l_pn + l_c if cfg.add_pos_neg else l_c
    return loss, label.int(), pc.detach().softmax(1), pc


def _binary_net(sd, p, x, layer_mode):
    """BinaryClassifier (model/field_type_classification_head.py:111-127): `layer` is a SingleLayer or a MultipleLayer -> [N,1]"""
    if layer_mode == "single":
        return F.linear(x, sd[p + "layer.linear.weight"], sd[p + "layer.linear.bias"])
    return _mlp(sd, p + "layer.", x)


def full_head(sd, fuse: Tensor, seg_classes: Sequence[Tensor], cfg: NetCfg, prefix="field_type_classification_head."):
    """FieldTypeClassification.forward (model/field_type_classification_head.py:334-407): a binary key / non-key net with
    BCELossRandomSample([num_hard_negative_1, num_hard_positive_1]), then one binary net per foreground class evaluated ONLY on
    the rows PREDICTED positive (sigmoid >= 0.5), each with its BCELossOHEM.  -> (loss, labels int32, class_pred [N,ncls])."""
    label = torch.cat(list(seg_classes), 0).long()
    fuse = fuse.reshape(-1, fuse.shape[-1])
    assert fuse.shape[0] == label.shape[0]
    pn = _binary_net(sd, prefix + "pos_neg_classification_net.", fuse, cfg.layer_mode).squeeze(1)
    l_pn = bce_random_sample(pn, (label > 0).float(), [cfg.num_hard_negative_main_1, cfg.num_hard_positive_main_1])
    m = pn.detach().sigmoid().ge(0.5)
    pos = fuse[m]
    pred = torch.zeros((fuse.shape[0], cfg.num_classes), dtype=pn.dtype)
    pred[:, 0] = pn.detach().sigmoid()
    l_c = torch.zeros((1,))
    if pos.shape[0] != 0:
        for ci in range(cfg.num_classes - 1):
            cp = _binary_net(sd, f"{prefix}category_classification_net_{ci}.", pos, cfg.layer_mode).squeeze(1)
            lab = (label[m] == (ci + 1)).float()
            l_c = l_c + bce_ohem(cp, lab, cfg.num_hard_positive_main_2, cfg.num_hard_negative_main_2, cfg.ohem_random)
            pred[:, ci + 1][m] = cp.detach().sigmoid()
    return l_pn + l_c, label.int(), pred


def crf_forward_alg(feats: Tensor, trans: Tensor, start: int, stop: int) -> Tensor:
    """log partition function of the linear-chain CRF (model/crf.py:48-79); trans[i, j] = score of j -> i"""
    T = trans.shape[0]
    fv = torch.full((T,), -10000.0)
    fv[start] = 0.0
    for feat in feats:
        fv = torch.logsumexp(fv.view(1, T) + trans + feat.view(T, 1), dim=1)
    return torch.logsumexp(fv + trans[stop], dim=0)


def crf_score(feats: Tensor, tags: Tensor, trans: Tensor, start: int, stop: int) -> Tensor:
    """score of the given tag sequence (model/crf.py:81-97)"""
    tg = torch.cat([torch.tensor([start], dtype=torch.long), tags.long()])
    s = torch.zeros(())
    for i in range(feats.shape[0]):
        s = s + trans[tg[i + 1], tg[i]] + feats[i, tg[i + 1]]
    return s + trans[stop, tg[-1]]


def crf_viterbi(feats: Tensor, trans: Tensor, start: int, stop: int):
    """(best path score, best tag sequence) (model/crf.py:99-145); first maximal index on ties like torch.max"""
    T = trans.shape[0]
    fv = torch.full((T,), -10000.0)
    fv[start] = 0.0
    back = []
    for feat in feats:
        cand = fv.view(1, T) + trans                      # [next, prev]
        best = cand.argmax(dim=1)
        back.append(best)
        fv = cand.gather(1, best.view(T, 1)).squeeze(1) + feat
    term = fv + trans[stop]
    cur = int(term.argmax())
    score = term[cur]
    path = [cur]
    for bp in reversed(back):
        cur = int(bp[cur])
        path.append(cur)
    assert path.pop() == start
    path.reverse()
    return score, path


def crf_head(sd, fuse: Tensor, seg_classes: Sequence[Tensor], cfg: NetCfg, train: bool, prefix="field_type_classification_head."):
    """CRFFieldTypeClassification.forward (model/field_type_classification_head.py:669-718): emissions over ncls + 2 tags
    (START = ncls, STOP = ncls + 1); training: mean over documents of (log Z - gold score) / len; eval: Viterbi tags."""
    label = torch.cat(list(seg_classes), 0)
    fuse = fuse.reshape(-1, fuse.shape[-1])
    p = prefix + "category_classification_net."
    em = F.linear(fuse, sd[p + "linear.weight"], sd[p + "linear.bias"]) if cfg.layer_mode == "single" else _mlp(sd, p, fuse)
    trans = sd[prefix + "crf_layer.transitions"]
    start, stop = cfg.num_classes, cfg.num_classes + 1
    score = torch.zeros((1,))
    o = 0
    tags_out = []
    for c in seg_classes:
        n = int(c.shape[0])
        f, t = em[o:o + n], label[o:o + n]
        o += n
        if train:
            score = score + (crf_forward_alg(f, trans, start, stop) - crf_score(f, t, trans, start, stop)) / n
        else:
            sc, path = crf_viterbi(f, trans, start, stop)
            score = score + sc
            tags_out.append(torch.tensor(path))
    score = score / len(seg_classes)
    if train:
        return score, label.int(), em.detach().float()
    return score, label.int(), torch.cat(tags_out, 0).unsqueeze(1).float()


# --------------------------------------------------------------------------------------
# a14. the whole forward  (model/ViBERTgrid_net.py:501-544)
# --------------------------------------------------------------------------------------
def forward(sd: Dict[str, Tensor], cfg: NetCfg, image, seg_indices, segment_classes, coors, corpus, mask,
            training: bool, min_sizes=None, return_intermediates: bool = False):
    """Returns (total_loss fp64[1], pred_mask, pred_ss, gt_label int32[N], pred_label fp32[N,ncls])
    plus, optionally, a dict of intermediates."""
    batch, icoors, _ = transform(image, coors, cfg, training, min_sizes)
    H, W = batch.shape[-2:]
    emb = bert_embedding(sd, "bert_model.", corpus, mask, cfg.bert, training)
    embs = [seg_aggregate(emb[b], mask[b], seg_indices[b], cfg.grid_mode) for b in range(emb.shape[0])]
    grid = grid_scatter(embs, icoors, H, W, cfg.stride)
    p_fuse = backbone_forward(sd, batch, grid, cfg.backbone, training)
    if cfg.classifier_mode == "simp":
        loss_aux, pred_mask, pred_ss = seg_head(sd, p_fuse, segment_classes, icoors, cfg, training)
    else:                                                 # full and crf share the two-stage seg head (:384-396, :443-456)
        loss_aux, pred_mask, pred_ss = seg_head_full(sd, p_fuse, segment_classes, icoors, cfg, training)
    roi = roi_align(p_fuse, [c.float() for c in icoors], cfg.roi_shape, 1.0 / float(cfg.p_fuse_stride))
    fuse = late_fusion(sd, roi, embs, training)
    logits = None
    if cfg.classifier_mode == "simp":
        loss_c, gt, pred, logits = simp_head(sd, fuse, segment_classes, cfg)
    elif cfg.classifier_mode == "full":
        loss_c, gt, pred = full_head(sd, fuse, segment_classes, cfg)
    else:
        loss_c, gt, pred = crf_head(sd, fuse, segment_classes, cfg, training)
    total = loss_c + cfg.loss_control_lambda * loss_aux
    out = (total, pred_mask, pred_ss, gt, pred)
    if return_intermediates:
        return out, dict(batch=batch, coors=icoors, emb=emb, embs=embs, grid=grid, p_fuse=p_fuse, roi=roi,
                         fuse=fuse, logits=logits, loss_aux=loss_aux, loss_c=loss_c)
    return out


# --------------------------------------------------------------------------------------
# a15. optimizer steps  (torch.optim.SGD / AdamW as configured at train_SROIE.py:223-235)
# --------------------------------------------------------------------------------------
def sgd_step(p: Tensor, g: Tensor, buf: Optional[Tensor], lr, momentum, wd):
    """torch.optim.SGD(momentum, weight_decay), dampening 0, no nesterov.  Returns new (p, buf)."""
    g = g + wd * p
    buf = g.clone() if buf is None else momentum * buf + g
    return p - lr * buf, buf


def adamw_step(p, g, m, v, step: int, lr, b1, b2, eps, wd):
    """torch.optim.AdamW (decoupled weight decay), amsgrad False.  `step` is 1-based."""
    p = p * (1 - lr * wd)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return p - (lr / bc1) * m / denom, m, v


# --------------------------------------------------------------------------------------
# deterministic weights shared by the golden generator, the oracle tests and the GPU tests
# --------------------------------------------------------------------------------------
def _key_seed(key: str) -> int:
    h = 1469598103934665603
    for ch in key.encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h & 0x7FFFFFFF


def synth_tensor(key: str, shape, kind: str) -> Tensor:
    """Deterministic parameter/buffer value derived from its state_dict key."""
    # BERT is registered twice in the reference (same storage): seed both names identically
    key = key.replace("BERTgrid_generator.model.", "bert_model.")
    g = torch.Generator().manual_seed(_key_seed(key))
    shape = tuple(shape)
    if kind == "ones_ish":          # BN / LN weights
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    if kind == "var":               # running_var
        return 0.5 + torch.rand(shape, generator=g)
    if kind == "small":             # biases, running_mean
        return 0.05 * torch.randn(shape, generator=g)
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    if kind == "embed":
        return 0.05 * torch.randn(shape, generator=g)
    return torch.randn(shape, generator=g) * (1.0 / math.sqrt(max(fan_in, 1)))


def synth_kind(key: str, shape) -> str:
    if key.endswith("num_batches_tracked") or key.endswith("position_ids") or key.endswith("token_type_ids"):
        return "skip"
    if "_loss" in key.rsplit(".", 2)[-2] and key.endswith(".weight"):       # class-weight buffers of the loss modules: set by the config
        return "skip"
    if key.endswith("running_var"):
        return "var"
    if key.endswith("running_mean"):
        return "small"
    if "embeddings.weight" in key or key.endswith("_embeddings.weight"):
        return "embed"
    if len(shape) == 1:
        # every 1-D `.weight` in this model is a BatchNorm / LayerNorm scale
        return "ones_ish" if key.endswith(".weight") else "small"
    return "weight"


def synth_state_dict(shapes: Dict[str, Tuple[int, ...]]) -> Dict[str, Tensor]:
    """shapes: key -> shape (from any module's state_dict).  Integer buffers are left out."""
    out = {}
    for k, shp in shapes.items():
        kind = synth_kind(k, shp)
        if kind == "skip":
            continue
        out[k] = synth_tensor(k, shp, kind)
    return out


# --------------------------------------------------------------------------------------
# state_dict inventory (SURVEY.md §8b "State"): key -> shape for a given configuration.
# --------------------------------------------------------------------------------------
def _bn_shapes(out, name, c):
    out[name + ".weight"] = (c,)
    out[name + ".bias"] = (c,)
    out[name + ".running_mean"] = (c,)
    out[name + ".running_var"] = (c,)
    out[name + ".num_batches_tracked"] = ()


def bert_shapes(prefix: str, bc: BertCfg, vocab: int, max_pos: int = 512, type_vocab: int = 2) -> Dict[str, tuple]:
    s = {}
    p = prefix
    h, im = bc.hidden, bc.intermediate
    s[p + "embeddings.word_embeddings.weight"] = (vocab, h)
    s[p + "embeddings.position_embeddings.weight"] = (max_pos, h)
    s[p + "embeddings.token_type_embeddings.weight"] = (type_vocab, h)
    s[p + "embeddings.LayerNorm.weight"] = (h,)
    s[p + "embeddings.LayerNorm.bias"] = (h,)
    for i in range(bc.layers):
        lp = f"{p}encoder.layer.{i}."
        for n in ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense"):
            s[lp + n + ".weight"] = (h, h)
            s[lp + n + ".bias"] = (h,)
        s[lp + "attention.output.LayerNorm.weight"] = (h,)
        s[lp + "attention.output.LayerNorm.bias"] = (h,)
        s[lp + "intermediate.dense.weight"] = (im, h)
        s[lp + "intermediate.dense.bias"] = (im,)
        s[lp + "output.dense.weight"] = (h, im)
        s[lp + "output.dense.bias"] = (h,)
        s[lp + "output.LayerNorm.weight"] = (h,)
        s[lp + "output.LayerNorm.bias"] = (h,)
    s[p + "pooler.dense.weight"] = (h, h)
    s[p + "pooler.dense.bias"] = (h,)
    return s


def backbone_shapes(kind: str, grid_channel: int = 768, prefix="backbone.") -> Dict[str, tuple]:
    sizes = [2, 2, 2, 2] if "18" in kind else [3, 4, 6, 3]
    chans = (64, 128, 256, 512)
    s = {}
    p = prefix
    if kind.endswith("_pretrained"):
        r = p + "resnet."
        s[r + "conv1.weight"] = (64, 3, 7, 7)
        _bn_shapes(s, r + "bn1", 64)
        cin = 64
        for li, (c, n) in enumerate(zip(chans, sizes), 1):
            for i in range(n):
                b = f"{r}layer{li}.{i}."
                s[b + "conv1.weight"] = (c, cin, 3, 3)
                _bn_shapes(s, b + "bn1", c)
                s[b + "conv2.weight"] = (c, c, 3, 3)
                _bn_shapes(s, b + "bn2", c)
                if i == 0 and li > 1:
                    s[b + "downsample.0.weight"] = (c, cin, 1, 1)
                    _bn_shapes(s, b + "downsample.1", c)
                cin = c
        s[r + "fc.weight"] = (1000, 512)
        s[r + "fc.bias"] = (1000,)
        s[p + "early_fusion.weight"] = (128, 128 + grid_channel, 1, 1)
    else:
        s[p + "conv_1.0.weight"] = (64, 3, 7, 7)
        _bn_shapes(s, p + "conv_1.1", 64)

        sc = (1, 2) if "_D_" in kind else (0, 1)          # DBlock's shortcut Sequential starts with a parameter-free AvgPool2d

        def block(b, cin, c, down):
            s[b + "conv_1.weight"] = (c, cin if down else c, 3, 3)
            if down:
                s[b + f"conv_shortcut.{sc[0]}.weight"] = (c, cin, 1, 1)
                _bn_shapes(s, b + f"conv_shortcut.{sc[1]}", c)
            _bn_shapes(s, b + "bn_1", c)
            s[b + "conv_2.weight"] = (c, c, 3, 3)
            _bn_shapes(s, b + "bn_2", c)

        for i in range(sizes[0]):
            block(f"{p}conv_2_x.{i}.", 64, 64, False)
        block(p + "conv_3_x.block_1.", 64, 128, True)
        s[p + "conv_3_x.early_fusion.weight"] = (128, 128 + grid_channel, 1, 1)
        s[p + "conv_3_x.early_fusion.bias"] = (128,)
        for i in range(sizes[1] - 1):
            block(f"{p}conv_3_x.layers.{i}.", 128, 128, False)
        for i in range(sizes[2]):
            block(f"{p}conv_4_x.{i}.", 128, 256, i == 0)
        for i in range(sizes[3]):
            block(f"{p}conv_5_x.{i}.", 256, 512, i == 0)
    s[p + "conv_6_x.weight"] = (256, 512, 1, 1)
    for j, c in ((1, 256), (2, 128), (3, 64)):
        s[p + f"skip_{j}.weight"] = (256, c, 1, 1)
        s[p + f"merge_{j}.weight"] = (256, 256, 3, 3)
    s[p + "fuse.weight"] = (256, 1024, 1, 1)
    return s


def head_shapes(ncls: int, hidden: int = 768, roi: int = 7, fuse: int = 1024, mode: str = "simp", layer_mode: str = "multi") -> Dict[str, tuple]:
    s = {}
    p = "late_fusion_net.ROI_embedding_net."
    s[p + "conv_1.weight"] = (256, 256, 3, 3)
    _bn_shapes(s, p + "bn_1", 256)
    s[p + "conv_2.weight"] = (256, 256, 3, 3)
    _bn_shapes(s, p + "bn_2", 256)
    s[p + "linear.weight"] = (1024, 256 * roi * roi)
    s[p + "linear.bias"] = (1024,)
    s["late_fusion_net.fuse_embedding_net.linear.weight"] = (1024, hidden + 1024)
    s["late_fusion_net.fuse_embedding_net.linear.bias"] = (1024,)
    def net_shapes(q, n, single):
        if single:
            s[q + "linear.weight"] = (n, fuse)
            s[q + "linear.bias"] = (n,)
        else:
            s[q + "linear_1.weight"] = (fuse // 2, fuse)
            s[q + "linear_1.bias"] = (fuse // 2,)
            s[q + "linear_2.weight"] = (n, fuse // 2)
            s[q + "linear_2.bias"] = (n,)

    h = "field_type_classification_head."
    if mode == "simp":                                    # always the 2-layer MLP (the "sigle" typo, :474)
        net_shapes(h + "pos_neg_classification_net.", 2, False)
        net_shapes(h + "category_classification_net.", ncls, False)
    elif mode == "full":
        net_shapes(h + "pos_neg_classification_net.layer.", 1, layer_mode == "single")
        for ci in range(ncls - 1):
            net_shapes(f"{h}category_classification_net_{ci}.layer.", 1, layer_mode == "single")
    else:
        net_shapes(h + "category_classification_net.", ncls + 2, layer_mode == "single")
        s[h + "crf_layer.transitions"] = (ncls + 2, ncls + 2)
    e = "semantic_segmentation_head.semantic_segmentation_encoder." if mode == "simp" else "semantic_segmentation_head.ss_encoder."
    if mode != "simp":
        for ci in range(ncls - 1):
            s[f"semantic_segmentation_head.ss_binary_classifier_{ci}.conv1.weight"] = (1, ncls, 1, 1)
            s[f"semantic_segmentation_head.ss_binary_classifier_{ci}.conv1.bias"] = (1,)
    s[e + "conv_1.weight"] = (256, 256, 3, 3)
    _bn_shapes(s, e + "bn_1", 256)
    s[e + "conv_2.weight"] = (256, 256, 3, 3)
    _bn_shapes(s, e + "bn_2", 256)
    s[e + "conv_3_1.weight"] = (3, 256, 1, 1)
    s[e + "conv_3_1.bias"] = (3,)
    s[e + "conv_3_2.weight"] = (ncls, 256, 1, 1)
    s[e + "conv_3_2.bias"] = (ncls,)
    return s


def state_shapes(cfg: NetCfg, vocab: int, max_pos: int = 512, type_vocab: int = 2, dup_bert: bool = True) -> Dict[str, tuple]:
    """Every key of `ViBERTgridNet.state_dict()` (simp classifier) -> shape.  BERT appears under
    `bert_model.` and again under `BERTgrid_generator.model.` (same storage in the reference)."""
    s = {}
    s.update(bert_shapes("bert_model.", cfg.bert, vocab, max_pos, type_vocab))
    s.update(backbone_shapes(cfg.backbone, cfg.bert.hidden))
    if dup_bert:
        s.update(bert_shapes("BERTgrid_generator.model.", cfg.bert, vocab, max_pos, type_vocab))
    s.update(head_shapes(cfg.num_classes, cfg.bert.hidden, cfg.roi_shape, mode=cfg.classifier_mode, layer_mode=cfg.layer_mode))
    if cfg.loss_weights is not None and cfg.classifier_mode == "simp":
        # nn.CrossEntropyLoss keeps its class weights as a BUFFER, so the reference's state_dict carries them ([probe], simp mode)
        s["field_type_classification_head.field_type_classification_loss.weight"] = (len(cfg.loss_weights),)
        s["semantic_segmentation_head.aux_loss_2.weight"] = (len(cfg.loss_weights),)
    return s


def synth_model_state(cfg: NetCfg, vocab: int, max_pos: int = 512, type_vocab: int = 2) -> Dict[str, Tensor]:
    """The deterministic weights the golden fixtures were generated with."""
    return synth_state_dict(state_shapes(cfg, vocab, max_pos, type_vocab))
