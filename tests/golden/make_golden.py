"""Generate the golden fixtures under tests/golden/ by importing the REAL reference.

Runs only in the build container (needs /root/reference); the fixtures it writes are plain
data (inputs + the reference's outputs) and are committed.  Nothing here travels as code the
tests execute: tests read the .npz files only.

    python tests/golden/make_golden.py            # regenerate everything

Shims: torchvision is not installed -> stub modules whose `ops.RoIAlign` is the
oracle's published-algorithm restatement (parity for RoIAlign is therefore pinned by KATs, not by
this script) and whose `models.resnet18/34` are structurally identical random-init ResNets; HF
weights cannot be downloaded -> a local `bert-base-uncased/` directory with a small BertConfig
(hidden 768, 2 layers) and a dummy vocab.  Weights everywhere are the deterministic
`oracle.vbg_oracle.synth_state_dict` values derived from the state_dict key names, so the
tests can rebuild them without storing 100 MB of parameters.
"""
import json
import os
import random
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import vbg_oracle as O  # noqa: E402

import transformers  # noqa: E402  (import BEFORE stubbing torchvision: its availability probe must see it absent)
from transformers import BertConfig, BertModel, BertTokenizer  # noqa: E402

REF = "/root/reference"


# ---------------------------------------------------------------------------------------------
# shims
# ---------------------------------------------------------------------------------------------
def install_torchvision_stub():
    import importlib.machinery
    import torch.nn as nn

    def mod(name):
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        sys.modules[name] = m
        return m

    tv = mod("torchvision")
    ops = mod("torchvision.ops")
    models = mod("torchvision.models")
    tfm = mod("torchvision.transforms")
    tv.ops, tv.models, tv.transforms = ops, models, tfm

    class RoIAlign(nn.Module):
        def __init__(self, output_size, spatial_scale, sampling_ratio, aligned=False):
            super().__init__()
            assert sampling_ratio <= 0 and not aligned
            self.output_size, self.spatial_scale = output_size, spatial_scale

        def forward(self, x, boxes):
            return O.roi_align(x, boxes, self.output_size, self.spatial_scale)

    ops.RoIAlign = RoIAlign

    class _Block(nn.Module):
        def __init__(self, cin, cout, stride):
            super().__init__()
            self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
            self.bn1 = nn.BatchNorm2d(cout)
            self.relu = nn.ReLU(inplace=True)
            self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
            self.bn2 = nn.BatchNorm2d(cout)
            self.downsample = None
            if stride != 1 or cin != cout:
                self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

        def forward(self, x):
            idt = x if self.downsample is None else self.downsample(x)
            out = self.relu(self.bn1(self.conv1(x)))
            out = self.bn2(self.conv2(out))
            return self.relu(out + idt)

    class _ResNet(nn.Module):
        def __init__(self, sizes):
            super().__init__()
            self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
            self.bn1 = nn.BatchNorm2d(64)
            self.relu = nn.ReLU(inplace=True)
            self.maxpool = nn.MaxPool2d(3, 2, 1)
            cin = 64
            for li, (c, n) in enumerate(zip((64, 128, 256, 512), sizes), 1):
                blocks = []
                for i in range(n):
                    blocks.append(_Block(cin, c, 2 if (i == 0 and li > 1) else 1))
                    cin = c
                setattr(self, f"layer{li}", nn.Sequential(*blocks))
            self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
            self.fc = nn.Linear(512, 1000)

    models.resnet18 = lambda pretrained=False, **k: _ResNet([2, 2, 2, 2])
    models.resnet34 = lambda pretrained=False, **k: _ResNet([3, 4, 6, 3])


def make_bert_dir(top, name, layers, vocab):
    d = os.path.join(top, name)
    os.makedirs(d, exist_ok=True)
    cfg = BertConfig(vocab_size=vocab, num_hidden_layers=layers, hidden_dropout_prob=0.0,
                     attention_probs_dropout_prob=0.0)
    cfg.save_pretrained(d)
    toks = ["[PAD]"] + [f"[unused{i}]" for i in range(1, 100)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    toks += [f"tok{i}" for i in range(len(toks), vocab)]
    with open(os.path.join(d, "vocab.txt"), "w") as f:
        f.write("\n".join(toks) + "\n")
    return d


def shapes_of(module):
    return {k: tuple(v.shape) for k, v in module.state_dict().items()}


def load_synth(module):
    sd = O.synth_state_dict(shapes_of(module))
    missing, unexpected = module.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    for k in missing:
        assert k.endswith("num_batches_tracked") or k.endswith("position_ids") or k.endswith("token_type_ids"), k
    return sd


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name, {k: v.shape for k, v in out.items()})


# ---------------------------------------------------------------------------------------------
def gen_transform(T):
    tr = T.GeneralizedViBERTgridTransform([0.9248, 0.9224, 0.9215], [0.1532, 0.1545, 0.1536], [48, 64], 56, 80)
    g = torch.Generator().manual_seed(1)
    imgs = (torch.rand(3, 61, 47, generator=g), torch.rand(3, 50, 90, generator=g), torch.rand(3, 64, 64, generator=g))
    coors = (torch.tensor([[3, 5, 40, 20], [0, 0, 47, 61], [10, 30, 11, 31], [46, 60, 47, 61]]),
             torch.tensor([[1, 2, 88, 49], [30, 10, 60, 40]]),
             torch.tensor([[0, 0, 64, 64], [7, 9, 23, 33], [63, 1, 64, 2]]))
    out = {}
    for i in range(3):
        out[f"img{i}"] = imgs[i]
        out[f"coor{i}"] = coors[i]
    # eval mode: test_min_size 56
    tr.eval()
    il, oc = tr(imgs, coors)
    out["eval_batch"] = il.tensors
    out["eval_sizes"] = np.array(il.image_sizes)
    for i in range(3):
        out[f"eval_coor{i}"] = oc[i]
    # train mode with a pinned torch RNG; record which sizes got drawn via the output sizes
    tr.train()
    torch.manual_seed(123)
    il, oc = tr(imgs, coors)
    out["train_batch"] = il.tensors
    out["train_sizes"] = np.array(il.image_sizes)
    for i in range(3):
        out[f"train_coor{i}"] = oc[i]
    # identity case
    tr2 = T.GeneralizedViBERTgridTransform([0.9248, 0.9224, 0.9215], [0.1532, 0.1545, 0.1536], [64], 64, 64)
    tr2.eval()
    il, oc = tr2((imgs[2],), (coors[2],))
    out["ident_batch"] = il.tensors
    out["ident_coor"] = oc[0]
    npz("transform.npz", **out)


def gen_windows(G):
    class Fake(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.calls = []

        def forward(self, input_ids=None, attention_mask=None):
            self.calls.append((input_ids.clone(), attention_mask.clone()))
            h = torch.stack([input_ids.float(), attention_mask.float(),
                             torch.arange(input_ids.shape[1]).float()[None].expand_as(input_ids)], -1)
            return types.SimpleNamespace(last_hidden_state=h)

    out = {}
    for T_ in (5, 509, 510, 511, 512, 1020, 1021):
        fake = Fake()
        gen = G.BERTgridGenerator(bert_model=fake, grid_mode="mean", stride=8)
        g = torch.Generator().manual_seed(T_)
        lens = [T_, max(1, T_ - 3)]
        corpus = torch.zeros((2, T_), dtype=torch.long)
        for b, L in enumerate(lens):
            corpus[b, :L] = torch.randint(1000, 30000, (L,), generator=g)
        mask = (corpus != 0).int()
        segs = tuple(torch.arange(L, dtype=torch.int32) // 2 for L in lens)
        embs = gen.BERT_embedding(corpus, mask, segs)
        out[f"T{T_}_corpus"] = corpus
        out[f"T{T_}_nwin"] = len(fake.calls)
        for w, (ids, am) in enumerate(fake.calls):
            out[f"T{T_}_ids{w}"] = ids
            out[f"T{T_}_am{w}"] = am
        for b in range(2):
            out[f"T{T_}_emb{b}"] = embs[b]
    npz("windows.npz", **out)


def gen_aggregate(G):
    out = {}
    g = torch.Generator().manual_seed(5)
    for mode in ("mean", "first"):
        class Fake(torch.nn.Module):
            def forward(self, input_ids=None, attention_mask=None):
                gg = torch.Generator().manual_seed(int(input_ids.sum()) % 100000)
                return types.SimpleNamespace(last_hidden_state=torch.randn(*input_ids.shape, 16, generator=gg))
        gen = G.BERTgridGenerator(bert_model=Fake(), grid_mode=mode, stride=8)
        corpus = torch.zeros((2, 11), dtype=torch.long)
        corpus[0, :11] = torch.randint(1000, 2000, (11,), generator=g)
        corpus[1, :7] = torch.randint(1000, 2000, (7,), generator=g)
        mask = (corpus != 0).int()
        segs = (torch.tensor([0, 0, 0, 1, 2, 2, 5, 5, 5, 5, 6], dtype=torch.int32),
                torch.tensor([3, 3, 4, 4, 4, 4, 9], dtype=torch.int32))
        # capture token embeddings through the same fake
        wins = O.bert_windows(corpus, mask)
        tok = torch.cat([Fake()(ids, am).last_hidden_state[:, 1:1 + cur] for ids, am, cur in wins], 1)
        embs = gen.BERT_embedding(corpus, mask, segs)
        out[f"{mode}_tok"] = tok
        out[f"{mode}_mask"] = mask
        for b in range(2):
            out[f"{mode}_seg{b}"] = segs[b]
            out[f"{mode}_out{b}"] = embs[b]
    npz("aggregate.npz", **out)


def gen_scatter(G):
    gen = G.BERTgridGenerator(bert_model=torch.nn.Identity(), grid_mode="mean", stride=8)
    g = torch.Generator().manual_seed(9)
    H, W = 64, 96
    boxes = (torch.tensor([[0, 0, 40, 24], [16, 8, 64, 40], [17, 9, 23, 15], [90, 50, 200, 100], [30, 30, 10, 10],
                           [-20, -9, 20, 9], [8, 56, 96, 64], [40, 16, 48, 24]], dtype=torch.int32),
             torch.tensor([[0, 0, 96, 64], [8, 8, 88, 56], [16, 16, 80, 48]], dtype=torch.int32),
             torch.zeros((0, 4), dtype=torch.int32))
    embs = tuple(torch.randn(b.shape[0], 6, generator=g, requires_grad=True) for b in boxes)
    grid = gen.BERTgrid_embedding((H, W), embs, boxes)
    gout = torch.randn(grid.shape, generator=g)
    grid.backward(gout)
    out = dict(H=H, W=W, grid=grid, gout=gout)
    for b in range(3):
        out[f"box{b}"] = boxes[b]
        out[f"emb{b}"] = embs[b]
        out[f"gemb{b}"] = embs[b].grad if embs[b].grad is not None else torch.zeros_like(embs[b])
    npz("scatter.npz", **out)


def gen_losses(L):
    out = {}
    g = torch.Generator().manual_seed(11)
    # random-sample CE on [B,3,H,W]
    x = torch.randn(2, 3, 24, 24, generator=g, requires_grad=True)
    t = torch.randint(0, 3, (2, 24, 24), generator=g)
    t[0, :12] = 0
    random.seed(77)
    l = L.CrossEntropyLossRandomSample(sample_list=[256, 512, 256])(x, t)
    l.backward()
    out.update(rs_x=x, rs_t=t, rs_loss=l, rs_grad=x.grad)
    # OHEM CE without random, 5 classes, no ties
    x2 = torch.randn(400, 5, generator=g, requires_grad=True)
    t2 = (torch.rand(400, generator=g) > 0.7).long() * torch.randint(1, 5, (400,), generator=g)
    l2 = L.CrossEntropyLossOHEM(num_hard_positive=32, num_hard_negative=32)(x2, t2)
    l2.backward()
    out.update(oh_x=x2, oh_t=t2, oh_loss=l2, oh_grad=x2.grad)
    # OHEM with random pre-sampling and class weights
    x3 = torch.randn(300, 5, generator=g, requires_grad=True)
    t3 = (torch.rand(300, generator=g) > 0.5).long() * torch.randint(1, 5, (300,), generator=g)
    w = torch.tensor([0.5, 1.0, 2.0, 1.5, 1.0])
    random.seed(5)
    l3 = L.CrossEntropyLossOHEM(num_hard_positive=16, num_hard_negative=16, weight=w, random=True)(x3, t3)
    l3.backward()
    out.update(ohr_x=x3, ohr_t=t3, ohr_w=w, ohr_loss=l3, ohr_grad=x3.grad)
    # OHEM with ties (replicated logits like the x4 nearest-upsampled seg head)
    base = torch.randn(1, 5, 6, 6, generator=g)
    x4 = torch.nn.functional.interpolate(base, scale_factor=4, mode="nearest").clone().requires_grad_(True)
    t4 = torch.zeros(1, 24, 24, dtype=torch.long)
    t4[0, 3:14, 2:17] = 2
    t4[0, 10:20, 12:22] = 4
    l4 = L.CrossEntropyLossOHEM(num_hard_positive=40, num_hard_negative=40)(x4, t4)
    l4.backward()
    out.update(tie_x=x4, tie_t=t4, tie_loss=l4, tie_grad=x4.grad)
    # fewer elements than k
    x5 = torch.randn(10, 2, generator=g, requires_grad=True)
    t5 = torch.tensor([0, 1, 0, 0, 1, 0, 0, 0, 0, 0])
    random.seed(1)
    l5 = L.CrossEntropyLossOHEM(num_hard_positive=16, num_hard_negative=16, random=True)(x5, t5)
    l5.backward()
    out.update(few_x=x5, few_t=t5, few_loss=l5, few_grad=x5.grad)
    npz("losses.npz", **out)


def gen_losses_bce(L):
    """BCELossRandomSample / BCELossOHEM (pipeline/custom_loss.py:204-382), the losses of classifier_mode full"""
    out = {}
    g = torch.Generator().manual_seed(13)
    # random sample: categories split by the sign of the PREDICTION; both categories larger than k
    x = torch.randn(300, generator=g, requires_grad=True)
    t = (torch.rand(300, generator=g) > 0.6).float()
    random.seed(3)
    l = L.BCELossRandomSample(sample_list=[32, 48])(x, t)
    l.backward()
    out.update(rs_x=x, rs_t=t, rs_loss=l, rs_grad=x.grad)
    # random sample with one category smaller than its k (kept whole), [N,1] input
    x1 = (torch.randn(40, 1, generator=g) - 1.5).requires_grad_(True)
    t1 = (torch.rand(40, generator=g) > 0.5).float()
    random.seed(4)
    l1 = L.BCELossRandomSample(sample_list=[16, 16])(x1, t1)
    l1.backward()
    out.update(rs2_x=x1, rs2_t=t1, rs2_loss=l1, rs2_grad=x1.grad)
    # OHEM without random
    x2 = torch.randn(400, generator=g, requires_grad=True)
    t2 = (torch.rand(400, generator=g) > 0.7).float()
    l2 = L.BCELossOHEM(num_hard_positive=32, num_hard_negative=32)(x2, t2)
    l2.backward()
    out.update(oh_x=x2, oh_t=t2, oh_loss=l2, oh_grad=x2.grad)
    # OHEM with random pre-sampling
    x3 = torch.randn(300, generator=g, requires_grad=True)
    t3 = (torch.rand(300, generator=g) > 0.5).float()
    random.seed(5)
    l3 = L.BCELossOHEM(num_hard_positive=16, num_hard_negative=16, random=True)(x3, t3)
    l3.backward()
    out.update(ohr_x=x3, ohr_t=t3, ohr_loss=l3, ohr_grad=x3.grad)
    # fewer elements than k, and no positives at all
    x4 = torch.randn(9, generator=g, requires_grad=True)
    t4 = torch.zeros(9)
    l4 = L.BCELossOHEM(num_hard_positive=16, num_hard_negative=4)(x4, t4)
    l4.backward()
    out.update(few_x=x4, few_t=t4, few_loss=l4, few_grad=x4.grad)
    npz("losses_bce.npz", **out)


def gen_e2e_modes(V, tmp):
    """classifier_mode full (layer_mode multi) and crf (layer_mode single) of the whole model on the e2e.npz documents:
    two-stage seg head + two-stage classifier / CRF head (model/field_type_classification_head.py:193-407, 591-718, model/crf.py)"""
    tokenizer = BertTokenizer(os.path.join(tmp, "bert-base-uncased", "vocab.txt"))
    e = np.load(os.path.join(HERE, "e2e.npz"))
    B = 2
    imgs = tuple(torch.from_numpy(e[f"img{b}"]) for b in range(B))
    coors = tuple(torch.from_numpy(e[f"coor{b}"]) for b in range(B))
    segs = tuple(torch.from_numpy(e[f"seg{b}"]) for b in range(B))
    classes = tuple(torch.from_numpy(e[f"class{b}"]) for b in range(B))
    corpus, mask = torch.from_numpy(e["corpus"]), torch.from_numpy(e["mask"])
    out = {}
    for mode, layer_mode in (("full", "multi"), ("crf", "single")):
        net = V.ViBERTgridNet(num_classes=5, image_mean=[0.9248, 0.9224, 0.9215], image_std=[0.1532, 0.1545, 0.1536],
                              image_min_size=[96], image_max_size=128, test_image_min_size=96,
                              bert_model="bert-base-uncased", tokenizer=tokenizer, backbone="resnet_18_fpn", grid_mode="mean",
                              loss_weights=None, num_hard_positive_main_1=4, num_hard_negative_main_1=4,
                              num_hard_positive_main_2=3, num_hard_negative_main_2=3,
                              loss_aux_sample_list=[64, 128, 64], num_hard_positive_aux=64, num_hard_negative_aux=64,
                              loss_control_lambda=1, add_pos_neg=True, classifier_mode=mode, ohem_random=True,
                              layer_mode=layer_mode, work_mode="eval",
                              tag_to_idx={f"c{i}": i for i in range(5)} if mode == "crf" else None)
        load_synth(net)
        if mode == "crf":          # the constructor's constraints on the transitions, re-applied over the synthetic values (model/crf.py:42-45)
            tr = net.field_type_classification_head.crf_layer.transitions
            tr.data = tr.data * 3.0
            tr.data[5, :] = -10000
            tr.data[:, 6] = -10000
            out["crf_transitions"] = tr.data.clone()
        net.eval()
        random.seed(7)
        with torch.no_grad():
            loss, pm, ps, gt, pred = net(imgs, segs, classes, coors, corpus, mask)
        out.update({f"{mode}_eval_loss": loss, f"{mode}_gt": gt, f"{mode}_pred": pred, f"{mode}_pred_ss": ps[:, :, ::8, ::8]})
        net.train()
        random.seed(7)
        loss = net(imgs, segs, classes, coors, corpus, mask)
        loss.backward()
        out[f"{mode}_train_loss"] = loss
        gn = {k: (0.0 if p.grad is None else float(p.grad.double().norm())) for k, p in net.named_parameters()
              if not k.startswith("BERTgrid_generator.")}
        out[f"{mode}_gradnorm_keys"] = np.array(sorted(gn.keys()))
        out[f"{mode}_gradnorm_vals"] = np.array([gn[k] for k in sorted(gn.keys())])
        out[f"{mode}_keys"] = np.array(list(shapes_of(net).keys()))
        out[f"{mode}_key_shapes"] = np.array([str(v) for v in shapes_of(net).values()])
    npz("e2e_modes.npz", **out)


def gen_labels(S):
    head = S.SimplifiedSemanticSegmentationClassifier(p_fuse_channel=8, num_classes=5,
                                                      loss_1_sample_list=[4, 4, 4], num_hard_positive=4,
                                                      num_hard_negative=4)
    rec = {}

    class Grab(torch.nn.Module):
        def __init__(self, name):
            super().__init__()
            self.name = name

        def forward(self, x, t):
            rec[self.name] = t.clone()
            return torch.zeros(())

    head.aux_loss_1 = Grab("pos_neg")
    head.aux_loss_2 = Grab("cls")
    coors = (torch.tensor([[0, 0, 20, 10], [5, 5, 30, 25], [6, 6, 8, 8], [60, 30, 90, 40], [10, 20, 10, 40]], dtype=torch.int32),
             torch.tensor([[2, 2, 62, 30], [30, 0, 64, 32]], dtype=torch.int32))
    classes = (torch.tensor([1, 0, 3, 2, 4], dtype=torch.int32), torch.tensor([0, 4], dtype=torch.int32))
    head.eval()
    head(torch.zeros(2, 8, 8, 16), classes, coors)
    out = dict(pos_neg=rec["pos_neg"], cls=rec["cls"])
    for b in range(2):
        out[f"coor{b}"] = coors[b]
        out[f"class{b}"] = classes[b]
    npz("labels.npz", **out)


def gen_bert(tmp):
    cfg = BertConfig(vocab_size=1200, num_hidden_layers=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    m = BertModel(cfg)
    load_synth(m)
    m.eval()
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(104, 1200, (2, 24), generator=g)
    am = torch.ones(2, 24, dtype=torch.long)
    ids[1, 15:] = 0
    am[1, 15:] = 0
    ids[1, 20] = 102
    am[1, 20] = 1
    with torch.no_grad():
        h = m(input_ids=ids, attention_mask=am).last_hidden_state
    npz("bert.npz", ids=ids, am=am, hidden=h, layers=2, vocab=1200)


def gen_backbone(R):
    out = {}
    g = torch.Generator().manual_seed(21)
    x = torch.randn(2, 3, 64, 96, generator=g)
    grid = torch.randn(2, 768, 8, 12, generator=g) * (torch.rand(2, 1, 8, 12, generator=g) > 0.5)
    out.update(x=x, grid=grid)
    for kind, fn, kw in (("resnet_18_fpn", R.resnet_18_fpn, {}), ("resnet_34_fpn_pretrained", R.resnet_34_fpn, dict(pretrained=True))):
        net = fn(grid_channel=768, **kw)
        load_synth(net)
        net.eval()
        with torch.no_grad():
            out[kind + "_eval"] = net(x, grid)
        net.train()
        with torch.no_grad():
            out[kind + "_train"] = net(x, grid)
        # running stats after one train step of a representative BN
        key = "conv_1.1.running_mean" if kind == "resnet_18_fpn" else "resnet.bn1.running_mean"
        out[kind + "_rm"] = net.state_dict()[key]
        out[kind + "_rv"] = net.state_dict()[key.replace("mean", "var")]
    npz("backbone.npz", **out)


def gen_backbone_d(R):
    """ResNet-D variant (DBlock: avg-pool shortcut, model/ResNetFPN_ViBERTgrid.py:187-269, factory :692-711)"""
    out = {}
    g = torch.Generator().manual_seed(21)
    x = torch.randn(2, 3, 64, 96, generator=g)
    grid = torch.randn(2, 768, 8, 12, generator=g) * (torch.rand(2, 1, 8, 12, generator=g) > 0.5)      # = backbone.npz's x / grid
    kind = "resnet_18_D_fpn"
    net = R.resnet_18_D_fpn(grid_channel=768)
    load_synth(net)
    out[kind + "_keys"] = np.array(sorted(net.state_dict().keys()))
    net.eval()
    with torch.no_grad():
        out[kind + "_eval"] = net(x, grid)
    net.train()
    with torch.no_grad():
        out[kind + "_train"] = net(x, grid)
    out[kind + "_rm"] = net.state_dict()["conv_4_x.0.conv_shortcut.2.running_mean"]
    out[kind + "_rv"] = net.state_dict()["conv_4_x.0.conv_shortcut.2.running_var"]
    npz("backbone_d.npz", **out)


def make_roberta_dir(top, name, layers, vocab):
    """local stand-in for `roberta-base` (RobertaConfig: 514 positions, 1 token type, pad id 1, LN eps 1e-5)"""
    import json
    from transformers import RobertaConfig
    d = os.path.join(top, name)
    os.makedirs(d, exist_ok=True)
    RobertaConfig(vocab_size=vocab, max_position_embeddings=514, type_vocab_size=1, num_hidden_layers=layers, hidden_dropout_prob=0.0,
                  attention_probs_dropout_prob=0.0, layer_norm_eps=1e-5).save_pretrained(d)
    voc = {"<s>": 0, "<pad>": 1, "</s>": 2, "<unk>": 3}
    for i in range(4, vocab):
        voc[f"t{i}"] = i
    json.dump(voc, open(os.path.join(d, "vocab.json"), "w"))
    open(os.path.join(d, "merges.txt"), "w").write("#version: 0.2\n")
    return d


def gen_e2e_roberta(V, tmp):
    """cfg3 / cfg5 flavour of the whole model: RobertaModel (RoBERTa position ids, one token type) + 4 classes, resnet_18_fpn;
    same documents as e2e.npz but with a PAD hole inside document 1 and class labels < 4."""
    from transformers import RobertaTokenizer
    d = make_roberta_dir(tmp, "roberta-base", layers=2, vocab=1300)
    tokenizer = RobertaTokenizer(os.path.join(d, "vocab.json"), os.path.join(d, "merges.txt"))
    net = V.ViBERTgridNet(num_classes=4, image_mean=[0.9248, 0.9224, 0.9215], image_std=[0.1532, 0.1545, 0.1536],
                          image_min_size=[96], image_max_size=128, test_image_min_size=96,
                          bert_model="roberta-base", tokenizer=tokenizer, backbone="resnet_18_fpn", grid_mode="mean",
                          loss_weights=None, num_hard_positive_main_1=4, num_hard_negative_main_1=4,
                          num_hard_positive_main_2=6, num_hard_negative_main_2=6,
                          loss_aux_sample_list=[64, 128, 64], num_hard_positive_aux=64, num_hard_negative_aux=64,
                          loss_control_lambda=1, add_pos_neg=True, classifier_mode="simp", ohem_random=True,
                          layer_mode="single", work_mode="eval")
    load_synth(net)
    e = np.load(os.path.join(HERE, "e2e.npz"))
    B = 2
    imgs = tuple(torch.from_numpy(e[f"img{b}"]) for b in range(B))
    coors = tuple(torch.from_numpy(e[f"coor{b}"]) for b in range(B))
    segs = tuple(torch.from_numpy(e[f"seg{b}"]) for b in range(B))
    classes = tuple(torch.from_numpy(e[f"class{b}"]) % 4 for b in range(B))
    corpus, mask = torch.from_numpy(e["corpus"]), torch.from_numpy(e["mask"])
    out = {"classes0": classes[0], "classes1": classes[1]}
    net.eval()
    random.seed(7)
    with torch.no_grad():
        loss, pm, ps, gt, pred = net(imgs, segs, classes, coors, corpus, mask)
    out.update(eval_loss=loss, gt=gt, pred=pred, pred_ss=ps[:, :, ::8, ::8])
    net.train()
    random.seed(7)
    loss = net(imgs, segs, classes, coors, corpus, mask)
    loss.backward()
    out["train_loss"] = loss
    gn = {k: (0.0 if p.grad is None else float(p.grad.double().norm())) for k, p in net.named_parameters()
          if not k.startswith("BERTgrid_generator.")}
    out["gradnorm_keys"] = np.array(sorted(gn.keys()))
    out["gradnorm_vals"] = np.array([gn[k] for k in sorted(gn.keys())])
    out["keys"] = np.array(list(shapes_of(net).keys()))
    out["key_shapes"] = np.array([str(v) for v in shapes_of(net).values()])
    npz("e2e_roberta.npz", **out)


def gen_e2e(V, tmp):
    out = {}
    tokenizer = BertTokenizer(os.path.join(tmp, "bert-base-uncased", "vocab.txt"))
    for tag, backbone in (("r18", "resnet_18_fpn"), ("r34p", "resnet_34_fpn_pretrained")):
        net = V.ViBERTgridNet(num_classes=5, image_mean=[0.9248, 0.9224, 0.9215], image_std=[0.1532, 0.1545, 0.1536],
                              image_min_size=[96], image_max_size=128, test_image_min_size=96,
                              bert_model="bert-base-uncased", tokenizer=tokenizer, backbone=backbone, grid_mode="mean",
                              loss_weights=None, num_hard_positive_main_1=4, num_hard_negative_main_1=4,
                              num_hard_positive_main_2=6, num_hard_negative_main_2=6,
                              loss_aux_sample_list=[64, 128, 64], num_hard_positive_aux=64, num_hard_negative_aux=64,
                              loss_control_lambda=1, add_pos_neg=True, classifier_mode="simp", ohem_random=True,
                              layer_mode="single", work_mode="eval")
        sd = load_synth(net)
        B, T_, S = 2, 24, 8
        g = torch.Generator().manual_seed(1234)
        imgs = tuple(torch.rand(3, 96, 128, generator=g) for _ in range(B))
        coors = []
        for b in range(B):
            x1 = torch.randint(0, 128 - 41, (S,), generator=g)
            y1 = torch.randint(0, 96 - 25, (S,), generator=g)
            w = torch.randint(8, 41, (S,), generator=g)
            h = torch.randint(8, 25, (S,), generator=g)
            coors.append(torch.stack([x1, y1, x1 + w, y1 + h], 1).long())
        coors = tuple(coors)
        segs = tuple(torch.arange(S, dtype=torch.int32).repeat_interleave(T_ // S) for _ in range(B))
        classes = tuple(torch.randint(0, 5, (S,), generator=g).int() for _ in range(B))
        corpus = torch.randint(1000, 1200, (B, T_), generator=g)
        mask = torch.ones(B, T_, dtype=torch.int32)
        # second doc shorter: 21 tokens, 7 segments
        corpus[1, 21:] = 0
        mask[1, 21:] = 0
        segs = (segs[0], segs[1][:21])
        coors = (coors[0], coors[1][:7])
        classes = (classes[0], classes[1][:7])

        net.eval()
        random.seed(7)
        with torch.no_grad():
            loss, pm, ps, gt, pred = net(imgs, segs, classes, coors, corpus, mask)
        out.update({f"{tag}_eval_loss": loss, f"{tag}_pred_mask": pm[:, :, ::4, ::4], f"{tag}_pred_ss": ps[:, :, ::4, ::4],
                    f"{tag}_gt": gt, f"{tag}_pred": pred})
        net.train()      # flips work_mode to "train" (model/ViBERTgrid_net.py:462-464)
        random.seed(7)
        loss = net(imgs, segs, classes, coors, corpus, mask)
        loss.backward()
        out[f"{tag}_train_loss"] = loss
        gn = {}
        for k, p in net.named_parameters():
            if k.startswith("BERTgrid_generator."):
                continue
            gn[k] = 0.0 if p.grad is None else float(p.grad.double().norm())
        out[f"{tag}_gradnorm_keys"] = np.array(sorted(gn.keys()))
        out[f"{tag}_gradnorm_vals"] = np.array([gn[k] for k in sorted(gn.keys())])
        pick = ["bert_model.encoder.layer.0.attention.self.query.weight", "late_fusion_net.fuse_embedding_net.linear.weight",
                "field_type_classification_head.category_classification_net.linear_2.weight",
                "bert_model.embeddings.word_embeddings.weight"]
        named = dict(net.named_parameters())
        for k in pick:
            out[f"{tag}_grad::{k}"] = named[k].grad.flatten()[:: max(1, named[k].numel() // 4096)][:4096]
        if tag == "r18":
            for b in range(B):
                out[f"img{b}"] = imgs[b]
                out[f"coor{b}"] = coors[b]
                out[f"seg{b}"] = segs[b]
                out[f"class{b}"] = classes[b]
            out["corpus"] = corpus
            out["mask"] = mask
            shapes = shapes_of(net)
            out["r18_keys"] = np.array(list(shapes.keys()))
        else:
            out["r34p_keys"] = np.array(list(shapes_of(net).keys()))
    npz("e2e.npz", **out)


def gen_e2e_amp(V, tmp):
    """`amp: True` (example_config.yaml:8): the reference's autocast region (pipeline/train_val_utils.py:264) around the r18 model
    of e2e.npz on the same documents.  CUDA autocast (fp16) cannot run here; CPU autocast to bfloat16 is the closest thing the
    reference itself can produce in this container -- same op list (conv / linear / matmul in reduced precision, normalisation
    and losses in fp32), same 8-bit-mantissa-class rounding as the product's bf16 matrix cores."""
    e = np.load(os.path.join(HERE, "e2e.npz"))
    tokenizer = BertTokenizer(os.path.join(tmp, "bert-base-uncased", "vocab.txt"))
    net = V.ViBERTgridNet(num_classes=5, image_mean=[0.9248, 0.9224, 0.9215], image_std=[0.1532, 0.1545, 0.1536],
                          image_min_size=[96], image_max_size=128, test_image_min_size=96,
                          bert_model="bert-base-uncased", tokenizer=tokenizer, backbone="resnet_18_fpn", grid_mode="mean",
                          loss_weights=None, num_hard_positive_main_1=4, num_hard_negative_main_1=4,
                          num_hard_positive_main_2=6, num_hard_negative_main_2=6,
                          loss_aux_sample_list=[64, 128, 64], num_hard_positive_aux=64, num_hard_negative_aux=64,
                          loss_control_lambda=1, add_pos_neg=True, classifier_mode="simp", ohem_random=True,
                          layer_mode="single", work_mode="eval")
    load_synth(net)
    imgs = tuple(torch.from_numpy(e[f"img{b}"]) for b in range(2))
    coors = tuple(torch.from_numpy(e[f"coor{b}"]) for b in range(2))
    segs = tuple(torch.from_numpy(e[f"seg{b}"]) for b in range(2))
    classes = tuple(torch.from_numpy(e[f"class{b}"]) for b in range(2))
    corpus, mask = torch.from_numpy(e["corpus"]), torch.from_numpy(e["mask"])
    out = {}
    net.eval()
    random.seed(7)
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        loss, pm, ps, gt, pred = net(imgs, segs, classes, coors, corpus, mask)
    out.update({"r18_eval_loss": loss.float(), "r18_pred": pred.float(), "r18_gt": gt, "r18_pred_ss": ps.float()[:, :, ::4, ::4]})
    net.train()
    random.seed(7)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        loss = net(imgs, segs, classes, coors, corpus, mask)
    loss.backward()
    out["r18_train_loss"] = loss.float()
    named = dict(net.named_parameters())
    for k in ["late_fusion_net.fuse_embedding_net.linear.weight", "field_type_classification_head.category_classification_net.linear_2.weight",
              "bert_model.encoder.layer.1.output.dense.weight"]:
        out[f"r18_grad::{k}"] = named[k].grad.float().flatten()[:: max(1, named[k].numel() // 4096)][:4096]
    npz("e2e_amp.npz", **out)


def _full_net(V, tmp, name):
    """the reference's ViBERTgridNet for a tests/full_scale.py case (12-layer BERT / RoBERTa of the real dimensions, random init replaced
    by the deterministic synthetic weights by the caller)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import full_scale as F
    c = F.CASES[name]
    top = os.path.join(tmp, "full_" + name)
    os.makedirs(top, exist_ok=True)
    if c["roberta"]:
        from transformers import RobertaConfig, RobertaTokenizer
        d = os.path.join(top, c["bert"])
        os.makedirs(d, exist_ok=True)
        RobertaConfig(vocab_size=c["vocab"], max_position_embeddings=514, type_vocab_size=1, num_hidden_layers=12, hidden_dropout_prob=0.0,
                      attention_probs_dropout_prob=0.0, layer_norm_eps=1e-5).save_pretrained(d)
        voc = {"<s>": 0, "<pad>": 1, "</s>": 2, "<unk>": 3}
        voc.update({f"t{i}": i for i in range(4, c["vocab"])})
        json.dump(voc, open(os.path.join(d, "vocab.json"), "w"))
        open(os.path.join(d, "merges.txt"), "w").write("#version: 0.2\n")
        tokenizer = RobertaTokenizer(os.path.join(d, "vocab.json"), os.path.join(d, "merges.txt"))
    else:
        d = make_bert_dir(top, c["bert"], layers=12, vocab=c["vocab"])
        tokenizer = BertTokenizer(os.path.join(d, "vocab.txt"))
    cwd = os.getcwd()
    os.chdir(top)
    try:
        net = V.ViBERTgridNet(num_classes=c["ncls"], image_mean=[0.9248, 0.9224, 0.9215], image_std=[0.1532, 0.1545, 0.1536],
                              image_min_size=[c["img"]], image_max_size=c["img"], test_image_min_size=c["img"],
                              bert_model=c["bert"], tokenizer=tokenizer, backbone=c["backbone"], grid_mode="mean",
                              work_mode="eval", **F.loss_kwargs(name))
    finally:
        os.chdir(cwd)
    return net


def gen_full_chunked(V, tmp, name, check_against=None, extra=None):
    """A full-scale fixture for a batch whose single reference step does not fit this container (tests/full_scale.py `chunk`): the reference
    runs on consecutive groups of `chunk` documents -- frozen BatchNorm, plain losses: documents independent, every group the same number of
    pixels and segments -- and loss / gradients of the batch are the means over the groups (gradients accumulate in `.grad` over the groups'
    backward passes, as the reference's own autograd would sum them).  check_against: an existing DIRECT fixture of the same batch -- prints and
    returns (median, max) relative L2 distance of every stored gradient sample instead of writing a file."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import full_scale as F
    import time
    c = F.CASES[name]
    assert c.get("plain") and c.get("bn_frozen") and c["B"] % c["chunk"] == 0
    nch = c["B"] // c["chunk"]
    net = _full_net(V, tmp, name)
    load_synth(net)
    batch = F.inputs(name)
    segs_per = {sum(int(t.shape[0]) for t in F.chunk_of(batch, i, c["chunk"])[2]) for i in range(nch)}
    assert len(segs_per) == 1, f"groups hold different numbers of segments {segs_per}: the batch loss is not the mean of the groups'"
    out = {"checksums": np.array(F.checksums(batch))}

    def eval_pass(tag):
        net.eval()
        acc = {"loss": [], "pm": [], "ps": [], "gt": [], "pred": []}
        t0 = time.time()
        for i in range(nch):
            random.seed(7)
            with torch.no_grad():
                loss, pm, ps, gt, pred = net(*F.chunk_of(batch, i, c["chunk"]))
            acc["loss"].append(loss.double())
            acc["pm"].append(pm[:, :, 5::16, 3::16].clone())
            acc["ps"].append(ps[:, :, 5::16, 3::16].clone())
            acc["gt"].append(gt)
            acc["pred"].append(pred)
            del pm, ps
        print(name, tag, "eval forwards", round(time.time() - t0, 1), "s")
        return (torch.stack(acc["loss"]).mean(0), torch.cat(acc["pm"]), torch.cat(acc["ps"]), torch.cat(acc["gt"]), torch.cat(acc["pred"]))

    def train_pass(tag):
        net.train()
        for m in net.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.eval()
        net.zero_grad()
        losses = []
        t0 = time.time()
        for i in range(nch):
            random.seed(7)
            loss = net(*F.chunk_of(batch, i, c["chunk"]))
            (loss / nch).backward()
            losses.append(loss.detach().double())
            print(name, tag, "group", i, "train step done at", round(time.time() - t0, 1), "s", flush=True)
        return torch.stack(losses).mean(0)

    loss, pm, ps, gt, pred = eval_pass("")
    out.update(eval_loss=loss, gt=gt, pred=pred, pred_mask=pm, pred_ss=ps)
    out["train_loss"] = train_pass("")
    named = dict(net.named_parameters())
    gn = {k: (0.0 if p.grad is None else float(p.grad.double().norm())) for k, p in named.items() if not k.startswith("BERTgrid_generator.")}
    out["gradnorm_keys"] = np.array(sorted(gn.keys()))
    out["gradnorm_vals"] = np.array([gn[k] for k in sorted(gn.keys())])
    for k in sorted(gn.keys()):
        if named[k].grad is not None:
            out[f"grad::{k}"] = F.sample(named[k].grad, 1024).clone()
    if check_against is not None:
        ref = np.load(os.path.join(HERE, check_against))
        d = []
        for f in ref.files:
            if f.startswith("grad::") and f in out and "key.bias" not in f:
                a, b = out[f].double(), torch.from_numpy(ref[f]).double()
                if float(b.norm()) > 0:
                    d.append((float((a - b).norm() / b.norm()), f[6:]))
        d.sort()
        dl = abs(float(out["train_loss"]) - float(np.asarray(ref["train_loss"]).reshape(-1)[0])) / abs(float(np.asarray(ref["train_loss"]).reshape(-1)[0]))
        dp = float((out["pred"] - torch.from_numpy(ref["pred"])).abs().max())
        print(f"{name}: mean-of-groups rule vs the direct step of {check_against}: loss rel {dl:.2e}, class probabilities max abs {dp:.2e}, "
              f"{len(d)} gradients rel-L2 median {d[len(d) // 2][0]:.2e} max {d[-1][0]:.2e} ({d[-1][1]}); the direct fixture's own one-ulp noise median "
              f"{float(np.median(ref['ulpnoise_vals'])):.2e}")
        return np.array([d[len(d) // 2][0], d[-1][0], dl, dp])
    sd = net.state_dict()
    bnk = "backbone.resnet.bn1" if c["backbone"].endswith("pretrained") else "backbone.conv_1.1"
    out["bn_rm"], out["bn_rv"] = sd[bnk + ".running_mean"].clone(), sd[bnk + ".running_var"].clone()
    # the reference's own one-ulp sensitivity on the same batch (see gen_full)
    first = {k: out[f"grad::{k}"].clone() for k in sorted(gn.keys()) if f"grad::{k}" in out}
    gen = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for k, p in net.named_parameters():
            if not k.startswith("BERTgrid_generator."):
                p.mul_(1 + 1.2e-7 * torch.randn(p.shape, generator=gen))
    out["train_loss_1ulp"] = train_pass("one-ulp")
    keys, vals = [], []
    for k, a in first.items():
        b = F.sample(named[k].grad, 1024).double()
        keys.append(k)
        vals.append(float((a.double() - b).norm() / (a.double().norm() + 1e-30)))
    out["ulpnoise_keys"], out["ulpnoise_vals"] = np.array(keys), np.array(vals)
    print(name, "1-ulp gradient noise: median", float(np.median(vals)), "max", float(np.max(vals)))
    _, pm1, ps1, _, pred1 = eval_pass("one-ulp")
    out.update(pred_1ulp=pred1, pred_mask_1ulp=pm1, pred_ss_1ulp=ps1)
    if extra:
        out.update(extra)
    out["keys"] = np.array(list(shapes_of(net).keys()))
    out["key_shapes"] = np.array([str(v) for v in shapes_of(net).values()])
    npz(f"full_{name}.npz", **out)


def gen_full_amp(V, tmp, name, dtype):
    """The reference's `amp: True` step at full scale: its autocast region (pipeline/train_val_utils.py:264) around the train forward of a
    tests/full_scale.py case (frozen BatchNorm, plain losses: the every-gradient setup), on the CPU -- CUDA autocast cannot run here;
    CPU autocast to `dtype` (fp16 where this torch build's CPU kernels take it, else bfloat16) is the closest thing the reference itself
    can produce in this container: the same op list in reduced precision (conv / linear / matmul), normalisation and losses in fp32.
    Stores the loss and the sampled gradients like gen_full."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import full_scale as F
    import time
    c = F.CASES[name]
    net = _full_net(V, tmp, name)
    load_synth(net)
    batch = F.inputs(name)
    net.train()
    for m in net.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.eval()
    random.seed(7)
    t0 = time.time()
    with torch.autocast("cpu", dtype=dtype):
        loss = net(*batch)
    # `scaler.scale(train_loss).backward()` of the reference's loop (pipeline/train_val_utils.py:274) with torch.cuda.amp.GradScaler's
    # initial scale 2^16, gradients unscaled afterwards as `scaler.step` does: without it the fp16 backward underflows (a first version of
    # this fixture: the reference's autocast gradients at a median cosine of 0.987 to its own fp32 gradients)
    scale = 65536.0
    (loss * scale).backward()
    print(name, "autocast", dtype, "train step", round(time.time() - t0, 1), "s, loss", float(loss))
    out = {"checksums": np.array(F.checksums(batch)), "train_loss": loss.float(), "autocast_dtype": np.array(str(dtype)), "loss_scale": np.array(scale)}
    for k, p in net.named_parameters():
        if not k.startswith("BERTgrid_generator.") and p.grad is not None:
            out[f"grad::{k}"] = F.sample(p.grad.float() / scale, 1024)
    npz(f"full_{name}_amp.npz", **out)


def gen_full(V, tmp, name):
    """Full-scale reference runs (12-layer bert-base / roberta-base dims, real vocab sizes, resnet-34, 512x512 / 1024x1024, T=512 ->
    two windows): tests/full_scale.py defines the cases and the seeded inputs; the outputs of model/ViBERTgrid_net.py:501-544 in
    eval mode (5-tuple) and train mode (loss, gradients) are stored.  Dropout 0 (the masks of two RNGs cannot agree)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import full_scale as F
    c = F.CASES[name]
    net = _full_net(V, tmp, name)
    load_synth(net)
    batch = F.inputs(name)
    imgs, segs, classes, coors, corpus, mask = batch
    out = {"checksums": np.array(F.checksums(batch))}
    import time
    t0 = time.time()
    net.eval()
    random.seed(7)
    with torch.no_grad():
        loss, pm, ps, gt, pred = net(imgs, segs, classes, coors, corpus, mask)
    print(name, "eval forward", round(time.time() - t0, 1), "s")
    out.update(eval_loss=loss, gt=gt, pred=pred, pred_mask=pm[:, :, 5::16, 3::16], pred_ss=ps[:, :, 5::16, 3::16])
    if name == "cfg1":          # the deployment entry point on the same document (model/ViBERTgrid_net.py:470-499)
        with torch.no_grad():
            out["inference"] = net.inference(imgs, segs, coors, corpus, mask)
    net.train()
    if c.get("bn_frozen"):
        for m in net.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.eval()
    random.seed(7)
    t0 = time.time()
    loss = net(imgs, segs, classes, coors, corpus, mask)
    loss.backward()
    print(name, "train step", round(time.time() - t0, 1), "s")
    out["train_loss"] = loss
    named = dict(net.named_parameters())
    gn = {k: (0.0 if p.grad is None else float(p.grad.double().norm())) for k, p in named.items() if not k.startswith("BERTgrid_generator.")}
    out["gradnorm_keys"] = np.array(sorted(gn.keys()))
    out["gradnorm_vals"] = np.array([gn[k] for k in sorted(gn.keys())])
    for k in (sorted(gn.keys()) if c.get("plain") else F.GRAD_PICK + F.GRAD_PICK_BACKBONE[c["backbone"]]):
        if named[k].grad is not None:
            out[f"grad::{k}"] = F.sample(named[k].grad, 1024 if c.get("plain") else 4096)
    sd = net.state_dict()
    bnk = "backbone.resnet.bn1" if c["backbone"].endswith("pretrained") else "backbone.conv_1.1"
    out["bn_rm"], out["bn_rv"] = sd[bnk + ".running_mean"].clone(), sd[bnk + ".running_var"].clone()
    if c.get("plain"):
        # conditioning of the reference ITSELF: the same step with every floating-point weight moved by about one ulp
        # (w * (1 + 1.2e-7 * N(0,1)), seeded) -> per-parameter relative L2 distance of the sampled gradients.  Any implementation,
        # however exact, differs from the reference by rounding errors of this size at every operation, so this is the floor a
        # gradient comparison at this model size can be held to (train-mode BN: median 6e-3; frozen BN: 4e-4).
        first = {k: out[f"grad::{k}"].clone() for k in sorted(gn.keys()) if f"grad::{k}" in out}
        net.zero_grad()
        sd1 = load_synth(net)
        gen = torch.Generator().manual_seed(1)
        with torch.no_grad():
            for k, p in net.named_parameters():
                if not k.startswith("BERTgrid_generator."):
                    p.mul_(1 + 1.2e-7 * torch.randn(p.shape, generator=gen))
        random.seed(7)
        loss2 = net(imgs, segs, classes, coors, corpus, mask)
        loss2.backward()
        out["train_loss_1ulp"] = loss2
        keys, vals = [], []
        for k, a in first.items():
            b = F.sample(named[k].grad, 1024).double()
            keys.append(k)
            vals.append(float((a.double() - b).norm() / (a.double().norm() + 1e-30)))
        out["ulpnoise_keys"], out["ulpnoise_vals"] = np.array(keys), np.array(vals)
        print(name, "1-ulp gradient noise: median", float(np.median(vals)), "max", float(np.max(vals)))
        # ... and of the eval-mode outputs: the same one-ulp move of the weights on the freshly reloaded state (the train steps
        # above moved the BatchNorm running statistics), sampled like pred_mask / pred_ss / pred above
        net.zero_grad()
        load_synth(net)
        gen = torch.Generator().manual_seed(1)
        with torch.no_grad():
            for k, p in net.named_parameters():
                if not k.startswith("BERTgrid_generator."):
                    p.mul_(1 + 1.2e-7 * torch.randn(p.shape, generator=gen))
        net.eval()
        random.seed(7)
        with torch.no_grad():
            _, pm1, ps1, _, pred1 = net(imgs, segs, classes, coors, corpus, mask)
        out.update(pred_1ulp=pred1, pred_mask_1ulp=pm1[:, :, 5::16, 3::16], pred_ss_1ulp=ps1[:, :, 5::16, 3::16])
        for nm, a, b in (("pred_mask", pm, pm1), ("pred_ss", ps, ps1), ("pred", pred, pred1)):
            print(name, nm, "1-ulp output noise: max abs", float((a - b).abs().max()), "of max |ref|", float(a.abs().max()))
    out["keys"] = np.array(list(shapes_of(net).keys()))
    out["key_shapes"] = np.array([str(v) for v in shapes_of(net).values()])
    npz(f"full_{name}.npz", **out)


def main():
    torch.set_num_threads(8)
    install_torchvision_stub()
    sys.path.insert(0, REF)
    tmp = tempfile.mkdtemp(prefix="vbg_golden_")
    make_bert_dir(tmp, "bert-base-uncased", layers=2, vocab=1200)
    os.chdir(tmp)
    import model.BERTgrid_generator as G
    import model.ResNetFPN_ViBERTgrid as R
    import model.semantic_segmentation_head as S
    import model.ViBERTgrid_net as V
    import pipeline.custom_loss as L
    import pipeline.transform as T

    which = sys.argv[1:] or ["transform", "windows", "aggregate", "scatter", "losses", "labels", "bert", "backbone", "backbone_d", "e2e", "e2e_roberta", "losses_bce", "e2e_modes", "e2e_amp", "full_cfg2", "full_cfg4", "full_cfg5", "full_cfg2p", "full_cfg2e", "full_cfg4e", "full_cfg5e", "full_cfg3", "full_cfg3e", "full_cfg2e8", "full_cfg4e8", "full_cfg1", "full_cfg5e16", "full_cfg3e8"]
    if "transform" in which:
        gen_transform(T)
    if "windows" in which:
        gen_windows(G)
    if "aggregate" in which:
        gen_aggregate(G)
    if "scatter" in which:
        gen_scatter(G)
    if "losses" in which:
        gen_losses(L)
    if "labels" in which:
        gen_labels(S)
    if "bert" in which:
        gen_bert(tmp)
    if "backbone" in which:
        gen_backbone(R)
    if "backbone_d" in which:
        gen_backbone_d(R)
    if "e2e" in which:
        gen_e2e(V, tmp)
    if "e2e_roberta" in which:
        gen_e2e_roberta(V, tmp)
    if "losses_bce" in which:
        gen_losses_bce(L)
    if "e2e_modes" in which:
        gen_e2e_modes(V, tmp)
    if "e2e_amp" in which:
        gen_e2e_amp(V, tmp)
    for name in ("cfg2", "cfg4", "cfg5", "cfg2p", "cfg2e", "cfg4e", "cfg5e", "cfg3", "cfg3e", "cfg2e8", "cfg4e8", "cfg1", "cfg3e8"):
        if "full_" + name in which:
            gen_full(V, tmp, name)
    if "full_cfg2e8_amp" in which:     # the reference under autocast at the benchmark's batch (fp16 if the CPU kernels take it, else bf16)
        try:
            gen_full_amp(V, tmp, "cfg2e8", torch.float16)
        except Exception as e:
            print("fp16 CPU autocast failed:", type(e).__name__, str(e)[:200], "-> bfloat16")
            gen_full_amp(V, tmp, "cfg2e8", torch.bfloat16)
    if "chunk_rule" in which:          # (prints only) the mean-of-groups rule against the direct batch-8 step of full_cfg2e8.npz
        gen_full_chunked(V, tmp, "cfg2e8c", check_against="full_cfg2e8.npz")
    if "full_cfg5e16" in which:        # batch 16 x 1024^2 from eight reference steps of two documents; the rule's check rides in the fixture
        chk = gen_full_chunked(V, tmp, "cfg2e8c", check_against="full_cfg2e8.npz")
        gen_full_chunked(V, tmp, "cfg5e16", extra={"chunk_rule_check": chk})


if __name__ == "__main__":
    main()
