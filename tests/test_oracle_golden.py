"""Pin the CPU oracle (oracle/vbg_oracle.py) against vectors produced by the real reference
(tests/golden/make_golden.py).  CPU only."""
import random

import numpy as np
import pytest
import torch

import vbg_oracle as O

T = torch.from_numpy


def test_transform(golden):
    g = golden("transform.npz")
    imgs = [T(g[f"img{i}"]) for i in range(3)]
    coors = [T(g[f"coor{i}"]) for i in range(3)]
    cfg = O.NetCfg(image_min_size=(48, 64), image_max_size=80, test_image_min_size=56)
    batch, oc, sizes = O.transform(imgs, coors, cfg, training=False)
    assert np.array_equal(np.array(sizes), g["eval_sizes"])
    assert torch.allclose(batch, T(g["eval_batch"]), rtol=1e-5, atol=1e-6)
    for i in range(3):
        assert np.array_equal(oc[i].numpy(), g[f"eval_coor{i}"]), i
    # train mode draws the min side with torch's global CPU RNG exactly like the reference
    torch.manual_seed(123)
    batch, oc, sizes = O.transform(imgs, coors, cfg, training=True)
    assert np.array_equal(np.array(sizes), g["train_sizes"])
    assert torch.allclose(batch, T(g["train_batch"]), rtol=1e-5, atol=1e-6)
    for i in range(3):
        assert np.array_equal(oc[i].numpy(), g[f"train_coor{i}"]), i
    cfg2 = O.NetCfg(image_min_size=(64,), image_max_size=64, test_image_min_size=64)
    batch, oc, _ = O.transform(imgs[2:], coors[2:], cfg2, training=False)
    assert torch.allclose(batch, T(g["ident_batch"]), rtol=1e-6, atol=1e-6)
    assert np.array_equal(oc[0].numpy(), g["ident_coor"])


@pytest.mark.parametrize("Tn", [5, 509, 510, 511, 512, 1020, 1021])
def test_windows(golden, Tn):
    g = golden("windows.npz")
    corpus = T(g[f"T{Tn}_corpus"])
    mask = (corpus != 0).int()
    wins = O.bert_windows(corpus, mask)
    assert len(wins) == int(g[f"T{Tn}_nwin"]) == Tn // 510 + 1
    for w, (ids, am, cur) in enumerate(wins):
        assert np.array_equal(ids.numpy(), g[f"T{Tn}_ids{w}"])
        assert np.array_equal(am.numpy(), g[f"T{Tn}_am{w}"])
    # kept slices + mean aggregation over pairs of tokens, through the same fake encoder
    tok = torch.cat([torch.stack([ids.float(), am.float(), torch.arange(ids.shape[1]).float()[None].expand_as(ids)], -1)[:, 1:1 + cur]
                     for ids, am, cur in wins], 1)
    lens = [Tn, max(1, Tn - 3)]
    for b in range(2):
        seg = torch.arange(lens[b], dtype=torch.int32) // 2
        out = O.seg_aggregate(tok[b], mask[b], seg, "mean")
        assert np.array_equal(out.numpy(), g[f"T{Tn}_emb{b}"])


@pytest.mark.parametrize("mode", ["mean", "first"])
def test_aggregate(golden, mode):
    g = golden("aggregate.npz")
    tok, mask = T(g[f"{mode}_tok"]), T(g[f"{mode}_mask"])
    for b in range(2):
        out = O.seg_aggregate(tok[b], mask[b], T(g[f"{mode}_seg{b}"]), mode)
        assert np.array_equal(out.numpy(), g[f"{mode}_out{b}"])      # bit exact, order-sensitive sum


def test_scatter(golden):
    g = golden("scatter.npz")
    H, W = int(g["H"]), int(g["W"])
    embs = [T(g[f"emb{b}"]).clone().requires_grad_(True) for b in range(3)]
    boxes = [T(g[f"box{b}"]) for b in range(3)]
    grid = O.grid_scatter(embs, boxes, H, W, 8)
    assert np.array_equal(grid.detach().numpy(), g["grid"])          # bit exact
    grid.backward(T(g["gout"]))
    for b in range(3):
        ge = embs[b].grad if embs[b].grad is not None else torch.zeros_like(embs[b])
        assert torch.allclose(ge, T(g[f"gemb{b}"]), rtol=1e-5, atol=1e-6)


def test_labels(golden):
    g = golden("labels.npz")
    coors = [T(g[f"coor{b}"]) for b in range(2)]
    classes = [T(g[f"class{b}"]) for b in range(2)]
    pn, cls = O.label_raster(classes, coors, 32, 64)
    assert np.array_equal(pn.numpy(), g["pos_neg"])
    assert np.array_equal(cls.numpy(), g["cls"])


def test_losses(golden):
    g = golden("losses.npz")
    x = T(g["rs_x"]).clone().requires_grad_(True)
    random.seed(77)
    l = O.ce_random_sample(x, T(g["rs_t"]), [256, 512, 256])
    assert l.dtype == torch.float64 and l.shape == (1,)
    assert torch.allclose(l, T(g["rs_loss"]), rtol=1e-6)
    l.backward()
    assert torch.allclose(x.grad, T(g["rs_grad"]), rtol=1e-5, atol=1e-8)

    x = T(g["oh_x"]).clone().requires_grad_(True)
    l = O.ce_ohem(x, T(g["oh_t"]), 32, 32)
    assert torch.allclose(l, T(g["oh_loss"]), rtol=1e-6)
    l.backward()
    assert torch.allclose(x.grad, T(g["oh_grad"]), rtol=1e-5, atol=1e-8)

    x = T(g["ohr_x"]).clone().requires_grad_(True)
    random.seed(5)
    l = O.ce_ohem(x, T(g["ohr_t"]), 16, 16, T(g["ohr_w"]), rand=True)
    assert torch.allclose(l, T(g["ohr_loss"]), rtol=1e-6)
    l.backward()
    assert torch.allclose(x.grad, T(g["ohr_grad"]), rtol=1e-5, atol=1e-8)

    # heavy ties (replicated logits): the reference's CPU sort order must be reproduced
    x = T(g["tie_x"]).clone().requires_grad_(True)
    l = O.ce_ohem(x, T(g["tie_t"]), 40, 40)
    assert torch.allclose(l, T(g["tie_loss"]), rtol=1e-6)
    l.backward()
    assert torch.allclose(x.grad, T(g["tie_grad"]), rtol=1e-5, atol=1e-8)

    x = T(g["few_x"]).clone().requires_grad_(True)
    random.seed(1)
    l = O.ce_ohem(x, T(g["few_t"]), 16, 16, rand=True)
    assert torch.allclose(l, T(g["few_loss"]), rtol=1e-6)
    l.backward()
    assert torch.allclose(x.grad, T(g["few_grad"]), rtol=1e-5, atol=1e-8)


def test_losses_bce(golden):
    """a13: BCELossRandomSample (categories by the sign of the prediction) and BCELossOHEM against the reference"""
    g = golden("losses_bce.npz")
    for tag, seed, sl in (("rs", 3, [32, 48]), ("rs2", 4, [16, 16])):
        x = T(g[tag + "_x"]).clone().requires_grad_(True)
        random.seed(seed)
        l = O.bce_random_sample(x, T(g[tag + "_t"]), sl)
        assert l.dtype == torch.float64 and l.shape == (1,)
        assert torch.allclose(l, T(g[tag + "_loss"]), rtol=1e-6)
        l.backward()
        assert torch.allclose(x.grad, T(g[tag + "_grad"]), rtol=1e-5, atol=1e-8)
    for tag, seed, kp, kn, rnd in (("oh", None, 32, 32, False), ("ohr", 5, 16, 16, True), ("few", None, 16, 4, False)):
        x = T(g[tag + "_x"]).clone().requires_grad_(True)
        if seed is not None:
            random.seed(seed)
        l = O.bce_ohem(x, T(g[tag + "_t"]), kp, kn, rnd)
        assert l.dim() == 0 and torch.allclose(l, T(g[tag + "_loss"]), rtol=1e-6)
        l.backward()
        assert torch.allclose(x.grad, T(g[tag + "_grad"]), rtol=1e-5, atol=1e-8)


def test_crf_small():
    """CRF forward algorithm / Viterbi of the oracle against brute force over all tag sequences"""
    import itertools
    g = torch.Generator().manual_seed(9)
    T_, n = 5, 4
    feats, trans = torch.randn(n, T_, generator=g), torch.randn(T_, T_, generator=g)
    start, stop = 3, 4
    scores = {}
    for seq in itertools.product(range(T_), repeat=n):
        scores[seq] = float(O.crf_score(feats, torch.tensor(seq), trans, start, stop))
    logz = torch.logsumexp(torch.tensor(list(scores.values())), 0)
    assert abs(float(O.crf_forward_alg(feats, trans, start, stop)) - float(logz)) < 1e-4
    best = max(scores, key=scores.get)
    sc, path = O.crf_viterbi(feats, trans, start, stop)
    assert tuple(path) == best and abs(float(sc) - scores[best]) < 1e-5


def modes_cfg(mode):
    return O.NetCfg(num_classes=5, image_min_size=(96,), image_max_size=128, test_image_min_size=96, backbone="resnet_18_fpn",
                    num_hard_positive_main_1=4, num_hard_negative_main_1=4, num_hard_positive_main_2=3,
                    num_hard_negative_main_2=3, loss_aux_sample_list=(64, 128, 64), num_hard_positive_aux=64,
                    num_hard_negative_aux=64, ohem_random=True, bert=O.BertCfg(layers=2, dropout=0.0),
                    classifier_mode=mode, layer_mode="multi" if mode == "full" else "single")


def modes_state(g, mode):
    cfg = modes_cfg(mode)
    sd = O.synth_state_dict(O.state_shapes(cfg, vocab=1200))
    if mode == "crf":
        sd["field_type_classification_head.crf_layer.transitions"] = T(g["crf_transitions"]).clone()
    return cfg, sd


@pytest.mark.parametrize("mode", ["full", "crf"])
def test_e2e_modes(golden, mode):
    """a9 two-stage seg head, a12 full / crf classifiers, a13 BCE losses: whole model against the reference run"""
    g = golden("e2e_modes.npz")
    batch = _e2e_inputs(golden("e2e.npz"))
    cfg, sd = modes_state(g, mode)
    ref_shapes = {str(k): str(v) for k, v in zip(g[mode + "_keys"], g[mode + "_key_shapes"])}
    extra_ok = lambda k: k.endswith("position_ids") or k.endswith("token_type_ids")
    shapes = O.state_shapes(cfg, vocab=1200)
    assert {k for k in set(ref_shapes) - set(shapes) if not extra_ok(k)} == set() and set(shapes) - set(ref_shapes) == set()
    for k, v in shapes.items():
        assert ref_shapes[k] == str(tuple(v)), (k, ref_shapes[k], v)
    random.seed(7)
    with torch.no_grad():
        loss, pm, ps, gt, pred = O.forward({k: v.clone() for k, v in sd.items()}, cfg, *batch, training=False)
    assert np.array_equal(gt.numpy(), g[mode + "_gt"])
    if mode == "crf":
        assert np.array_equal(pred.numpy(), g["crf_pred"])                      # Viterbi tags
    else:
        assert torch.allclose(pred, T(g["full_pred"]), rtol=1e-4, atol=1e-5)
    assert torch.allclose(ps[:, :, ::8, ::8], T(g[mode + "_pred_ss"]), rtol=1e-3, atol=1e-4)
    assert torch.allclose(loss.float(), T(g[mode + "_eval_loss"]).float(), rtol=1e-4)
    sdg = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
    random.seed(7)
    out = O.forward(sdg, cfg, *batch, training=True)
    out[0].backward()
    assert torch.allclose(out[0].detach().float(), T(g[mode + "_train_loss"]).float(), rtol=1e-4)
    for k, v in zip([str(k) for k in g[mode + "_gradnorm_keys"]], g[mode + "_gradnorm_vals"]):
        gr = sdg[k].grad
        n = 0.0 if gr is None else float(gr.double().norm())
        assert abs(n - v) <= 2e-3 * max(abs(v), 1e-3), (k, n, v)


def test_bert(golden):
    g = golden("bert.npz")
    bc = O.BertCfg(layers=int(g["layers"]), dropout=0.0)
    sd = O.synth_state_dict(O.bert_shapes("", bc, int(g["vocab"])))
    h = O.bert_forward(sd, "", T(g["ids"]), T(g["am"]), bc)
    ref = T(g["hidden"])
    am = T(g["am"]).bool()
    # only mask==1 positions are ever consumed downstream
    assert torch.allclose(h[am], ref[am], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("kind", ["resnet_18_fpn", "resnet_34_fpn_pretrained"])
def test_backbone(golden, kind):
    g = golden("backbone.npz")
    sd = O.synth_state_dict(O.backbone_shapes(kind, 768, prefix=""))
    x, grid = T(g["x"]), T(g["grid"])
    with torch.no_grad():
        ev = O.backbone_forward(sd, x, grid, kind, False, prefix="")
        assert torch.allclose(ev, T(g[kind + "_eval"]), rtol=1e-4, atol=1e-4)
        tr = O.backbone_forward(sd, x, grid, kind, True, prefix="")
        assert torch.allclose(tr, T(g[kind + "_train"]), rtol=1e-3, atol=1e-3)
    key = "conv_1.1" if kind == "resnet_18_fpn" else "resnet.bn1"
    assert torch.allclose(sd[key + ".running_mean"], T(g[kind + "_rm"]), rtol=1e-5, atol=1e-6)
    assert torch.allclose(sd[key + ".running_var"], T(g[kind + "_rv"]), rtol=1e-5, atol=1e-6)


def test_backbone_resnet_d(golden):
    """a6 DBlock variant (avg-pool projection shortcut): oracle vs the reference's resnet_18_D_fpn, incl. the key inventory"""
    g, gi = golden("backbone_d.npz"), golden("backbone.npz")
    kind = "resnet_18_D_fpn"
    shapes = O.backbone_shapes(kind, 768, prefix="")
    ref_keys = set(str(k) for k in g[kind + "_keys"])
    assert ref_keys == set(shapes.keys())
    sd = O.synth_state_dict(shapes)
    x, grid = T(gi["x"]), T(gi["grid"])
    with torch.no_grad():
        assert torch.allclose(O.backbone_forward(sd, x, grid, kind, False, prefix=""), T(g[kind + "_eval"]), rtol=1e-4, atol=1e-4)
        assert torch.allclose(O.backbone_forward(sd, x, grid, kind, True, prefix=""), T(g[kind + "_train"]), rtol=1e-3, atol=1e-3)
    assert torch.allclose(sd["conv_4_x.0.conv_shortcut.2.running_mean"], T(g[kind + "_rm"]), rtol=1e-5, atol=1e-6)
    assert torch.allclose(sd["conv_4_x.0.conv_shortcut.2.running_var"], T(g[kind + "_rv"]), rtol=1e-5, atol=1e-6)


def _e2e_inputs(g):
    imgs = [T(g[f"img{b}"]) for b in range(2)]
    coors = [T(g[f"coor{b}"]) for b in range(2)]
    segs = [T(g[f"seg{b}"]) for b in range(2)]
    classes = [T(g[f"class{b}"]) for b in range(2)]
    return imgs, segs, classes, coors, T(g["corpus"]), T(g["mask"])


def e2e_cfg(backbone):
    return O.NetCfg(num_classes=5, image_min_size=(96,), image_max_size=128, test_image_min_size=96, backbone=backbone,
                    num_hard_positive_main_1=4, num_hard_negative_main_1=4, num_hard_positive_main_2=6,
                    num_hard_negative_main_2=6, loss_aux_sample_list=(64, 128, 64), num_hard_positive_aux=64,
                    num_hard_negative_aux=64, ohem_random=True, bert=O.BertCfg(layers=2, dropout=0.0))


def roberta_cfg():
    return O.NetCfg(num_classes=4, image_min_size=(96,), image_max_size=128, test_image_min_size=96, backbone="resnet_18_fpn",
                    num_hard_positive_main_1=4, num_hard_negative_main_1=4, num_hard_positive_main_2=6,
                    num_hard_negative_main_2=6, loss_aux_sample_list=(64, 128, 64), num_hard_positive_aux=64,
                    num_hard_negative_aux=64, ohem_random=True, bert=O.BertCfg(layers=2, dropout=0.0, roberta=True, ln_eps=1e-5))


def roberta_inputs(golden):
    g, e = golden("e2e_roberta.npz"), golden("e2e.npz")
    imgs, segs, _, coors, corpus, mask = _e2e_inputs(e)
    return g, (imgs, segs, [T(g["classes0"]), T(g["classes1"])], coors, corpus, mask)


def test_e2e_roberta(golden):
    """cfg3 / cfg5 flavour: RobertaModel container (514 positions, padding_idx position rule, 1 token type, LN eps 1e-5) and
    4 classes, against the reference run of tests/golden/make_golden.py::gen_e2e_roberta"""
    g, batch = roberta_inputs(golden)
    cfg = roberta_cfg()
    shapes = O.state_shapes(cfg, vocab=1300, max_pos=514, type_vocab=1)
    ref_shapes = {str(k): str(v) for k, v in zip(g["keys"], g["key_shapes"])}
    extra_ok = lambda k: k.endswith("position_ids") or k.endswith("token_type_ids")
    assert {k for k in set(ref_shapes) - set(shapes) if not extra_ok(k)} == set()
    assert set(shapes) - set(ref_shapes) == set()
    for k, v in shapes.items():
        assert ref_shapes[k] == str(tuple(v)), (k, ref_shapes[k], v)
    sd = O.synth_state_dict(shapes)
    random.seed(7)
    with torch.no_grad():
        loss, pm, ps, gt, pred = O.forward({k: v.clone() for k, v in sd.items()}, cfg, *batch, training=False)
    assert np.array_equal(gt.numpy(), g["gt"])
    assert torch.allclose(pred, T(g["pred"]), rtol=1e-4, atol=1e-5)
    assert torch.allclose(ps[:, :, ::8, ::8], T(g["pred_ss"]), rtol=1e-3, atol=1e-4)
    assert torch.allclose(loss, T(g["eval_loss"]), rtol=1e-4)
    sdg = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
    random.seed(7)
    out = O.forward(sdg, cfg, *batch, training=True)
    out[0].backward()
    assert torch.allclose(out[0].detach(), T(g["train_loss"]), rtol=1e-4)
    for k, v in zip([str(k) for k in g["gradnorm_keys"]], g["gradnorm_vals"]):
        gr = sdg[k].grad
        n = 0.0 if gr is None else float(gr.double().norm())
        assert abs(n - v) <= 2e-3 * max(abs(v), 1e-3), (k, n, v)


@pytest.mark.parametrize("tag,backbone", [("r18", "resnet_18_fpn"), ("r34p", "resnet_34_fpn_pretrained")])
def test_e2e(golden, tag, backbone):
    g = golden("e2e.npz")
    cfg = e2e_cfg(backbone)
    shapes = O.state_shapes(cfg, vocab=1200)
    # boundary: the state_dict key inventory equals the reference's
    ref_keys = set(str(k) for k in g[f"{tag}_keys"])
    mine = set(shapes.keys())
    extra_ok = lambda k: k.endswith("position_ids") or k.endswith("token_type_ids")
    assert {k for k in ref_keys - mine if not extra_ok(k)} == set()
    assert mine - ref_keys == set()
    sd = O.synth_state_dict(shapes)
    imgs, segs, classes, coors, corpus, mask = _e2e_inputs(g)

    random.seed(7)
    with torch.no_grad():
        loss, pm, ps, gt, pred = O.forward({k: v.clone() for k, v in sd.items()}, cfg, imgs, segs, classes, coors, corpus, mask, training=False)
    assert np.array_equal(gt.numpy(), g[f"{tag}_gt"])
    assert torch.allclose(pred, T(g[f"{tag}_pred"]), rtol=1e-4, atol=1e-5)
    assert torch.allclose(pm[:, :, ::4, ::4], T(g[f"{tag}_pred_mask"]), rtol=1e-3, atol=1e-4)
    assert torch.allclose(ps[:, :, ::4, ::4], T(g[f"{tag}_pred_ss"]), rtol=1e-3, atol=1e-4)
    assert loss.dtype == torch.float64 and loss.shape == (1,)
    assert torch.allclose(loss, T(g[f"{tag}_eval_loss"]), rtol=1e-4)

    # train mode (batch-stat BN), loss + gradients
    sdg = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
    random.seed(7)
    out = O.forward(sdg, cfg, imgs, segs, classes, coors, corpus, mask, training=True)
    out[0].backward()
    assert torch.allclose(out[0].detach(), T(g[f"{tag}_train_loss"]), rtol=1e-4)
    keys = [str(k) for k in g[f"{tag}_gradnorm_keys"]]
    vals = g[f"{tag}_gradnorm_vals"]
    for k, v in zip(keys, vals):
        gr = sdg[k].grad
        n = 0.0 if gr is None else float(gr.double().norm())
        assert abs(n - v) <= 2e-3 * max(abs(v), 1e-3), (k, n, v)
    for name in g.files:
        if name.startswith(f"{tag}_grad::"):
            k = name.split("::")[1]
            gr = sdg[k].grad
            samp = gr.flatten()[:: max(1, gr.numel() // 4096)][:4096]
            ref = T(g[name])
            rel = float((samp - ref).norm() / ref.norm())
            assert rel < 5e-3, (k, rel)


def test_oracle_full_scale_vs_reference(golden):
    """The oracle at FULL scale (12-layer bert-base, vocab 30522, resnet-34 torchvision layout, 512x512, two token windows, ragged
    second document) against the reference's outputs in tests/golden/full_cfg2e.npz (default plain losses, frozen BatchNorm in the
    training step): class probabilities, both losses and every sampled parameter gradient."""
    import random
    import full_scale as F
    name = "cfg2e"
    g, c, cfg = golden(f"full_{name}.npz"), F.CASES[name], F.net_cfg(name)
    batch = F.inputs(name)
    assert np.array_equal(np.array(F.checksums(batch)), g["checksums"])
    sd = O.synth_state_dict(O.state_shapes(cfg, vocab=c["vocab"], max_pos=c["max_pos"], type_vocab=c["type_vocab"], dup_bert=False))
    random.seed(7)
    with torch.no_grad():
        loss, pm, ps, gt, pred = O.forward({k: v.clone() for k, v in sd.items()}, cfg, *batch, training=False)
    assert np.array_equal(gt.numpy(), g["gt"])
    assert torch.allclose(pred, T(g["pred"]), rtol=1e-4, atol=1e-5)
    assert torch.allclose(ps[:, :, 5::16, 3::16], T(g["pred_ss"]), rtol=1e-4, atol=1e-5)
    assert abs(float(loss) - float(np.asarray(g["eval_loss"]).reshape(-1)[0])) <= 1e-5 * abs(float(loss))
    sdg = {k: (v.requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd.items()}
    O.BN_FROZEN = True
    try:
        random.seed(7)
        tl = O.forward(sdg, cfg, *batch, training=True)[0]
        tl.backward()
    finally:
        O.BN_FROZEN = False
    assert abs(float(tl.detach()) - float(np.asarray(g["train_loss"]).reshape(-1)[0])) <= 1e-5 * abs(float(tl.detach()))
    bad = []
    for f in g.files:
        if f.startswith("grad::") and "key.bias" not in f:
            k = f[6:]
            a, b = F.sample(sdg[k].grad, 1024).double(), T(g[f]).double()
            r = float((a - b).norm() / (b.norm() + 1e-30))
            if r > 1e-3:
                bad.append((k, r))
    assert not bad, bad[:8]
