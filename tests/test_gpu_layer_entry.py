"""The one-call encoder layer (include/vbg.h vbg_bert_layer_fwd, csrc/encoder.hip) launches the same seven kernels with the same
descriptors as the per-launch path of vbg/functions.py BertLayerFn.forward: results must be BIT-identical in every form the layer runs
(three bf16 planes / fp16 pairs / the one-product forms of an autocast region; with and without dropout; training and inference), and
the entry must actually be the path taken.  Needs a real MI355X."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_gpu_model import build_product, load_synth, to_dev
from test_oracle_golden import _e2e_inputs, e2e_cfg

LAYERS = 2


def _net(tmp_path, dropout):
    cfg = e2e_cfg("resnet_18_fpn")
    net = build_product(tmp_path, "resnet_18_fpn", cfg, layers=LAYERS, dropout=dropout)
    load_synth(net, cfg, 1200)
    return net


def _batch(dev):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "e2e.npz"))
    return to_dev(_e2e_inputs(g), dev)


@pytest.mark.parametrize("form", ["bf16x3", "pair", "amp"])
@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_training_step_is_bit_identical_through_the_layer_entry(tmp_path, form, dropout):
    from vbg import ops
    from vbg.optim import FusedAdamW, FusedSGD, split_parameters
    dev = torch.device("cuda")
    net = _net(tmp_path, dropout).to(dev).train()
    cnn, bert = split_parameters(net)
    opts = [FusedSGD(cnn, dev, lr=0.0), FusedAdamW(bert, dev, lr=0.0)]
    dbatch = _batch(dev)
    gen = net.BERTgrid_generator
    was_entry, was_overlap = ops._LAYER_ENTRY[0], ops.overlap_enabled()
    ops.set_overlap(False)                       # one stream: the comparison is about the launches, not their interleaving
    ops.set_pair(True, force=form != "bf16x3")   # (the fixture is smaller than the tiles the pair form waits for)
    if form == "bf16x3":
        ops.set_pair(False)
    res = {}
    try:
        for run, entry in enumerate((False, True, False)):
            ops.set_layer_entry(entry)
            for o in opts:
                o.zero_grad()
            gen._step_seed = 41
            random.seed(7)
            log = ops.dispatch_log(True)
            with torch.autocast("cuda", dtype=torch.float16, enabled=form == "amp"):
                loss = net(*dbatch)
            ops.dispatch_log(False)
            loss.backward()
            torch.cuda.synchronize()
            assert log.get("bert_layer_fwd:entry", 0) == (LAYERS if entry else 0), log
            want = {"bf16x3": "plane_gemm:bf16x3", "pair": "plane_gemm:pair", "amp": "plane_gemm:onep"}[form]
            assert log.get(want, 0) >= 3 * LAYERS, log
            res[run] = (float(loss.detach()), dict(log), [o.group.gflat.clone() for o in opts])
    finally:
        ops.set_layer_entry(was_entry); ops.set_overlap(was_overlap); ops.set_pair(True, force=False); ops.set_amp(False)
    (l0, log0, g0), (l1, log1, g1), (l2, log2, g2) = res[0], res[1], res[2]
    assert l0 == l1 == l2, (l0, l1, l2)          # the forward is deterministic: same kernels, same arguments -> the same bits
    log1.pop("bert_layer_fwd:entry")
    assert log0 == log1 == log2, (log0, log1)    # ... and the same kernel families / tiles, launch for launch
    for a, b, c in zip(g0, g1, g2):
        assert float(a.abs().max()) > 0
        # backward: float atomics, order differs run to run; the third run is the per-launch path again: the yardstick.  Inside an
        # autocast region two runs of the SAME path already differ by the fp16 rounding error itself: a difference d in a gradient flips
        # the rounding of a fraction d / 2^-11 of the elements of the next product's operand, which moves that product's result by
        # sqrt(d 2^-11) -- 5e-8 of atomics noise is 5e-3 six nodes later (tools/bwd_diff.py, profiles/r06_amp_reproducibility.txt)
        tol = 2e-2 if form == "amp" else 1e-5
        noise = float((a - c).norm() / a.norm())
        assert noise < tol, noise
        assert float((a - b).norm() / a.norm()) < max(tol, 3 * noise)


@pytest.mark.parametrize("pair", [False, True])
def test_inference_is_bit_identical_and_runs_one_qkv_product(tmp_path, pair):
    """no flat buffers in evaluation mode: the stacked planes of the three projections come from the cache on the weights
    (vbg.ops.stacked_qkv), refreshed when a weight changes"""
    from vbg import ops
    dev = torch.device("cuda")
    net = _net(tmp_path, 0.0).to(dev).eval()
    imgs, segs, classes, coors, corpus, mask = _batch(dev)
    was = ops._LAYER_ENTRY[0]
    ops.set_pair(True, force=pair)
    if not pair:
        ops.set_pair(False)
    out, logs = {}, {}
    try:
        with torch.no_grad():
            for entry in (False, True):
                ops.set_layer_entry(entry)
                log = ops.dispatch_log(True)
                out[entry] = net.inference(imgs, segs, coors, corpus, mask).clone()
                ops.dispatch_log(False)
                assert log.pop("bert_layer_fwd:entry", 0) == (LAYERS if entry else 0), log
                logs[entry] = dict(log)
            assert logs[False] == logs[True], logs            # the same kernel families and tiles, launch for launch
            assert torch.equal(out[False], out[True])
            for layer in net.BERTgrid_generator.model.encoder.layer:          # one Q/K/V product per layer, from the stacked planes
                cache = layer.attention.self.query.weight.__dict__.get("_vbg_wplanes", {})
                assert any(k[0] == "qkv" for k in cache), list(cache)
            # a weight updated in place is seen by the cache
            q = net.BERTgrid_generator.model.encoder.layer[0].attention.self.query.weight
            q.mul_(1.5)
            moved = net.inference(imgs, segs, coors, corpus, mask)
            assert not torch.equal(moved, out[True])
            q.div_(1.5)
            back = net.inference(imgs, segs, coors, corpus, mask)
            assert torch.allclose(back, out[True], rtol=1e-5, atol=1e-6)
    finally:
        ops.set_layer_entry(was); ops.set_pair(True, force=False)


def test_inference_inside_autocast_takes_the_one_product_forms_on_the_small_tile(tmp_path):
    """`amp: True` validation / inference of a document or two: the encoder's products run form 2 on the 64 x 64 tile (round 6), through
    the layer entry; probabilities stay within the fp16 operand rounding of the fp32-grade call"""
    from vbg import ops
    dev = torch.device("cuda")
    net = _net(tmp_path, 0.0).to(dev).eval()
    imgs, segs, classes, coors, corpus, mask = _batch(dev)
    with torch.no_grad():
        ref = net.inference(imgs, segs, coors, corpus, mask).clone()
        log = ops.dispatch_log(True)
        with torch.autocast("cuda", dtype=torch.float16):
            got = net.inference(imgs, segs, coors, corpus, mask).clone()
        ops.dispatch_log(False)
    ops.set_amp(False)
    assert log.get("bert_layer_fwd:entry", 0) == LAYERS and log.get("plane_gemm:onep", 0) >= 3 * LAYERS and log.get("attn:onep", 0) == LAYERS, log
    assert log.get("plane_gemm:tile64004", 0) >= 3 * LAYERS, log
    assert bool(torch.isfinite(got).all()) and torch.allclose(got.sum(1), torch.ones(got.shape[0], device=dev), atol=1e-5)
    assert float((got - ref).abs().max()) < 2e-2 and float((got - ref).abs().max()) > 0.0


def test_layer_entry_rejects_a_descriptor_without_its_operands():
    import ctypes as C
    from vbg.lib import BertLayerFwdDesc, lib
    d = BertLayerFwdDesc()
    d.ntok, d.hidden, d.inter, d.heads = 128, 768, 3072, 12
    assert lib.vbg_bert_layer_fwd(C.byref(d), None) == -1          # argument error, nothing launched
    assert lib.vbg_bert_layer_fwd(None, None) == -1
