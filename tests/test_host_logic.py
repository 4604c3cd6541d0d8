"""Host-side logic that needs no GPU kernels (the library itself must load: there is no CPU fallback, `vbg.lib` raises without
libvbg.so): optimizer checkpoint indices, amax slot pool, dispatch predicates, reducer construction errors."""
import os
import socket

import pytest
import torch
import torch.distributed as dist


class _Toy(torch.nn.Module):
    """parameter names as in the product: a `bert_model` group with a never-used `pooler`, a CNN group with a never-used `resnet.fc`"""

    def __init__(self):
        super().__init__()
        self.bert_model = torch.nn.ModuleDict({"embeddings": torch.nn.Linear(4, 4), "pooler": torch.nn.Linear(4, 4), "encoder": torch.nn.Linear(4, 2)})
        self.backbone = torch.nn.ModuleDict({"resnet": torch.nn.ModuleDict({"conv1": torch.nn.Linear(3, 3), "fc": torch.nn.Linear(3, 5)}),
                                             "fuse": torch.nn.Linear(3, 2)})


def test_optimizer_checkpoint_indices_follow_the_reference_list():
    """torch.optim checkpoints key their state by index into the optimizer's parameter list; the reference keeps the tensors that never
    receive a gradient in that list (train_SROIE.py:215-221).  split_parameters() keeps them OUT of the flat buffers but remembers the
    full order, and state_dict / load_state_dict use the reference's indices."""
    from vbg.optim import FusedSGD, FlatGroup, split_parameters, _FlatOptimizer
    net = _Toy()
    cnn, bert = split_parameters(net)
    assert [n for n, _ in bert] == ["bert_model.embeddings.weight", "bert_model.embeddings.bias", "bert_model.encoder.weight", "bert_model.encoder.bias"]
    assert bert.ref_names == ["bert_model.embeddings.weight", "bert_model.embeddings.bias", "bert_model.pooler.weight", "bert_model.pooler.bias",
                              "bert_model.encoder.weight", "bert_model.encoder.bias"]
    assert "backbone.resnet.fc.weight" in cnn.ref_names and all("fc" not in n for n, _ in cnn)
    opt = FusedSGD(cnn, torch.device("cpu"), lr=0.1, momentum=0.9)
    ref = [p for n, p in net.named_parameters() if "bert_model" not in n]            # the reference's params_cnn
    t = torch.optim.SGD(ref, lr=0.1, momentum=0.9)
    for p in ref:
        if p.shape != (5, 3) and p.shape != (5,):                                    # fc never gets a gradient
            p.grad = torch.ones_like(p)
    t.step()
    sd_t = t.state_dict()
    assert sorted(sd_t["state"].keys()) == [0, 1, 4, 5]                              # fc.weight / fc.bias (2, 3) have no state
    opt.load_state_dict(sd_t)                                                        # reference checkpoint -> fused optimizer
    assert opt.steps == 1
    named = dict(cnn)
    for i, n in enumerate(cnn.ref_names):
        if n in named:
            mine = opt.group.view(opt.mom, opt.group.names.index(n))
            assert torch.equal(mine, sd_t["state"][i]["momentum_buffer"]), n
    sd_f = opt.state_dict()                                                          # ... and back
    assert sorted(sd_f["state"].keys()) == [0, 1, 4, 5] and sd_f["param_groups"][0]["params"] == list(range(6))
    t2 = torch.optim.SGD(ref, lr=0.1, momentum=0.9)
    t2.load_state_dict(sd_f)
    for i in (0, 1, 4, 5):
        assert torch.equal(t2.state_dict()["state"][i]["momentum_buffer"], sd_t["state"][i]["momentum_buffer"])
    # a checkpoint over a different parameter list is refused, not mis-assigned
    bad = torch.optim.SGD(ref[:4], lr=0.1, momentum=0.9).state_dict()
    with pytest.raises(ValueError):
        opt.load_state_dict(bad)


def test_amax_slot_pool():
    """amax slots (include/vbg.h VBG_AMAX_WORDS x VBG_AMAX_STRIDE): fresh, zero, disjoint, and never rewound -- a slot saved for
    backward stays valid when the pool is exhausted"""
    from vbg import ops
    dev = torch.device("cpu")
    ops._AMAX_POOL.clear()
    a, b = ops.amax_slot(dev), ops.amax_slot(dev)
    assert a.numel() == ops.AMAX_WORDS * ops.AMAX_STRIDE == 2048 and a.dtype == torch.int32
    assert int(a.abs().max()) == 0 and a.data_ptr() + 4 * 2048 == b.data_ptr()
    a[5 * ops.AMAX_STRIDE] = 77
    for _ in range(300):                                                             # exhausts the 256-slot pool
        s = ops.amax_slot(dev)
        assert int(s.abs().max()) == 0
    assert int(a[5 * ops.AMAX_STRIDE]) == 77                                          # the old pool was replaced, not cleared


def test_dispatch_predicates():
    from vbg import ops
    # wide 3x3 / stride-1 convolutions of cfg2 (batch 8): row-reuse kernels, fp16 form for forward, input and weight gradient
    assert ops.conv3_ok(8, 128, 128, 256, 256, 3, 3, 1, 1, fwd=True) and ops.conv3_f16_bwd_ok(8, 128, 128, 256, 256, 3, 3, 1, 1)
    assert ops.conv3_f16_wgrad_ok(8, 128, 128, 256, 256, 3, 3, 1, 1) and ops.conv3_f16_wgrad_ok(8, 16, 16, 512, 512, 3, 3, 1, 1)
    assert not ops.conv3_ok(8, 128, 128, 256, 256, 3, 3, 2, 1)
    # 64 -> 64 weight gradients: the row-reuse kernel's [64 x 9 x 64] blocks since the end of round 5 (64 filters need a multiple of 64 channels)
    assert ops.conv3_f16_wgrad_ok(8, 128, 128, 64, 64, 3, 3, 1, 1) and not ops.conv3_f16_wgrad_ok(8, 128, 128, 32, 64, 3, 3, 1, 1)
    # the late trunk stages (256 channels at 32 x 32: 128 tiles, 512 at 16 x 16: 64): forward and input gradient on the row-reuse
    # kernel with several workgroups per tile; without that form the input gradient falls back to the generic kernel
    assert ops.conv3_ok(8, 32, 32, 256, 256, 3, 3, 1, 1, fwd=True) and ops.conv3_ok(8, 32, 32, 256, 256, 3, 3, 1, 1)
    assert ops.conv3_split(8, 32, 32, 256, 256) == 3 and ops.conv3_split(8, 16, 16, 512, 512) == 4 and ops.conv3_split(8, 128, 128, 256, 256) == 1
    ops._CONV3_SPLITK[0] = False
    try:
        assert ops.conv3_ok(8, 32, 32, 256, 256, 3, 3, 1, 1, fwd=True) and not ops.conv3_ok(8, 32, 32, 256, 256, 3, 3, 1, 1)
        assert not ops.conv3_ok(8, 16, 16, 512, 512, 3, 3, 1, 1, fwd=True)
    finally:
        ops._CONV3_SPLITK[0] = True
    # 7 x 7 region maps (two images per tile) and the 64-filter stage
    assert ops.conv3_ok(1024, 7, 7, 256, 256, 3, 3, 1, 1) and ops.conv3_f16_wgrad_ok(1024, 7, 7, 256, 256, 3, 3, 1, 1)
    assert not ops.conv3_ok(40, 7, 7, 256, 256, 3, 3, 1, 1, fwd=True)               # a handful of regions: the generic kernel
    assert ops.conv3_ok(8, 128, 128, 64, 64, 3, 3, 1, 1) and not ops.conv3_ok(1, 64, 64, 64, 64, 3, 3, 1, 1)
    # the fp16-pair plane products need the 8-wave tiles: batch 8 takes them, a single document does not (unless forced)
    assert ops.pair_tile(4128, 768) == 128129 and ops.pair_tile(4128, 3072, True) == 256128 and ops.pair_tile(516, 768) == 0
    ops.set_pair(True, force=True)
    try:
        assert ops.pair_tile(516, 768) == 128129
    finally:
        ops.set_pair(True, force=False)
    ops.set_pair(False)
    try:
        assert not ops.pair_enabled() and not ops.pair_bwd_enabled()
    finally:
        ops.set_pair(True)
    # an autocast region keeps the fast paths (their one-product forms) unless VBG_AMP_FAST=0 / set_amp_fast(False) hands every product
    # of the region to the generic kernels, as rounds 1-3 did
    assert not ops.amp_one_product()
    with ops.amp_scope(True):
        assert ops.amp_one_product() and ops.pair_enabled() and ops.planes_enabled() and ops.conv3_ok(8, 128, 128, 256, 256, 3, 3, 1, 1)
        assert ops.conv3_f16_bwd_enabled() and ops.conv3_f16_wgrad_ok(8, 128, 128, 256, 256, 3, 3, 1, 1)
        ops.set_amp_fast(False)
        try:
            assert not ops.amp_one_product() and not ops.pair_enabled() and not ops.planes_enabled()
            assert not ops.conv3_ok(8, 128, 128, 256, 256, 3, 3, 1, 1) and not ops.conv3_f16_bwd_enabled()
        finally:
            ops.set_amp_fast(True)
    assert not ops.amp_enabled() and ops.pair_enabled()


def test_reducer_refuses_a_subgroup_without_a_syncbn_group():
    """dist.new_group is a collective over the DEFAULT group: building the own SyncBatchNorm communicator from a sub-group would hang
     -- the constructor raises instead"""
    from vbg.optim import FlatReducer, FusedSGD, split_parameters
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        cnn, _ = split_parameters(_Toy())
        opt = FusedSGD(cnn, torch.device("cpu"), lr=0.1)

        class _Two:                      # a stand-in "group" of world size 2 so that the reducer is enabled
            pass
        real = dist.get_world_size
        dist.get_world_size = lambda group=None: 2
        try:
            with pytest.raises(ValueError):
                FlatReducer([opt], group=_Two(), sync_bn_group="new")
        finally:
            dist.get_world_size = real
    finally:
        dist.destroy_process_group()


def test_arm_and_drop_untouched_follow_torch_semantics():
    """ADVICE r5 (vbg/optim.py): FlatGroup.arm() -- the start of a backward, after the loop's `optimizer.zero_grad()` dropped the gradients --
    (i) zeroes a contiguous RUN of missing parameters with one memset instead of one launch per parameter, (ii) leaves a gradient the
    caller kept in place (accumulated into, as torch would), (iii) takes a `.grad` that is somebody else's tensor (what DDP's
    finalize_backward leaves on a locally-unused parameter) into the flat view; ModelHome.drop_untouched() only hands `None` back for
    gradients that arm() itself attached in this backward -- one that existed when the backward started survives (gradient accumulation,
    DDP no_sync micro-steps)."""
    from vbg.optim import FlatGroup, ModelHome
    net = _Toy()
    named = [(n, p) for n, p in net.named_parameters() if "bert_model" not in n and "fc" not in n]
    g = FlatGroup(named, torch.device("cpu"))
    n = len(g.params)
    assert n == 4
    # all dropped -> one memset of the whole buffer, every view back
    g.gflat.fill_(7.0)
    for p in g.params:
        p.grad = None
    g.arm()
    assert float(g.gflat.abs().max()) == 0.0 and all(p.grad is gv for p, gv in zip(g.params, g.gviews)) and g._armed == list(range(n))
    # one gradient kept, the others dropped: the kept one is accumulated into (not zeroed), the run of missing ones is zeroed
    g.gflat.fill_(3.0)
    for i, p in enumerate(g.params):
        p.grad = g.gviews[i] if i == 1 else None
    g.arm()
    assert float(g.gviews[1].min()) == 3.0 and g._armed == [0, 2, 3]
    assert all(float(g.gviews[i].abs().max()) == 0.0 for i in (0, 2, 3))
    # a foreign .grad tensor: its contents move into the flat view, which becomes .grad again
    foreign = torch.full_like(g.params[2].data, 5.0)
    g.params[2].grad = foreign
    g.arm()
    assert g.params[2].grad is g.gviews[2] and float(g.gviews[2].min()) == 5.0
    # drop_untouched: only what arm() attached in THIS backward and nobody touched
    home = ModelHome.__new__(ModelHome)
    home.groups, home.touched = [g], set()
    for i, p in enumerate(g.params):
        p.grad = g.gviews[i] if i in (0, 1) else None          # 0, 1 existed before this backward; 2, 3 were dropped by zero_grad()
    g.arm()
    assert g._armed == [2, 3]
    home.touched.add(id(g.params[2]))                          # parameter 2 took part in the backward, 3 did not
    home.drop_untouched()
    assert g.params[0].grad is g.gviews[0] and g.params[1].grad is g.gviews[1] and g.params[2].grad is g.gviews[2] and g.params[3].grad is None


def test_rehoming_under_a_live_optimizer_raises():
    """`_home()` used to rebuild the flat groups silently when `valid()` failed (a `.to()` / `.half()` / fresh `p.data` after an
    optimizer existed) -- the optimizer then went on stepping buffers the model no longer read.  A group a live optimizer / reducer owns is
    never replaced: the model raises."""
    from model.ViBERTgrid_net import ViBERTgridNet
    from vbg.optim import FlatGroup, FusedSGD, ModelHome, split_parameters
    net = _Toy()
    cnn, _ = split_parameters(net)
    opt = FusedSGD(cnn, torch.device("cpu"), lr=0.1)
    home = ModelHome.__new__(ModelHome)
    home.groups, home.touched = [opt.group], None
    holder = type("M", (), {})()
    holder.__dict__["_vbg_home_state"] = home
    holder.classifier_mode = "simp"
    assert ViBERTgridNet._home(holder) is home                 # still valid: kept
    cnn[0][1].data = cnn[0][1].data.clone()                    # the parameter left its flat storage
    assert not home.valid()
    with pytest.raises(RuntimeError, match="flat storage"):
        ViBERTgridNet._home(holder)
    del opt                                                    # nobody steps the old buffers any more: re-homing is fine (not exercised here)
