"""world_size-2 `gloo` test (CPU) of the data-parallel path: flat gradient buckets, readiness hooks, async
all-reduce, 1/world folded into the optimizer scale, unused parameters kept out of the buffers."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.bert_model = torch.nn.Linear(6, 5)               # name contains "bert_model" -> AdamW group
        self.head = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
        self.conv = torch.nn.Conv2d(4, 4, 3, padding=1).to(memory_format=torch.channels_last)
        self.pooler = torch.nn.Linear(3, 3)                   # never used: must stay out of the flat buffers

    def forward(self, x, img, skip_conv=False):
        y = self.head(self.bert_model(x)).sum()
        return y if skip_conv else y + self.conv(img).square().mean()


STEPS = 3


def _worker(rank, world, port, out, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vbg import functions as Fn
    from vbg.optim import FlatGroup, FlatReducer, split_parameters

    class Opt:          # the reducer only needs .group and .grad_scale (the fused steps themselves are GPU kernels)
        def __init__(self, named):
            self.group = FlatGroup(named, torch.device("cpu"))
            self.grad_scale = 1.0

    net = _Toy()
    cnn, bert = split_parameters(net)
    assert all("pooler" not in n for n, _ in cnn + bert)
    assert [n for n, _ in bert] == ["bert_model.weight", "bert_model.bias"]
    opts = [Opt(cnn), Opt(bert)]
    # tiny buckets -> several async all-reduces in flight
    if mode in ("shared", "violated"):   # SyncBatchNorm statistics on the reducer's communicator, buckets from inside backward: the caller's
        red = FlatReducer(opts, bucket_mb=1e-4, static_graph=True)      # word that the graph is rank-invariant is what allows the overlap
        assert Fn.SyncCtx.group is None and red.sync_bn_mode == "shared communicator" and red.overlap
    elif mode == "default":          # THE DEFAULT: shared communicator, nothing asserted about the graph -> buckets after backward
        red = FlatReducer(opts, bucket_mb=1e-4)
        assert red.sync_bn_mode == "shared communicator" and not red.overlap
    elif mode == "own":              # opt-in: an own communicator for the statistics (tolerates rank-varying graphs with overlap)
        red = FlatReducer(opts, bucket_mb=1e-4, sync_bn_group="new", serialize_syncbn=True)
        assert dist.get_world_size(Fn.SyncCtx.group) == 2 and Fn.SyncCtx.group is not None and Fn.SyncCtx.before is not None
    else:                            # every bucket from finish(): rank-invariant for any graph
        red = FlatReducer(opts, bucket_mb=1e-4, overlap=False)
        assert not red.overlap
    assert red.enabled and len(red.buckets) >= 3 and all(o.grad_scale == 0.5 for o in opts)
    # channels_last conv weight keeps its physical layout inside the flat buffer
    assert net.conv.weight.is_contiguous(memory_format=torch.channels_last)
    for step in range(STEPS):
        g = torch.Generator().manual_seed(100 + rank + 10 * step)
        x, img = torch.randn(4, 6, generator=g), torch.randn(2, 4, 5, 5, generator=g)
        for o in opts:
            o.group.zero_grad()
        # data-dependent graph (classifier_mode full / crf): in the last step rank 1 skips a sub-module, so its conv bucket never
        # completes during backward -- the fixed launch sequence must keep both ranks pairing the same buffers
        net(x, img, skip_conv=(step == STEPS - 1 and rank == 1)).backward()
        if step > 0:
            assert red.order is not None and sorted(red.order) == list(range(len(red.buckets)))
            if step == 1:            # (in the last step rank 1's sequence may be held up by the bucket it skipped)
                assert (len(red.handles) > 0) == (mode not in ("late", "default"))      # buckets left during backward unless overlap is off
            if mode == "own" and step == 1:
                # a statistics collective behind buckets in flight waits for them.  Only in a step where both ranks run the same graph:
                # serialising the two communicators against each other is a deadlock once the ranks' sequences interleave differently
                # (rank 0 waits for a bucket that rank 1 only issues from finish(), behind the statistics collective rank 0 has not
                # reached) -- FlatReducer.__init__ says so, and it is why serialisation is not the default
                Fn.SyncCtx.all_reduce(torch.ones(2))
            if mode == "violated" and step == 1:
                Fn.SyncCtx.all_reduce(torch.ones(2))          # (a SyncBatchNorm statistics collective on the shared communicator)
            assert "buckets issued" in red.describe_pending() and f"rank {rank}" in red.describe_pending()
        if mode == "violated" and step == STEPS - 1 and rank == 1:
            # static_graph=True was asserted and this rank's graph skipped a sub-module: the late flush is reported, not passed over
            with pytest.raises(RuntimeError, match="static_graph=True"):
                red.finish()
            red.handles = []
            continue
        red.finish()
        assert red.steps_done == step + 1 and not red.handles
    orders = [None, None]
    dist.all_gather_object(orders, red.order)
    assert orders[0] == orders[1]
    res = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["shared", "default", "own", "late", "violated"])
def test_flat_reducer_two_ranks(tmp_path, mode):
    port, out = _free_port(), str(tmp_path / "g.pt")
    mp.spawn(_worker, args=(2, port, out, mode), nprocs=2, join=True)
    got = torch.load(out)
    # expected: SUM over the two ranks of the last step's local gradients (averaging happens in the optimizer kernel)
    net = _Toy()
    exp = None
    for rank in range(2):
        g = torch.Generator().manual_seed(100 + rank + 10 * (STEPS - 1))
        x, img = torch.randn(4, 6, generator=g), torch.randn(2, 4, 5, 5, generator=g)
        net.zero_grad()
        net(x, img, skip_conv=(rank == 1)).backward()
        cur = {n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in net.named_parameters() if "pooler" not in n}
        exp = cur if exp is None else {k: exp[k] + cur[k] for k in exp}
    assert set(got) == set(exp)
    for k in exp:
        assert torch.allclose(got[k], exp[k], rtol=1e-5, atol=1e-6), k
