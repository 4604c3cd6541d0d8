"""vbg.batch.PackedBatch: the six forward arguments through one (pinned) buffer and one copy."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _batch():
    g = torch.Generator().manual_seed(3)
    imgs = (torch.rand(3, 40, 56, generator=g), torch.rand(3, 33, 47, generator=g))
    segs = (torch.arange(5, dtype=torch.int32).repeat_interleave(3), torch.arange(4, dtype=torch.int32))
    classes = (torch.randint(0, 5, (5,), generator=g).int(), torch.randint(0, 5, (4,), generator=g).int())
    coors = (torch.randint(0, 30, (5, 4), generator=g), torch.randint(0, 30, (4, 4), generator=g))
    corpus = torch.zeros(2, 15, dtype=torch.long)
    corpus[0] = torch.randint(1000, 1200, (15,), generator=g)
    corpus[1, :4] = torch.randint(1000, 1200, (4,), generator=g)
    return imgs, segs, classes, coors, corpus, (corpus != 0).int()


def test_pack_roundtrip_cpu():
    from vbg.batch import PackedBatch, host_mirror, packed_collate
    b = _batch()
    pk = PackedBatch.pack(*b)
    out = pk.to("cpu")
    for got, ref in zip(out[:4], b[:4]):
        assert len(got) == len(ref)
        for x, y in zip(got, ref):
            assert x.dtype == y.dtype and x.shape == y.shape and torch.equal(x, y)
    assert torch.equal(out[4], b[4]) and out[4].dtype == torch.int64
    assert torch.equal(out[5], b[5]) and out[5].dtype == torch.int32
    # host mirrors of the pieces the model indexes on the host
    assert np.array_equal(host_mirror(out[4]), b[4].numpy()) and np.array_equal(host_mirror(out[5]), b[5].numpy())
    assert all(np.array_equal(host_mirror(x), y.numpy()) for x, y in zip(out[1], b[1]))
    assert host_mirror(out[0][0]) is None
    # every slot starts on a 16-byte boundary (float4 kernels read the images in place)
    assert all(o % 16 == 0 for _, o, _, _ in pk.table)
    # the collate wrapper keeps whatever follows the six model arguments (eval mode: texts, key dicts)
    pk2 = packed_collate(lambda samples: b + (("a", "b"), ({}, {})))(None)
    assert pk2.extras == (("a", "b"), ({}, {})) and torch.equal(pk2.to("cpu")[4], b[4])


def test_pack_empty_document():
    from vbg.batch import PackedBatch
    b = list(_batch())
    b[1] = (b[1][0], torch.zeros(0, dtype=torch.int32))
    b[2] = (b[2][0], torch.zeros(0, dtype=torch.int32))
    b[3] = (b[3][0], torch.zeros(0, 4, dtype=torch.long))
    out = PackedBatch.pack(*b).to("cpu")
    assert out[1][1].numel() == 0 and out[3][1].shape == (0, 4)


@pytest.mark.gpu
def test_packed_forward_equals_plain(golden, tmp_path):
    import random
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_model import build_product, load_synth, to_dev
    from test_oracle_golden import _e2e_inputs, e2e_cfg
    from vbg.batch import PackedBatch
    g = golden("e2e.npz")
    cfg = e2e_cfg("resnet_18_fpn")
    dev = torch.device("cuda")
    net = build_product(tmp_path, "resnet_18_fpn", cfg)
    load_synth(net, cfg, 1200)
    net = net.to(dev).eval()
    batch = _e2e_inputs(g)
    random.seed(7)
    with torch.no_grad():
        ref = net(*to_dev(batch, dev))
    pk = PackedBatch.pack(*batch)
    assert pk.buf.is_pinned()
    random.seed(7)
    with torch.no_grad():
        got = net(*pk.to(dev))
    assert all(torch.equal(a, b) for a, b in zip(ref, got))
