"""debug: the seg-head bias gradients of a full-scale case against the fixture (values, not just the relative error)"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("vibertgrid-pytorch_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch, tempfile
import test_gpu_full_scale as T
import full_scale as F
name = sys.argv[1] if len(sys.argv) > 1 else "cfg5e"
golden = lambda f: np.load(os.path.join(ROOT, "tests", "golden", f), allow_pickle=True)
g, c, net, dbatch = T._setup(golden, tempfile.mkdtemp(), name)
net.train()
if c.get("bn_frozen"):
    for m in net.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.eval()
random.seed(7)
tl = net(*dbatch)
tl.backward()
named = dict(net.named_parameters())
for k in ("semantic_segmentation_head.semantic_segmentation_encoder.conv_3_1.bias", "semantic_segmentation_head.semantic_segmentation_encoder.conv_3_2.bias",
          "backbone.conv_4_x.1.conv_1.weight" if "backbone.conv_4_x.1.conv_1.weight" in named else "semantic_segmentation_head.semantic_segmentation_encoder.conv_3_1.weight"):
    a = named[k].grad.detach().cpu().double().reshape(-1)
    b = torch.from_numpy(g["grad::" + k]).double().reshape(-1)
    if a.numel() <= 16:
        print(k, "\n  mine", a.tolist(), "\n  ref ", b.tolist(), "\n  sum mine %.3e sum ref %.3e" % (float(a.sum()), float(b.sum())))
    else:
        s = F.sample(named[k].grad, 4096).cpu().double()
        print(k, "rel-L2 %.3e  norm ref %.3e  max|ref| %.3e" % (float((s - b).norm() / b.norm()), float(b.norm()), float(b.abs().max())))
