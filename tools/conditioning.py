"""Is an e2e gradient mismatch a kernel bug or fp32 conditioning of the tiny fixture?  Run the SAME product step with
different GEMM tile configurations (different fp32 summation orders, identical math) and compare the gradients."""
import os, sys, random, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("vibertgrid-pytorch_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import vbg_oracle as O
from vbg import ops
from test_gpu_model import build_product, load_synth, to_dev
from test_oracle_golden import _e2e_inputs, e2e_cfg

g = np.load(os.path.join(ROOT, "tests/golden/e2e.npz"))
dev = torch.device("cuda")
for tag, bb in (("r18", "resnet_18_fpn"), ("r34p", "resnet_34_fpn_pretrained")):
    cfg = e2e_cfg(bb)
    grads = {}
    for name, force in (("64/32", (64064, 32)),):
        ops._FORCE[0], ops._FORCE[1] = force
        net = build_product(tempfile.mkdtemp(), bb, cfg)
        sd = load_synth(net, cfg, 1200)
        net = net.to(dev).train()
        random.seed(7)
        l = net(*to_dev(_e2e_inputs(g), dev))
        l.backward()
        grads[name] = {n: p.grad.detach().double().cpu() for n, p in net.named_parameters() if p.grad is not None}
        print(tag, name, "loss", float(l))
    ops._FORCE[0] = ops._FORCE[1] = 0
    O.OHEM_STABLE_SORT = True
    sdg = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
    random.seed(7)
    ol = O.forward(sdg, cfg, *_e2e_inputs(g), training=True)[0]
    ol.backward()
    og = {k: v.grad.double() for k, v in sdg.items() if getattr(v, "grad", None) is not None}
    def worst(a, b):
        w = sorted(((float((a[k] - b[k]).norm() / (b[k].norm() + 1e-30)), k) for k in a if k in b and float(b[k].norm()) > 1e-6), reverse=True)
        return [(round(x, 4), k[-40:]) for x, k in w[:3]]
    a, b = grads["64/32"], og
    for k in a:
        if k in b and float(b[k].norm()) > 1e-6 and ("backbone" in k or "late" in k or "semantic" in k) and k.endswith(("weight",)) and a[k].dim() > 1:
            print(tag, f"{float((a[k]-b[k]).norm()/b[k].norm()):.5f}", k)
    for n in grads:
        print(tag, n, "vs oracle:", worst(grads[n], og))
