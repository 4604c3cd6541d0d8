import sys, random, pathlib, tempfile
sys.path[:0] = ["/root/repo", "/root/repo/oracle", "/root/repo/vibertgrid-pytorch_amd", "/root/repo/tests"]
import numpy as np, torch
from test_gpu_model import build_product, load_synth, to_dev
from test_oracle_golden import _e2e_inputs, e2e_cfg
from vbg.optim import FusedAdamW, FusedSGD, split_parameters, clip_grad_norm_
g = np.load("/root/repo/tests/golden/e2e.npz")
cfg = e2e_cfg("resnet_18_fpn")
dev = torch.device("cuda")
dbatch = to_dev(_e2e_inputs(g), dev)
def run(mode):
    net = build_product(pathlib.Path(tempfile.mkdtemp()), "resnet_18_fpn", cfg)
    load_synth(net, cfg, 1200)
    net = net.to(dev).train()
    if mode == "torch":
        pc = [p for n, p in net.named_parameters() if "bert_model" not in n]
        pb = [p for n, p in net.named_parameters() if "bert_model" in n]
        oc = torch.optim.SGD(pc, lr=0.005, momentum=0.9, weight_decay=0.005)
        ob = torch.optim.AdamW(pb, lr=5e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    else:
        cnn, bert = split_parameters(net)
        oc = FusedSGD(cnn, dev, lr=0.005, momentum=0.9, weight_decay=0.005)
        ob = FusedAdamW(bert, dev, lr=5e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    snaps = []
    for step in range(2):
        random.seed(11 + step)
        loss = net(*dbatch)
        oc.zero_grad(); ob.zero_grad()
        loss.backward()
        if mode == "torch":
            tn = float(torch.nn.utils.clip_grad_norm_(net.parameters(), max_norm=2))
        else:
            tn = clip_grad_norm_([oc, ob], 2.0)
        gs = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
        oc.step(); ob.step()
        torch.cuda.synchronize()
        snaps.append(({n: p.detach().clone() for n, p in net.named_parameters()}, gs, float(loss), tn))
    return snaps
def cmp(a, b, tag):
    for step in range(2):
        for what in (0, 1):
            A, B = a[step][what], b[step][what]
            worst = sorted(((float((A[k] - B[k]).norm() / (A[k].norm() + 1e-20)), k) for k in A if k in B and "pooler" not in k and "key.bias" not in k), reverse=True)[:3]
            print(tag, "step", step, "params" if what == 0 else "grads ", a[step][2], b[step][2], a[step][3], b[step][3], [(f"{d:.1e}", k[-40:]) for d, k in worst])
runs = {m: [run(m) for _ in range(3)] for m in ("torch", "fused")}
cmp(runs["torch"][0], runs["torch"][1], "T-T")
cmp(runs["torch"][0], runs["torch"][2], "T-T")
cmp(runs["fused"][0], runs["fused"][1], "F-F")
cmp(runs["fused"][0], runs["fused"][2], "F-F")
cmp(runs["torch"][0], runs["fused"][0], "T-F")
