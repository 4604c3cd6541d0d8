#!/usr/bin/env python3
"""Diagnostic for tests/test_gpu_ddp.py::test_stock_ddp_wrapper_equals_flat_reducer: the reference's DDP + torch.optim wiring around the
drop-in model on two ranks sharing one GPU (gloo) with learning rate 0 -- the parameters never move, so the gradients of steps 2 and 3
must equal those of step 1 up to atomics noise.  Prints, per step, the parameters whose gradient moved most against step 1.
    python tools/ddp_stock_probe.py            (VBG_HOME=0 / 1)"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "vibertgrid-pytorch_amd")):
    sys.path.insert(0, p)
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import test_gpu_ddp as T
    from test_gpu_model import to_dev
    from vbg import ops
    v = "conv3"
    T._dispatch(ops, v)
    dbatch = to_dev(T._slice(T._docs(), rank, rank + 1), dev)
    model = T._build(os.path.join(tmp, f"probe{rank}"), sync_bn=False, v=v)
    model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model).to(dev)
    model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], find_unused_parameters=True)
    pc = [p for n, p in model.named_parameters() if "bert_model" not in n and p.requires_grad]
    pb = [p for n, p in model.named_parameters() if "bert_model" in n and p.requires_grad]
    live = os.environ.get("VBG_PROBE_LIVE", "0") != "0"          # the test's hyper-parameters instead of learning rate 0
    oc = torch.optim.SGD(params=pc, lr=0.005 if live else 0.0, momentum=0.9, weight_decay=0.005 if live else 0.0)
    ob = torch.optim.AdamW(params=pb, lr=5e-5 if live else 0.0, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01 if live else 0.0)
    model.train()
    if not live:
        for m in model.modules():          # frozen running statistics would still move; keep BatchNorm in train mode but momentum 0
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.momentum = 0.0
    trace = []
    first = None
    for step in range(3):
        random.seed(5)
        loss = model(*dbatch)
        lv = loss.item()
        oc.zero_grad()
        ob.zero_grad()
        loss.backward()
        torch.cuda.synchronize()
        g = {n: p.grad.detach().clone() for n, p in model.module.named_parameters() if p.grad is not None}
        if first is None:
            first = g
            if rank == 0 and os.environ.get("VBG_PROBE_DUMP"):
                torch.save({k: v.cpu() for k, v in g.items()}, os.environ["VBG_PROBE_DUMP"])
        else:
            d = sorted(((float((g[k] - first[k]).norm() / (first[k].norm() + 1e-30)), k) for k in first if "key.bias" not in k), reverse=True)
            if rank == 0:
                print(f"step {step + 1} loss {lv:.7f}: gradients vs step 1, worst:", [(f"{a:.2e}", k) for a, k in d[:6]], "median", f"{d[len(d) // 2][0]:.2e}", flush=True)
        oc.step()
        ob.step()
        if live and rank == 0:
            torch.cuda.synchronize()
            trace.append(({k: v.cpu() for k, v in g.items()}, {n: p.detach().cpu().clone() for n, p in model.module.named_parameters()}, lv))
    if live and rank == 0 and os.environ.get("VBG_PROBE_DUMP"):
        torch.save(trace, os.environ["VBG_PROBE_DUMP"])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import socket
    import tempfile
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(worker, args=(2, port, tempfile.mkdtemp(prefix="vbg_probe_")), nprocs=2, join=True)
