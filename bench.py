#!/usr/bin/env python3
"""Training-throughput bench of the MI355X-native ViBERTgrid step (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = the body of the reference's train loop (pipeline/train_val_utils.py:248-287) on one synthetic
batch that is already resident in HBM: forward (ViBERTgridNet, train mode, dropout on), loss .item(), zero_grad,
backward (+ bucketed RCCL all-reduce when N > 1), conditional grad-norm clip, fused SGD + AdamW steps, and for N > 1 the
reference's per-step torch.distributed.barrier() (pipeline/train_val_utils.py:286-287; --no-step-barrier leaves it out).
Workload = BASELINE.json configs[1]: SROIE line-level, resnet_34_fpn_pretrained + bert-base-uncased (12 layers,
vocab 30522, random init: no network for checkpoints), 512x512, seq_len 512, 128 segments, batch 8 per GPU.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import random
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: what RCCL needs on this driver for N > 1

import torch
import torch.distributed as dist

NCLS, VOCAB = 5, 30522
F_STEP_GF = 690.1          # algorithmic GFLOP per document of one training step at cfg2 (PAD excluded)
PEAK_F32_TF = 157.3        # MI355X fp32 MFMA peak (MI355X_MICROARCH.md)
PEAK_BF16_TF = 2500.0      # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
SPLIT_PRODUCTS = 6         # bf16 piece products per fp32-grade product of the split form (csrc/gemm.hip, PREC 3); the fp16-pair form: 3


def make_bert_dir(top, layers=12, vocab=VOCAB, dropout=0.1):
    from transformers import BertConfig
    d = os.path.join(top, "bert-base-uncased")
    os.makedirs(d, exist_ok=True)
    BertConfig(vocab_size=vocab, num_hidden_layers=layers, hidden_dropout_prob=dropout, attention_probs_dropout_prob=dropout).save_pretrained(d)
    toks = ["[PAD]"] + [f"[unused{i}]" for i in range(1, 100)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    toks += [f"tok{i}" for i in range(len(toks), vocab)]
    with open(os.path.join(d, "vocab.txt"), "w") as f:
        f.write("\n".join(toks) + "\n")
    return d


def make_roberta_dir(top, layers=12, vocab=50265, dropout=0.1):
    """roberta-base dimensions (vocab 50265, 514 positions, one token type, eps 1e-5), random init, with a synthetic tokenizer"""
    from transformers import RobertaConfig
    d = os.path.join(top, "roberta-base")
    os.makedirs(d, exist_ok=True)
    RobertaConfig(vocab_size=vocab, max_position_embeddings=514, type_vocab_size=1, num_hidden_layers=layers, hidden_dropout_prob=dropout,
                  attention_probs_dropout_prob=dropout, layer_norm_eps=1e-5).save_pretrained(d)
    voc = {"<s>": 0, "<pad>": 1, "</s>": 2, "<unk>": 3}
    voc.update({f"t{i}": i for i in range(4, vocab)})
    json.dump(voc, open(os.path.join(d, "vocab.json"), "w"))
    open(os.path.join(d, "merges.txt"), "w").write("#version: 0.2\n")
    return d


def synthetic_batch(B, H, W, T, S, ncls, vocab, seed):
    """SURVEY.md §8(d) synthetic documents (fixed across steps: data loading is off the clock)."""
    g = torch.Generator().manual_seed(seed)
    imgs = tuple(torch.rand(3, H, W, generator=g) for _ in range(B))
    coors, segs, classes = [], [], []
    for _ in range(B):
        x1 = torch.randint(0, W - 72, (S,), generator=g)
        y1 = torch.randint(0, H - 24, (S,), generator=g)
        w = torch.randint(8, 73, (S,), generator=g)
        h = torch.randint(8, 25, (S,), generator=g)
        coors.append(torch.stack([x1, y1, x1 + w, y1 + h], 1).long())
        segs.append(torch.arange(S, dtype=torch.int32).repeat_interleave(T // S))
        classes.append(torch.randint(0, ncls, (S,), generator=g).int())
    corpus = torch.randint(1000, vocab, (B, T), generator=g)
    mask = torch.ones(B, T, dtype=torch.int32)
    return imgs, tuple(segs), tuple(classes), tuple(coors), corpus, mask


def build_model(tmp, backbone="resnet_34_fpn_pretrained", layers=12, vocab=VOCAB, dropout=0.1, img=512, ncls=NCLS, roberta=False):
    import warnings
    from transformers import BertTokenizer
    from model.ViBERTgrid_net import ViBERTgridNet
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if roberta:
                from transformers import RobertaTokenizer
                d = make_roberta_dir(tmp, layers, vocab, dropout)
                name, tok = "roberta-base", RobertaTokenizer(os.path.join(d, "vocab.json"), os.path.join(d, "merges.txt"))
            else:
                d = make_bert_dir(tmp, layers, vocab, dropout)
                name, tok = "bert-base-uncased", BertTokenizer(os.path.join(d, "vocab.txt"))
            # work_mode="eval" builds BERT from its config (no checkpoint download); .train() flips work_mode to "train"
            net = ViBERTgridNet(num_classes=ncls, image_mean=[0.9248, 0.9224, 0.9215], image_std=[0.1532, 0.1545, 0.1536],
                                image_min_size=[img], image_max_size=img, test_image_min_size=img, bert_model=name,
                                tokenizer=tok, backbone=backbone, grid_mode="mean",
                                loss_weights=None, num_hard_positive_main_1=16, num_hard_negative_main_1=16,
                                num_hard_positive_main_2=32, num_hard_negative_main_2=32, loss_aux_sample_list=[256, 512, 256],
                                num_hard_positive_aux=256, num_hard_negative_aux=256, loss_control_lambda=1, add_pos_neg=True,
                                classifier_mode="simp", ohem_random=True, layer_mode="single", work_mode="eval")
    finally:
        os.chdir(cwd)
    return net


def cpu_baseline(threads, shape="cfg2"):
    """The CPU oracle (oracle/vbg_oracle.py: the pinned restatement of the reference's step) timed on this box's
    host cores on a bounded sample: batches of 2 cfg2-shaped documents, 1 warm-up step + 3 timed steps of forward + backward +
    SGD / AdamW updates.  `cores` = the threads torch was given (what the number was measured on), `host_cores` = the box's logical
    cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import vbg_oracle as O
    torch.set_num_threads(threads)
    if shape == "cfg1":          # BASELINE configs[0]: resnet_18_fpn + bert-base-uncased, ONE 256 x 256 document, T = 128, S = 32
        cfg = O.NetCfg(num_classes=NCLS, backbone="resnet_18_fpn", image_min_size=(256,), image_max_size=256, test_image_min_size=256,
                       bert=O.BertCfg(layers=12, dropout=0.1))
        nd, warm, steps, img, T_, S_ = 1, 1, 5, 256, 128, 32
    else:
        cfg = O.NetCfg(num_classes=NCLS, backbone="resnet_34_fpn_pretrained", bert=O.BertCfg(layers=12, dropout=0.1))
        nd, warm, steps, img, T_, S_ = 2, 1, 3, 512, 512, 128
    sd = O.synth_state_dict(O.state_shapes(cfg, vocab=VOCAB, dup_bert=False))
    sd = {k: (v.requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd.items()}
    batch = synthetic_batch(nd, img, img, T_, S_, NCLS, VOCAB, 4321)
    state = {}
    times = []
    for it in range(warm + steps):
        random.seed(it)
        t0 = time.time()
        loss = O.forward(sd, cfg, *batch, training=True)[0]
        loss.backward()
        with torch.no_grad():
            for k, v in sd.items():
                if v.grad is None:
                    continue
                if "bert_model" in k:
                    m, vv = state.get(k, (torch.zeros_like(v), torch.zeros_like(v)))
                    p, m, vv = O.adamw_step(v, v.grad, m, vv, it + 1, 5e-5, 0.9, 0.999, 1e-8, 0.01)
                    state[k] = (m, vv)
                else:
                    p, m = O.sgd_step(v, v.grad, state.get(k), 0.005, 0.9, 0.005)
                    state[k] = m
                v.copy_(p)
                v.grad = None
        times.append(time.time() - t0)
    dt = sum(times[warm:])
    return {"value": round(nd * steps / dt, 5), "unit": "docs/sec", "cores": threads, "host_cores": os.cpu_count() or 1, "threads": threads, "kind": "port",
            "sample": f"{steps} timed steps (after {warm} warm-up) of {nd} document(s) each ({shape} shape: {img}x{img}, T={T_}, S={S_}, "
                      f"{'r18' if shape == 'cfg1' else 'r34'}+bert-base 12L), fwd+bwd+SGD/AdamW, {dt:.1f} s, torch CPU fp32"}


def stock_loop_leg(make_net, batch, dev, world, steps, warm, amp=False, resident=False, sync_bn=False, barrier=True):
    """The reference's training loop around the drop-in model, line for line (pipeline/train_val_utils.py:248-287 with the wiring of
    train_SROIE.py:202-235): per-tensor pageable `.to(device)` of the six collate outputs, `torch.cuda.amp.autocast`, `train_loss.item()`
    between forward and backward, `optimizer.zero_grad()` (set_to_none), `train_loss > loss_clip_tresh` on the device tensor,
    torch.optim.SGD + torch.optim.AdamW split by "bert_model" in name, GradScaler when amp, DistributedDataParallel(find_unused_parameters=
    True) + convert_sync_batchnorm when distributed.  Nothing of vbg.optim / vbg.batch is constructed by this function: what the model
    needs (flat parameter storage, plane images) it sets up itself at its first forward.  resident=True leaves the H2D copies out of
    the loop (the same loop on a batch uploaded once) -- the decomposition of the difference to the headline, not the leg's value."""
    model = make_net()
    if sync_bn:
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
    model = model.to(dev)
    distributed = world > 1
    if distributed:
        model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev.index], find_unused_parameters=True)
    model.train()
    params_cnn, params_bert = [], []
    for name, parameters in model.named_parameters():
        if "bert_model" in name and parameters.requires_grad:
            params_bert.append(parameters)
        elif parameters.requires_grad:
            params_cnn.append(parameters)
    optimizer_cnn = torch.optim.SGD(params=params_cnn, lr=0.005, momentum=0.9, weight_decay=0.005)
    optimizer_bert = torch.optim.AdamW(params=params_bert, lr=5e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    scaler = torch.amp.GradScaler("cuda") if amp else None
    loss_clip_tresh, clip_norm = 10, 2
    device = dev
    if resident:
        on_dev = tuple(tuple(t.to(device) for t in g) if isinstance(g, tuple) else g.to(device) for g in batch)

    def step():
        image_list, seg_indices, token_classes, ocr_coors, ocr_corpus, mask = on_dev if resident else batch
        if not resident:
            image_list = tuple(image.to(device) for image in image_list)
            seg_indices = tuple(seg_index.to(device) for seg_index in seg_indices)
            token_classes = tuple(token_class.to(device) for token_class in token_classes)
            ocr_coors = tuple(ocr_coor.to(device) for ocr_coor in ocr_coors)
            ocr_corpus = ocr_corpus.to(device)
            mask = mask.to(device)
        with torch.autocast("cuda", dtype=torch.float16, enabled=scaler is not None):
            train_loss = model(image_list, seg_indices, token_classes, ocr_coors, ocr_corpus, mask)
        train_loss_value = train_loss.item()
        optimizer_cnn.zero_grad()
        optimizer_bert.zero_grad()
        if scaler is not None:
            scaler.scale(train_loss).backward()
            scaler.step(optimizer_cnn)
            scaler.step(optimizer_bert)
            scaler.update()
        else:
            train_loss.backward()
            if train_loss > loss_clip_tresh:
                torch.nn.utils.clip_grad_norm(model.parameters(), max_norm=clip_norm)
            optimizer_cnn.step()
            optimizer_bert.step()
        if distributed and barrier:
            dist.barrier()
        return train_loss_value

    for _ in range(warm):
        step()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    del model, optimizer_cnn, optimizer_bert
    torch.cuda.empty_cache()
    return dt, last


def _free_port():
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def self_launch(n, argv):
    """`python bench.py --gpus N` (N > 1) without a launcher around it -- the form the driver's BENCH command has: re-exec this file under
    `python -m torch.distributed.run --nnodes 1 --nproc-per-node N` on a free loopback port, the way the reference is started
    (readme.md:87 `torchrun --nnodes 1 --nproc_per_node N`, pipeline/distributed_utils.py:74-77 reads RANK / WORLD_SIZE / LOCAL_RANK).
    Rank 0's JSON line is the only thing the ranks write to stdout, and they inherit this process's stdout: nothing to relay.
    `--syncbn-comm direct` (the opt-in second communicator, vbg/rccl.py) is first PROBED: three dry steps in a run of their own under a
    20 s stall watchdog; if the probe stalls or fails the timed run falls back to `--syncbn-comm shared` with a warning -- a deadlock
    between two communicators cannot be undone from inside the process that is stuck in it, a parent can.  Returns the exit code."""
    import signal
    import subprocess

    def run(extra, limit, quiet=False):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv + extra
        print("[bench] launching: " + " ".join(cmd), file=sys.stderr, flush=True)
        p = subprocess.Popen(cmd, start_new_session=True, env=dict(os.environ, VBG_SELF_LAUNCHED="1"), stdout=2 if quiet else None)   # (a probe's stdout goes to stderr)
        try:
            return p.wait(timeout=limit)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(p.pid, signal.SIGKILL)          # (the session this function started: launcher + ranks, nothing else)
            except ProcessLookupError:
                pass
            p.wait()
            return -9

    direct = any(a == "direct" for i, a in enumerate(argv) if i and argv[i - 1] == "--syncbn-comm") or "--syncbn-comm=direct" in argv
    if direct and ("--launch-check" not in argv or os.environ.get("VBG_PROBE_IN_CHECK") == "1"):
        rc = run(["--comm-probe"], float(os.environ.get("VBG_PROBE_LIMIT", "420")), quiet=True)
        if rc != 0:
            print(f"[bench] WARNING: the two-communicator probe (--syncbn-comm direct) did not pass (exit code {rc}); "
                  "falling back to --syncbn-comm shared (one communicator, one order of collectives)", file=sys.stderr, flush=True)
            argv = [a for a in argv if a != "--syncbn-comm=direct"]
            argv = [a for i, a in enumerate(argv) if not (a == "--syncbn-comm" or (i and argv[i - 1] == "--syncbn-comm"))] + ["--syncbn-comm", "shared", "--comm-fallback"]
    return run([], None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="documents per GPU (default: the configuration's own: 8, cfg5: 16)")
    ap.add_argument("--no-step-barrier", action="store_true", help="N > 1: leave out the reference loop's per-step torch.distributed.barrier()")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (default: all cores AND 16, both reported)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-syncbn", action="store_true")
    ap.add_argument("--no-single-stream-pass", action="store_true",
                    help="skip the extra pass that times the roofline families' launches with everything on one stream (profiled runs: the kernel trace then holds launches as they run only)")
    ap.add_argument("--syncbn-comm", default="shared", choices=["direct", "shared", "own"],
                    help="N > 1: SyncBatchNorm statistics on the gradient buckets' torch.distributed communicator (shared, the default: one "
                         "communicator, one order of collectives), as ncclAllReduce calls of a communicator of the library's own on the compute "
                         "stream (direct, vbg/rccl.py: opt-in, probed by the self-launcher before the timed run), or on a second torch.distributed "
                         "communicator (own) -- the A/Bs")
    ap.add_argument("--comm-probe", action="store_true", help=argparse.SUPPRESS)        # child of self_launch: three dry steps under a watchdog
    ap.add_argument("--comm-fallback", action="store_true", help=argparse.SUPPRESS)     # set by self_launch when the probe failed
    ap.add_argument("--launch-check", action="store_true", help=argparse.SUPPRESS)      # launcher plumbing only (CPU test): rendezvous + one all-reduce
    ap.add_argument("--no-ddp-overlap", action="store_true", help="N > 1: launch every gradient bucket after backward (A/B of the overlap)")
    ap.add_argument("--dist-timeout", type=int, default=int(os.environ.get("VBG_DIST_TIMEOUT", "240")),
                    help="N > 1: process-group timeout in seconds; a stalled step also prints which bucket / SyncBatchNorm collective every rank is at")
    ap.add_argument("--amp", action="store_true", help="time the `amp: True` path (one reduced-precision product per product) as the headline value instead "
                    "of fp32; the default run reports it beside the fp32 value under \"amp\"")
    ap.add_argument("--no-amp-leg", action="store_true", help="skip the secondary amp / fp32-MFMA measurements of the default run")
    ap.add_argument("--fp32-mfma", action="store_true", help="time the step with every product on the fp32 matrix pipe "
                    "(vbg.ops.set_precision('fp32')) as the headline instead of the fp32-grade split form")
    ap.add_argument("--shape", default="cfg2", choices=["cfg2", "cfg3", "cfg4", "cfg5"],
                    help="cfg2 (default, the BASELINE metric's configuration); cfg3 / cfg4 / cfg5: the other SURVEY §8 shapes as exploratory "
                         "runs (FUNSD 4 classes own-layout resnet / char-level S=T=512, 12 classes, vocab 21128 / 1024x1024 images with "
                         "roberta-base dimensions at 16 documents per GPU), reported under config.workload")
    ap.add_argument("--h2d", action="store_true", help="make the headline the PCIe-inclusive step (packed pinned H2D transfer of the batch "
                    "inside every step: SURVEY.md §8d's step body); the default run reports that rate beside the HBM-resident headline")
    ap.add_argument("--sync-loss", action="store_true", help="read the loss with a blocking .item() between forward and backward")
    ap.add_argument("--no-h2d-leg", action="store_true", help="skip the PCIe-inclusive leg of the default run")
    ap.add_argument("--no-stock-leg", action="store_true", help="skip the `stock_loop` leg (the reference's loop verbatim: torch.optim, .item(), "
                    "per-tensor pageable H2D, GradScaler under amp) of the default run")
    ap.add_argument("--stock", action="store_true", help="N > 1: run the stock_loop leg as well (DistributedDataParallel + SyncBatchNorm around the "
                    "drop-in model; by default only N = 1 runs it)")
    ap.add_argument("--cpu-baseline-only", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-baseline-shape", default="cfg2", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_only:          # child process of the default run: the CPU oracle at that many threads, one JSON line
        print(json.dumps(cpu_baseline(args.cpu_baseline_only, args.cpu_baseline_shape)), flush=True)
        return
    if args.gpus > 1 and "RANK" not in os.environ:
        # the driver's command form is `python bench.py --gpus N ...`: become the launcher
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    if args.launch_check:
        # what the launcher hands a rank, proven without a GPU (tests/test_bench_launcher.py, gloo): rendezvous on 127.0.0.1, one all-reduce
        import datetime
        if args.comm_probe and os.environ.get("VBG_PROBE_FAIL") == "1":          # (test hook: a probe that stalls ends with the watchdog's code)
            os._exit(17)
        world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        real_stdout = os.dup(1)          # (as in the real run: gloo's connection notes must not share stdout with the JSON line)
        sys.stdout.flush()
        os.dup2(2, 1)
        dist.init_process_group(os.environ.get("VBG_DIST_BACKEND", "gloo"), rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        seen = dist.get_world_size()
        dist.barrier()
        if rank == 0:
            os.write(real_stdout, (json.dumps({"launch_check": True, "world_size": seen, "gpus_arg": args.gpus, "rank_sum": float(t.item()),
                                               "self_launched": os.environ.get("VBG_SELF_LAUNCHED") == "1", "syncbn_comm": args.syncbn_comm,
                                               "comm_fallback": bool(args.comm_fallback)}) + "\n").encode())
        dist.destroy_process_group()
        return

    # stdout carries the ONE JSON line and nothing else: file descriptor 1 is pointed at stderr for the rest of the run (librccl prints a
    # version banner to stdout when the first communicator is built, gloo its connection notes), the line goes out through a duplicate
    real_stdout = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP library is the product, there is no CPU fallback")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # VBG_FORCE_REDUCER=1 with --gpus 1: a process group of ONE rank on the real backend (ProcessGroupNCCL = RCCL), FlatReducer and the
    # SyncBatchNorm statistics collectives running through it -- the code path of N > 1 on the one GPU of a test box
    forced = world == 1 and os.environ.get("VBG_FORCE_REDUCER", "0") != "0"
    if forced:
        import socket
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
        sk.close()
    cpu_pg = None
    if world > 1 or forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # VBG_DIST_BACKEND=gloo lets two ranks share ONE GPU for functional validation of the N>1 path
        import datetime
        dist.init_process_group(os.environ.get("VBG_DIST_BACKEND", "nccl"), rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=args.dist_timeout),
                                **({"device_id": dev} if os.environ.get("VBG_DIST_BACKEND", "nccl") == "nccl" else {}))
        # host-side agreement on per-step decisions (gradient clipping): a gloo group over the loopback interface (single node)
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        try:
            cpu_pg = dist.new_group(backend="gloo")
        except Exception as e:          # no host-side group: every rank decides on its own loss, like the reference loop
            cpu_pg = None
            print(f"[bench] gloo side group unavailable ({type(e).__name__}: {e}); clipping decided per rank", file=sys.stderr, flush=True)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"

    from vbg import functions as vbg_functions
    from vbg import ops
    from vbg.lib import OP_DENSE_K
    from vbg.optim import FlatReducer, FusedAdamW, FusedSGD, clip_grad_norm_, split_parameters

    torch.manual_seed(42)
    random.seed(42 + rank)
    tmp = tempfile.mkdtemp(prefix="vbg_bench_")
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):        # the reference's ctor prints; stdout carries the ONE JSON line only
        shape = {"cfg2": dict(img=512, S=128, ncls=NCLS, vocab=VOCAB, backbone="resnet_34_fpn_pretrained", batch=8, roberta=False),
                 "cfg3": dict(img=512, S=128, ncls=4, vocab=VOCAB, backbone="resnet_34_fpn", batch=8, roberta=False),
                 "cfg4": dict(img=512, S=512, ncls=12, vocab=21128, backbone="resnet_34_fpn", batch=8, roberta=False),
                 "cfg5": dict(img=1024, S=128, ncls=NCLS, vocab=50265, backbone="resnet_34_fpn", batch=16, roberta=True)}[args.shape]
        net = build_model(tmp, backbone=shape["backbone"], vocab=shape["vocab"], img=shape["img"], ncls=shape["ncls"], roberta=shape["roberta"])
    sync_bn = (world > 1 or forced) and not args.no_syncbn
    if sync_bn:
        net = torch.nn.SyncBatchNorm.convert_sync_batchnorm(net)      # example_config.yaml syncBN: True
    net = net.to(dev).train()
    cnn, bert = split_parameters(net)
    opt_cnn = FusedSGD(cnn, dev, lr=0.005, momentum=0.9, weight_decay=0.005)
    opt_bert = FusedAdamW(bert, dev, lr=5e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    opts = [opt_cnn, opt_bert]
    # one communicator in flight by default (classifier_mode simp: the same graph on every rank, every step); --syncbn-comm own
    # is the overlapped two-communicator form of round 3, kept as the A/B
    # (static_graph: classifier_mode simp runs the same autograd graph on every rank, every step -- the word FlatReducer needs to let the
    #  buckets leave from inside backward while the SyncBatchNorm statistics share their communicator)
    gloo_run = os.environ.get("VBG_DIST_BACKEND", "nccl") != "nccl"          # (functional runs of several ranks on one GPU: no RCCL)
    reducer = FlatReducer(opts, sync_bn_group={"own": "new", "shared": "default", "direct": "default" if gloo_run else "direct"}[args.syncbn_comm],
                          overlap=not args.no_ddp_overlap,
                          static_graph=True, force_enable=forced)
    if world > 1 or forced:
        reducer.start_watchdog(max(30.0, args.dist_timeout / 2))        # a stalled step leaves a line per rank on stderr

    B = args.batch or shape["batch"]
    batch = synthetic_batch(B, shape["img"], shape["img"], 512, shape["S"], shape["ncls"], shape["vocab"], 1234 + rank)
    from vbg.batch import AsyncScalar, PackedBatch
    packed_src = PackedBatch.pack(*batch)
    # resident batch: uploaded ONCE, outside the timed region, through the same packed buffer (device views that keep the host copy
    # of the index tensors they came from, so the model builds its index tables without a device->host copy, vbg/batch.py)
    dbatch = packed_src.to(dev)
    torch.cuda.synchronize()
    use_h2d = [bool(args.h2d)]

    amp_on = [bool(args.amp)]
    if args.fp32_mfma:
        ops.set_precision("fp32")

    def step():
        # `amp: True` = the reference's autocast region around the model call (pipeline/train_val_utils.py:264)
        with torch.autocast("cuda", dtype=torch.float16, enabled=amp_on[0]):      # (torch.cuda.amp.autocast's dtype: fp16)
            loss = net(*(packed_src.to(dev) if use_h2d[0] else dbatch))
        # train_loss.item() of the reference loop (pipeline/train_val_utils.py:270), read through a side stream so that it does not
        # park the GPU between forward and backward (--sync-loss: the blocking read at the reference's position)
        val = loss.item() if args.sync_loss else AsyncScalar(loss)
        opt_cnn.zero_grad()
        opt_bert.zero_grad()
        loss.backward()
        reducer.finish()
        if world > 1 and os.environ.get("VBG_SYNC_DEBUG"):
            named = cnn + bert
            gs = torch.stack([p.grad.detach().double().sum() for _, p in named])
            lo3, hi3 = gs.clone(), gs.clone()
            dist.all_reduce(lo3, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi3, op=dist.ReduceOp.MAX)
            badg = (lo3 != hi3).nonzero().flatten().tolist()
            if rank == 0:
                print(f"[sync check] gradients after finish(): {len(badg)} differ; first {[named[i][0] for i in badg[:6]]}", file=sys.stderr, flush=True)
        if not args.sync_loss:
            val = val.get()
        clip = val > 10
        if world > 1 and cpu_pg is not None:
            # the reference decides on the LOCAL loss (pipeline/train_val_utils.py:280-281); ranks that decide differently scale the same
            # all-reduced gradient differently and their parameters drift apart.  Agree on the decision (any rank over the threshold)
            # through a host-side gloo group: the GPU queue is not touched.
            cf = torch.tensor([1.0 if clip else 0.0])
            dist.all_reduce(cf, op=dist.ReduceOp.MAX, group=cpu_pg)
            clip = bool(cf.item() > 0)
        if clip:
            clip_grad_norm_(opts, 2.0, 1.0 / world)
        opt_cnn.step()
        opt_bert.step()
        if world > 1 and not args.no_step_barrier:
            dist.barrier()          # `if distributed: torch.distributed.barrier()` of the reference loop (pipeline/train_val_utils.py:286-287)
        return val

    if args.comm_probe:
        # child of self_launch: does the two-communicator configuration make progress on THIS node?  Three dry steps; a step that makes no
        # progress for 20 s ends the process with code 17 (the watchdog prints which bucket / statistics collective every rank is at)
        reducer.start_watchdog(20.0, exit_code=17)
        for _ in range(3):
            step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        if rank == 0:
            print(f"[bench] communicator probe passed: 3 steps, {int(vbg_functions.SyncCtx.seq)} SyncBatchNorm collectives on '{reducer.sync_bn_mode}'", file=sys.stderr, flush=True)
        os._exit(0)                     # (no teardown of two communicators in a probe)
    for _ in range(args.warmup):
        last = step()

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    t_enq = time.perf_counter() - t0          # the host has enqueued every timed step (what it could not run ahead of, it waited for inside)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    def timed_leg(n_warm=2):
        for _ in range(n_warm):
            step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            lv = step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        d = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([d], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            d = float(tt.item())
        return d, lv

    # roofline leg: the same steps again with the dense NT GEMM launches timed one by one (start / stop events in the dispatch packets:
    # vbg_*_timed) -- in a leg of its own, so that the headline region above carries no instrumentation
    prof = ops.GemmProfiler(OP_DENSE_K, OP_DENSE_K, False)       # the dense NT GEMM (BERT linears, 1x1 convs): dominant kernel
    ops.set_gemm_profiler(prof)
    c3rec = []
    ops.set_conv3_profiler(c3rec)                                # the second family by time: the row-reuse 3x3 convolutions (fwd + dgrad)
    timed_leg(0)
    ops.set_gemm_profiler(None)
    ops.set_conv3_profiler(None)
    launches, flops, ms, mfma_flops = prof.summary()
    c3_ms = sum(e0.elapsed_time(e1) for _, _, e0, e1 in c3rec)
    c3_fl, c3_exec = sum(r[0] for r in c3rec), sum(r[0] * r[1] for r in c3rec)
    # In the step as it runs the encoder's launches share the chip with the CNN's first stage (second stream, VBG_OVERLAP): the step is
    # shorter, every overlapped launch longer -- the figures above are those of the launches AS RUN (what a kernel trace of this command
    # shows).  One more pass on ONE stream gives the same launches alone on the chip: `roofline*.single_stream`.
    single = None
    if ops.overlap_enabled() and world == 1 and not args.no_single_stream_pass:
        ops.set_overlap(False)
        prof1, c3rec1 = ops.GemmProfiler(OP_DENSE_K, OP_DENSE_K, False), []
        ops.set_gemm_profiler(prof1)
        ops.set_conv3_profiler(c3rec1)
        timed_leg(1)
        ops.set_gemm_profiler(None)
        ops.set_conv3_profiler(None)
        ops.set_overlap(True)
        l1, f1, ms1, mf1 = prof1.summary()
        c3_ms1 = sum(e0.elapsed_time(e1) for _, _, e0, e1 in c3rec1)
        single = {"nt": (l1, ms1, mf1), "c3": (len(c3rec1), c3_ms1, sum(r[0] * r[1] for r in c3rec1))}

    h2d_leg = None
    if not args.h2d and not args.no_h2d_leg:      # the same steps with the batch uploaded inside every step (SURVEY §8d step body)
        use_h2d[0] = True
        hdt, _ = timed_leg()
        use_h2d[0] = False
        h2d_leg = {"value": round(B * world * args.steps / hdt, 3), "unit": "docs/sec", "ms_per_step": round(1e3 * hdt / args.steps, 3),
                   "bytes_per_step": packed_src.nbytes(), "how": "one pinned packed buffer, one asynchronous H2D copy per step (vbg/batch.py)"}
    stock_leg = None
    if not args.no_stock_leg and (world == 1 or args.stock) and not args.fp32_mfma:
        def make_net():
            with contextlib.redirect_stdout(sys.stderr):
                torch.manual_seed(42)
                return build_model(tempfile.mkdtemp(prefix="vbg_bench_stock_"), backbone=shape["backbone"], vocab=shape["vocab"], img=shape["img"],
                                   ncls=shape["ncls"], roberta=shape["roberta"])
        sdt, slast = stock_loop_leg(make_net, batch, dev, world, args.steps, 3, amp=bool(args.amp), sync_bn=sync_bn, barrier=not args.no_step_barrier)
        rdt, _ = stock_loop_leg(make_net, batch, dev, world, args.steps, 3, amp=bool(args.amp), resident=True, sync_bn=sync_bn, barrier=not args.no_step_barrier)
        stock_leg = {"value": round(B * world * args.steps / sdt, 3), "unit": "docs/sec", "ms_per_step": round(1e3 * sdt / args.steps, 3),
                     "vs_headline": round((B * world * args.steps / sdt) / (B * world * args.steps / dt), 4), "last_loss": round(float(slast), 4),
                     "resident_inputs": {"value": round(B * world * args.steps / rdt, 3), "ms_per_step": round(1e3 * rdt / args.steps, 3),
                                         "what": "the same loop with the six collate outputs uploaded once (no per-step H2D): what the loop's "
                                                 f"{4 * B + 2} pageable .to(device) copies cost is the difference"},
                     "how": "pipeline/train_val_utils.py:248-287 line for line around the drop-in model on a fresh model instance: per-tensor pageable "
                            ".to(device), autocast(enabled=amp), train_loss.item(), optimizer.zero_grad() x2 (set_to_none), backward, "
                            "`train_loss > 10` on the device tensor, torch.optim.SGD + torch.optim.AdamW (GradScaler when amp); nothing from vbg.optim / vbg.batch"}
    amp_leg = None
    if not args.amp and not args.no_amp_leg:      # the same steps with `amp: True` (reported beside the fp32 headline, never as it)
        amp_on[0] = True
        for _ in range(2):
            step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            amp_last = step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        adt = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([adt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            adt = float(t.item())
        amp_on[0] = False
        # ... and with every product on the fp32 matrix pipe (the form the split form replaces)
        fp32_leg = strict_leg = None
        if not args.fp32_mfma:
            ops.set_precision("fp32")
            for _ in range(2):
                step()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for _ in range(args.steps):
                step()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            fdt = time.perf_counter() - t2
            if world > 1:
                t = torch.tensor([fdt], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                fdt = float(t.item())
            ops.set_precision("split")
            fp32_leg = {"value": round(B * world * args.steps / fdt, 3), "unit": "docs/sec", "ms_per_step": round(1e3 * fdt / args.steps, 3),
                        "dtype": "f32 MFMA (v_mfma_f32_32x32x2_f32) for every product"}
            # ... and the strict six-product form everywhere: no fp16-pair products (forward BERT linears, wide 3x3 convolutions)
            ops.set_pair(False)          # (also takes the BERT backward off the pair form)
            ops.set_conv3_f16(False)
            ops.set_gemm_f16(False)
            sdt, _ = timed_leg()
            ops.set_pair(True)
            ops.set_conv3_f16(True)
            ops.set_gemm_f16(True)
            strict_leg = {"value": round(B * world * args.steps / sdt, 3), "unit": "docs/sec", "ms_per_step": round(1e3 * sdt / args.steps, 3),
                          "dtype": "three bf16 pieces per operand / six piece products for EVERY fp32-grade product (VBG_PAIR=0 VBG_CONV3_F16=0 VBG_GEMM_F16=0)"}
        amp_leg = {"value": round(B * world * args.steps / adt, 3), "unit": "docs/sec", "ms_per_step": round(1e3 * adt / args.steps, 3),
                   "dtype": "one reduced-precision MFMA product per product (fp16 hi pieces on the plane / row-reuse kernels, bf16 on the generic ones), "
                            "f32 accumulate / storage / everything else", "last_loss": round(float(amp_last), 4)}
    ranks_in_sync = None
    if world > 1:          # self-check of the gradient exchange: every rank must hold bit-identical parameters after the timed steps
        chk = torch.stack([o.group.pflat.double().sum() for o in opts] + [o.group.pflat.double().abs().sum() for o in opts])
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        ranks_in_sync = bool(torch.equal(lo, hi))
        if not ranks_in_sync or os.environ.get("VBG_SYNC_DEBUG"):
            # which parameters differ between ranks (per-parameter checksums, MIN / MAX over ranks)
            named = cnn + bert
            sums = torch.stack([p.detach().double().sum() for _, p in named] + [p.detach().double().abs().sum() for _, p in named])
            lo2, hi2 = sums.clone(), sums.clone()
            dist.all_reduce(lo2, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi2, op=dist.ReduceOp.MAX)
            bad = ((lo2 != hi2)[:len(named)] | (lo2 != hi2)[len(named):]).nonzero().flatten().tolist()
            if rank == 0:
                print(f"[sync check] {len(bad)} of {len(named)} parameters differ between ranks; first: {[named[i][0] for i in bad[:12]]}", file=sys.stderr, flush=True)
    # HBM-side bytes per launch of the same kernel: rocprofv3 PMC passes of this command (cannot be collected in-process),
    # summarised in profiles/ by the round that produced them; null when the file is absent
    def pmc_traffic(stem):
        for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
            tf = os.path.join(ROOT, "profiles", f"{rnd}_{stem}.json")
            if os.path.exists(tf):
                with open(tf) as fh:
                    return (round(float(json.load(fh)["bytes_per_launch"]), 0),
                            f"profiles/{rnd}_{stem}.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (counters cannot be read in-process), not this run")
        return None, None
    def trace_avg(fam):
        """average launch duration of a roofline family in the committed rocprofv3 kernel trace of this command (tools/make_profiles.py)"""
        for rnd in ("r06",):
            tf = os.path.join(ROOT, "profiles", f"{rnd}_trace_avg.json")
            if os.path.exists(tf):
                with open(tf) as fh:
                    d = json.load(fh)
                if fam in d:
                    return float(d[fam]["avg_us"]), f"profiles/{rnd}_trace_avg.json"
        return None, None
    traffic, traffic_src = pmc_traffic("gemm_hbm_traffic")
    traffic_c3, traffic_c3_src = pmc_traffic("conv3_hbm_traffic")

    if rank == 0:
        docs = B * world * args.steps
        value = docs / dt
        ach = (flops / (ms * 1e-3)) / 1e12 if ms > 0 else 0.0
        mfma_rate = (mfma_flops / (ms * 1e-3)) / 1e12 if ms > 0 else 0.0       # what the matrix pipe executes: piece products included
        mfma_peak = PEAK_F32_TF if args.fp32_mfma else PEAK_BF16_TF
        f_step = {"cfg2": F_STEP_GF, "cfg3": 689.7, "cfg4": 862.4, "cfg5": 1715.3}[args.shape]
        out = {
            "metric": "training docs/sec, 512x512 img + seq_len 512, bert-base+resnet34; 1/2/4/8 GPU",
            "value": round(value, 3), "unit": "docs/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp16+bf16" if args.amp else "f32", "data": "synthetic",
            "arithmetic": ("one reduced-precision MFMA product per product of fp32 tensors, f32 accumulate: operands rounded to fp16 (the hi pieces of the "
                           "fp16-pair planes / pre-split filters; gradients scaled into range by their amax slots) on the BERT linears and the wide 3x3 "
                           "convolutions and the attention-output projection, to bf16 on the generic kernels (1x1, heads, stem); the fused attention itself stays on six bf16 piece products" if args.amp else
                           "f32 MFMA for every product" if args.fp32_mfma else
                           "fp32-grade on the bf16 / fp16 matrix cores, f32 accumulate: (a) exact 3-way bf16 split of every operand, 6 piece products per product -- fused attention, the attention-output projection, the generic convolutions, 1x1 / heads; (b) 2 fp16 pieces per operand (round to nearest, hi + lo 2^-11: 2^-23 relative), 3 piece products, same measured error against fp64 -- the BERT linears QKV / FFN1 / FFN2 forward, all BERT data and weight gradients, the wide 3x3 convolutions forward, input gradient and weight gradient, and (round 6) the forward of the generic kernels' products on 64 x 64 tiles: strided / 1x1 convolutions, early fusion, heads (gradient operands scaled by the power of two that centres their largest magnitude in fp16's range: exact); an operand outside fp16's range becomes inf, never a clipped value; the strided conv weight gradients and the unaligned stem on the f32 MFMA; the strict form (a) everywhere is the `bf16x3_strict` leg"),
            "config": {"workload": ("SROIE line-level cfg2: resnet_34_fpn_pretrained + bert-base-uncased (12L, vocab 30522, random init), "
                                    f"512x512, T=512 tokens, S=128 segments, batch {B}/GPU, fwd+bwd+SGD/AdamW, dropout 0.1, simp classifier")
                       if args.shape == "cfg2" else f"EXPLORATORY {args.shape}: {shape}, T=512, batch {B}/GPU (not the BASELINE metric's configuration)",
                       "global_batch": B * world, "seq_len": 512, "parallelism": f"dp{world}" + ("+syncbn" if sync_bn else ""),
                       **({"step_barrier": not args.no_step_barrier, "syncbn_comm": reducer.sync_bn_mode, "ddp_overlap": reducer.overlap,
                           "buckets": len(reducer.buckets), "backend": dist.get_backend(), "syncbn_collectives": int(vbg_functions.SyncCtx.seq),
                           "world_size": dist.get_world_size(),          # as torch.distributed sees it (== --gpus, asserted above)
                           "comm_nranks": {"buckets": dist.get_world_size(),
                                           "syncbn": (vbg_functions.SyncCtx.direct.nranks() if vbg_functions.SyncCtx.direct is not None else dist.get_world_size())},
                           "launcher": "self (bench.py re-executed under torch.distributed.run)" if os.environ.get("VBG_SELF_LAUNCHED") == "1" else "external",
                           **({"syncbn_comm_fallback": "the --syncbn-comm direct probe failed; shared communicator used"} if args.comm_fallback else {}),
                           **({"forced_reducer_on_one_rank": True} if forced else {})} if (world > 1 or forced) else {}),
                       "last_loss": round(float(last), 4), **({"ranks_in_sync": ranks_in_sync} if ranks_in_sync is not None else {}), **({"h2d_in_step": packed_src.nbytes()} if args.h2d else {})},
            # algorithmic (fp32-equivalent, PAD-free) TFLOP/s of the whole step per GPU, SURVEY.md 8d, and as a fraction of the matrix-core
            # ceiling of the arithmetic form the step's products run by default: dense peak / 3 piece products (two fp16 pieces per operand),
            # dense peak / 1 under amp, the fp32 MFMA peak with --fp32-mfma
            "step_tflops": round(value / world * f_step / 1e3, 2),
            "step_frac": round(value / world * f_step / 1e3 / (PEAK_F32_TF if args.fp32_mfma else PEAK_BF16_TF / (1 if args.amp else 3)), 4),
            "step_frac_peak": round(PEAK_F32_TF if args.fp32_mfma else PEAK_BF16_TF / (1 if args.amp else 3), 1),
            # who paces the step: the host had enqueued all timed steps after `host_enqueue_ms_per_step` x steps; `device_tail_ms` = what the
            # device still had queued at that moment (a few ms: the device paces and the host runs ahead; ~0: the host does)
            "host": {"enqueue_ms_per_step": round(1e3 * t_enq / args.steps, 3), "device_tail_ms": round(1e3 * (dt - t_enq), 3)},
        }
        # Roofline objects (the step is bound by the matrix cores).  Definition, the same for both: `achieved` = ALGORITHMIC
        # (fp32-equivalent, 2 M N K per product; only real pixels on the 7x7 region maps) TFLOP/s over the launches' own time -- event pairs
        # in the dispatch packets / on the launch stream, in a leg of its own --; `peak` = the dense bf16 / fp16 MFMA peak divided by the
        # piece products the arithmetic form issues per product (3: two fp16 pieces, 6: three bf16 pieces; the launch-weighted mean when a
        # family mixes them); `frac` = achieved / peak = executed matrix-core flops / the dense peak (`mfma_executed` / `mfma_peak`).
        pp_nt = mfma_flops / max(flops, 1.0)
        nt = {"bound": "mfma",
              "kernel": ("vbg::plane_gemm_kernel<*,*,*,*,*,false,*,2> + vbg::gemm_kernel<*,*,*,DENSE_K,DENSE_K,*,1> (one-product NT GEMM, amp: fp16 hi planes / bf16; every ungrouped launch)" if args.amp else
                         "vbg::gemm_kernel<*,*,*,DENSE_K,DENSE_K,*,0> (fp32 MFMA NT GEMM; every ungrouped launch)" if args.fp32_mfma else
                         "vbg::plane_gemm_kernel<*,*,*,*,*,false,*,0|1> + vbg::gemm_kernel<*,*,*,DENSE_K,DENSE_K,*,3> (fp32-grade NT GEMM: the BERT linears from pre-split planes, 1x1 convs / heads with the in-kernel split; every ungrouped launch)"),
              "achieved": round(ach, 2), "peak": round(mfma_peak / max(pp_nt, 1.0), 1), "unit": "TFLOP/s", "frac": round(mfma_rate / mfma_peak, 4),
              "mfma_executed": round(mfma_rate, 1), "mfma_peak": round(mfma_peak, 1), "piece_products_per_product": round(pp_nt, 3),
              "traffic": traffic, "traffic_source": traffic_src, "launches": launches, "avg_us": round(1e3 * ms / max(launches, 1), 2),
              "ms_per_step": round(ms / args.steps, 3), "vs_fp32_mfma_peak": round(ach / PEAK_F32_TF, 4)}
        tavg, tsrc = trace_avg("nt")
        if tavg and launches and not (args.amp or args.fp32_mfma):
            # `frac` with the committed kernel trace's average launch duration in place of this run's event pairs
            nt["frac_trace"] = round(mfma_flops / launches / (tavg * 1e-6) / 1e12 / mfma_peak, 4)
            nt["frac_trace_source"] = f"{tsrc}: {tavg:.2f} us per launch in the rocprofv3 kernel trace of this command (another run, same box class)"
        if getattr(prof, "ingest", None) and prof.ingest[0] and prof.ingest[2] > 0:
            # what the plane products of the family are bound by: bytes their workgroups move from L2 into LDS -- tiles x k-tiles
            # x stage bytes, every operand panel once per tile that uses it -- over the same launch times
            nt["l2_to_lds"] = {"launches": prof.ingest[0], "MB_per_launch": round(prof.ingest[1] / prof.ingest[0] / 1e6, 1),
                               "TB_per_s": round(prof.ingest[1] / (prof.ingest[2] * 1e-3) / 1e12, 2), "ms_per_step": round(prof.ingest[2] / args.steps, 3)}
        if c3rec and c3_ms > 0:
            # the dominant kernel family by time since round 3: the row-reuse 3x3 convolutions, forward + input gradient
            pp_c3 = c3_exec / max(c3_fl, 1.0)
            c3 = {"bound": "mfma", "kernel": "vbg::conv3x3_kernel<*,*,*> (3x3 / stride-1 convolutions, forward + input gradient, csrc/conv3.hip)",
                               "achieved": round(c3_fl / c3_ms / 1e9, 2), "peak": round(mfma_peak / max(pp_c3, 1.0), 1), "unit": "TFLOP/s",
                               "frac": round(c3_exec / c3_ms / 1e9 / mfma_peak, 4), "mfma_executed": round(c3_exec / c3_ms / 1e9, 1), "mfma_peak": round(mfma_peak, 1),
                               "piece_products_per_product": round(pp_c3, 3), "traffic": traffic_c3, "traffic_source": traffic_c3_src,
                               "launches": len(c3rec), "avg_us": round(1e3 * c3_ms / len(c3rec), 2), "ms_per_step": round(c3_ms / args.steps, 3),
                               "vs_fp32_mfma_peak": round(c3_fl / c3_ms / 1e9 / PEAK_F32_TF, 4)}
            tavg, tsrc = trace_avg("conv3")
            if tavg and not (args.amp or args.fp32_mfma):
                c3["frac_trace"] = round(c3_exec / len(c3rec) / (tavg * 1e-6) / 1e12 / mfma_peak, 4)
                c3["frac_trace_source"] = f"{tsrc}: {tavg:.2f} us per launch in the rocprofv3 kernel trace of this command (another run, same box class)"
            if single is not None and single["c3"][1] > 0 and single["nt"][1] > 0:
                how = "the same launches in a pass with everything on ONE stream (VBG_OVERLAP=0): alone on the chip"
                n1, m1, x1 = single["nt"]
                nt["single_stream"] = {"frac": round(x1 / m1 / 1e9 / mfma_peak, 4), "avg_us": round(1e3 * m1 / max(n1, 1), 2),
                                       "ms_per_step": round(m1 / (args.steps + 1), 3), "how": how}          # (the pass runs one warm-up step)
                n1, m1, x1 = single["c3"]
                c3["single_stream"] = {"frac": round(x1 / m1 / 1e9 / mfma_peak, 4), "avg_us": round(1e3 * m1 / max(n1, 1), 2),
                                       "ms_per_step": round(m1 / (args.steps + 1), 3), "how": how}
                nt["as_run"] = c3["as_run"] = ("`frac` / `avg_us` are the launches as they run in the step: the encoder on a second stream beside the "
                                               "CNN's first stage -- a shorter step made of longer launches; `single_stream` is the kernel alone")
            # `roofline` = the family that takes more of the step (ms_per_step); the other one rides beside it under its own name
            if c3["ms_per_step"] >= nt["ms_per_step"]:
                out["roofline"], out["roofline_nt"] = c3, nt
            else:
                out["roofline"], out["roofline_conv3"] = nt, c3
            out["roofline"]["kernel"] += ": the largest kernel family of the step by time"
        else:
            out["roofline"] = nt
        if stock_leg is not None:
            out["stock_loop"] = stock_leg
        if h2d_leg is not None:
            out["h2d_inclusive"] = h2d_leg
        if amp_leg is not None:
            out["amp"] = amp_leg
            if fp32_leg is not None:
                out["fp32_mfma"] = fp32_leg
            if strict_leg is not None:
                out["bf16x3_strict"] = strict_leg
        if world == 1 and not args.no_cpu_baseline:
            # the CPU restatement of the step on this box's host cores, cfg2 shape (B = 2) and cfg1 beside it.  The
            # oracle's many small ops oversubscribe at torch.set_num_threads(os.cpu_count()) (256 threads: minutes per step), so the thread
            # count is swept -- each count in a child process with a time limit -- and the BEST count that finishes is the baseline
            # (`cores` = the threads it was measured with); every attempt is listed under `sweep`.
            import subprocess
            ncpu = os.cpu_count() or 1

            def child(threads, shape, limit):
                try:
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", str(threads), "--cpu-baseline-shape", shape],
                                       capture_output=True, text=True, timeout=limit)
                    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                    return json.loads(line[-1]) if line else {"value": None, "threads": threads, "note": "child failed: " + r.stderr[-200:]}
                except subprocess.TimeoutExpired:
                    return {"value": None, "threads": threads, "note": f"did not finish in {limit} s"}
            if args.cpu_threads:
                out["cpu_baseline"] = cpu_baseline(args.cpu_threads)
            else:
                counts = sorted({min(ncpu, c) for c in (16, 32, 64)} | ({ncpu} if ncpu <= 128 else set()))
                runs = [child(c, "cfg2", 40) for c in counts]
                done = [r for r in runs if r.get("value")]
                best = max(done, key=lambda r: r["value"]) if done else cpu_baseline(min(ncpu, 16))
                out["cpu_baseline"] = dict(best, sweep=[{"threads": r.get("threads"), "value": r.get("value"), **({"note": r["note"]} if r.get("note") else {})} for r in runs])
                out["cpu_baseline"]["cfg1"] = child(best["threads"], "cfg1", 40)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if world > 1 or forced:
        if vbg_functions.SyncCtx.direct is not None:          # the library's own RCCL communicator goes first
            vbg_functions.SyncCtx.direct.destroy()
            vbg_functions.SyncCtx.direct = None
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
