#!/usr/bin/env python3
"""fp16-form row-reuse 3x3 convolution (csrc/conv3.hip): filter split inside every workgroup vs the pre-split plane image streamed by
LDS-DMA (PW, round 4), forward and input gradient on the cfg2 shapes, library split counts and a sweep of them on the late stages.
   python tools/conv3_pw_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vibertgrid-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch

from gemm_bench import report, timeit
from vbg import ops

dev = torch.device("cuda")
GEN = torch.Generator(device=dev).manual_seed(5)
SHAPES = [(8, 128, 128, 256, 256), (8, 128, 128, 128, 128), (8, 64, 64, 128, 128), (8, 64, 64, 256, 256), (8, 32, 32, 256, 256),
          (8, 16, 16, 512, 512), (8, 128, 128, 64, 64), (1024, 7, 7, 256, 256)]
for (B, H, W, Ci, Co) in SHAPES:
    x = torch.randn(B, H, W, Ci, device=dev, generator=GEN)
    wd = (torch.randn(Co, Ci, 3, 3, device=dev, generator=GEN) / (3 * Ci ** 0.5)).contiguous(memory_format=torch.channels_last)
    w4 = wd.permute(0, 2, 3, 1)
    dy = torch.randn(B, H, W, Co, device=dev, generator=GEN) * 1e-6
    am = ops.amax(dy)
    fl = 2.0 * B * H * W * Ci * Co * 9
    tag = f"B{B} {H}x{W} {Ci}->{Co}"
    wp, wpf, wf = ops.conv3_planes(wd, w4, False), ops.conv3_planes(wd, w4, True), ops.conv3x3_wflip(w4)
    nz = ops.conv3_split(B, H, W, Ci, Co)
    report(f"fwd   in-kernel split  {tag} nsplit {nz}", fl, timeit(lambda: ops.conv3x3(x, w4, f16x2=True)))
    report(f"fwd   PW               {tag} nsplit {nz}", fl, timeit(lambda: ops.conv3x3(x, w4, f16x2=True, w_planes=wp)))
    nzb = ops.conv3_split(B, H, W, Co, Ci)
    report(f"dgrad in-kernel split  {tag} nsplit {nzb}", fl, timeit(lambda: ops.conv3x3(dy, wf, f16x2=True, x_amax=am)))
    report(f"dgrad PW               {tag} nsplit {nzb}", fl, timeit(lambda: ops.conv3x3(dy, w4, f16x2=True, x_amax=am, w_planes=wpf, n_out=Ci)))
    assert torch.equal(ops.conv3x3(x, w4, f16x2=True), ops.conv3x3(x, w4, f16x2=True, w_planes=wp))
    late = ops.conv3_late_choice(B, H, W, Ci, Co)
    if late is not None:          # the late trunk stages as the library runs them: 64-filter tiles, its split count
        wp64, wpf64 = ops.conv3_planes(wd, w4, False, bn=late[0]), ops.conv3_planes(wd, w4, True, bn=late[0])
        report(f"fwd   PW bn64          {tag} nsplit {late[1]}", fl, timeit(lambda: ops.conv3x3(x, w4, f16x2=True, w_planes=wp64, nsplit=late[1], bn=late[0])))
        report(f"dgrad PW bn64          {tag} nsplit {late[1]}", fl, timeit(lambda: ops.conv3x3(dy, w4, f16x2=True, x_amax=am, w_planes=wpf64, n_out=Ci, nsplit=late[1], bn=late[0])))
    if os.environ.get("VBG_BENCH_HASH"):          # bit-identity across builds: a checksum of the PW results
        y = ops.conv3x3(x, w4, f16x2=True, w_planes=wp)
        d = ops.conv3x3(dy, w4, f16x2=True, x_amax=am, w_planes=wpf, n_out=Ci)
        print(f"hash {tag}: fwd {float(y.double().sum()):.17g} {float(y.double().abs().sum()):.17g} dgrad {float(d.double().sum()):.17g} {float(d.double().abs().sum()):.17g}", flush=True)
    if H * W % 128 == 0 and H != 7 and W < 128:
        for z in (1, 2, 3, 4, 6, 8):
            cs = z // 3 if z % 3 == 0 else z
            if Ci % cs or (Ci // cs) % 16 or (z == 1 and nz > 1):
                continue
            report(f"fwd   PW               {tag} nsplit {z} (forced)", fl, timeit(lambda: ops.conv3x3(x, w4, f16x2=True, w_planes=wp, nsplit=z)))
# the prep launch itself: every 3x3 filter of a resnet-34 trunk + FPN + heads, both directions, one launch
ws = []
for (co, ci, n) in [(64, 64, 6), (128, 128, 7), (256, 256, 11), (512, 512, 5), (256, 256, 7)]:
    for _ in range(n):
        ws.append(torch.randn(co, ci, 3, 3, device=dev).contiguous(memory_format=torch.channels_last))
for w in ws:
    ops.conv3_planes(w, w.permute(0, 2, 3, 1), False)
    ops.conv3_planes(w, w.permute(0, 2, 3, 1), True)
def refresh():
    ops.bump_weight_epoch()
    ops.conv3_planes(ws[0], ws[0].permute(0, 2, 3, 1), False)
tot = sum(w.numel() for w in ws)
t = timeit(refresh)
print(f"prep of {len(ws)} filters x 2 directions ({tot / 1e6:.1f} M weights) in one launch: {t * 1e6:.1f} us, {tot * 4 * 4 / t / 1e12:.2f} TB/s", flush=True)
