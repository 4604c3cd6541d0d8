import sys, torch
sys.path.insert(0, "/root/repo/vibertgrid-pytorch_amd")
from vbg import ops
dev = torch.device("cuda")
for (M, C) in [(524288, 64), (131072, 64), (32768, 128), (8192, 256), (2048, 512), (131072, 256)]:
    x = torch.randn(M, C, device=dev); dy = torch.randn(M, C, device=dev); y = torch.relu(x)
    st = ops.bn_stats(x); mean, invstd = ops.bn_finalize(st, C, ops.bn_slots(), M, 1e-5, 0.1, None, None)
    def t(f, n=20):
        for _ in range(3): f()
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
    a = t(lambda: ops.bn_stats(x)); ops.bn_fold(st, C); b = t(lambda: ops.bn_bwd_reduce(dy, y, x, mean, invstd, True)); ops.bn_fold(st, C)
    c = t(lambda: ops.bn_finalize(st, C, ops.bn_slots(), M, 1e-5, 0.1, None, None))
    print(f"M={M} C={C}: stats {a:.1f} us ({M*C*4/a/1e6:.2f} TB/s)  bwd_reduce {b:.1f} us ({3*M*C*4/b/1e6:.2f} TB/s) finalize {c:.1f} us")
