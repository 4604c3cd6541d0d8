import sys, torch
sys.path.insert(0, "/root/repo/vibertgrid-pytorch_amd")
from vbg import lib as L
import ctypes as C
n = 109_482_240
d = torch.device("cuda")
p, g, m, v = (torch.randn(n, device=d) * 0.02 for _ in range(4))
v = v.abs()
P = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(3): L.lib.vbg_adamw_step(P(p), P(g), P(m), P(v), n, 5e-5, 0.9, 0.999, 1e-8, 0.01, 3, 1.0, st)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): L.lib.vbg_adamw_step(P(p), P(g), P(m), P(v), n, 5e-5, 0.9, 0.999, 1e-8, 0.01, 3, 1.0, st)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print(f"adamw {us:.1f} us  {28.0 * n / us / 1e6:.2f} TB/s  checksum {float(p.double().sum()):.6f}")
