cd /root/repo
python - <<'PY' 2>&1 | grep -v Warning
import sys, torch
sys.path.insert(0, 'vibertgrid-pytorch_amd'); sys.path.insert(0, 'tools')
from vbg import ops
from vbg.lib import lib
from gemm_bench import timeit
d = torch.device('cuda')
torch.set_grad_enabled(False)
g = torch.Generator().manual_seed(1)
M, N, K = 1024, 1024, 512
dy = torch.randn(M, N, generator=g).to(d); w = (torch.randn(N, K, generator=g) / 32).to(d); x = torch.randn(M, K, generator=g).to(d)
sl = ops.amax(dy)
log = ops.dispatch_log(True)
a = ops.linear_dgrad(dy, w, dy_amax=sl)
print(log); ops.dispatch_log(False)
b = ops.linear_dgrad(dy, w)
print('NN equal', torch.equal(a, b), float((a - b).abs().max()))
print('NN with amax', timeit(lambda: ops.linear_dgrad(dy, w, dy_amax=sl)) * 1e6, 'us; without', timeit(lambda: ops.linear_dgrad(dy, w)) * 1e6)
dw1, dw2 = torch.zeros(N, K, device=d), torch.zeros(N, K, device=d)
ops.linear_wgrad(dy, x, dw1, accumulate=True, dy_amax=sl); ops.linear_wgrad(dy, x, dw2, accumulate=True)
print('TN equal', torch.equal(dw1, dw2), float((dw1 - dw2).abs().max()))
print('TN with amax', timeit(lambda: ops.linear_wgrad(dy, x, dw1, accumulate=True, dy_amax=sl)) * 1e6, 'us; without', timeit(lambda: ops.linear_wgrad(dy, x, dw2, accumulate=True)) * 1e6)
for (M, N, K) in ((1024, 12544, 1024), (131072, 256, 256)):
    dy = torch.randn(M, N, generator=g).to(d); x = torch.randn(M, K, generator=g).to(d); sl = ops.amax(dy); dw1 = torch.zeros(N, K, device=d)
    print('TN', M, N, K, 'with amax', timeit(lambda: ops.linear_wgrad(dy, x, dw1, accumulate=True, dy_amax=sl)) * 1e6, 'us; without', timeit(lambda: ops.linear_wgrad(dy, x, dw1, accumulate=True)) * 1e6)
PY
