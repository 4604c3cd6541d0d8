# round 5, call 6: stock DDP probe: step-1 gradients with homing on vs off (AdamW's first step hides a wrong gradient scale); forced reducer stderr
cd /root/repo
mkdir -p gpurun_out
for h in 1 0; do VBG_PROBE_DUMP=gpurun_out/probe_g$h.pt VBG_HOME=$h timeout 300 python tools/ddp_stock_probe.py 2>&1 | grep "^step 2" | cut -c1-200; done
python - <<'PY'
import torch
a, b = torch.load("gpurun_out/probe_g1.pt"), torch.load("gpurun_out/probe_g0.pt")
d = sorted(((float((a[k] - b[k]).norm() / (b[k].norm() + 1e-30)), float(a[k].norm() / (b[k].norm() + 1e-30)), k) for k in b if k in a), reverse=True)
print("step-1 gradients, homed vs plain under DDP: worst (rel-L2, norm ratio, name):", [(f"{x:.2e}", f"{r:.4f}", k) for x, r, k in d[:12]])
print("median", d[len(d) // 2])
print("missing in homed:", [k for k in b if k not in a][:5], "extra:", [k for k in a if k not in b][:5])
PY
VBG_FORCE_REDUCER=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-leg --no-h2d-leg --no-stock-leg > gpurun_out/r5c6_forced.json 2> gpurun_out/r5c6_forced.err; echo "rc=$?"; tail -12 gpurun_out/r5c6_forced.err | cut -c1-300
