# round 5, call 7: stock DDP route step by step, homing on vs off: gradients and parameters after every step
cd /root/repo
rm -rf gpurun_out/dump_* gpurun_out/prof* 2>/dev/null
for h in 1 0; do VBG_PROBE_LIVE=1 VBG_PROBE_DUMP=/tmp/probe_t$h.pt VBG_HOME=$h timeout 300 python tools/ddp_stock_probe.py 2>&1 | grep "^step" | cut -c1-120; done
python - <<'PY'
import torch
a, b = torch.load("/tmp/probe_t1.pt"), torch.load("/tmp/probe_t0.pt")
for s in range(3):
    ga, pa, la = a[s]; gb, pb, lb = b[s]
    dg = sorted(((float((ga[k] - gb[k]).norm() / (gb[k].norm() + 1e-30)), k) for k in gb if "key.bias" not in k), reverse=True)
    dp = sorted(((float((pa[k] - pb[k]).norm() / (pb[k].norm() + 1e-30)), k) for k in pb if "key.bias" not in k and "pooler" not in k), reverse=True)
    print(f"step {s+1}: loss homed {la:.7f} plain {lb:.7f}")
    print("   gradients homed vs plain, worst:", [(f"{x:.2e}", k) for x, k in dg[:5]], "median", f"{dg[len(dg)//2][0]:.2e}")
    print("   parameters after the step, worst:", [(f"{x:.2e}", k) for x, k in dp[:5]], "median", f"{dp[len(dp)//2][0]:.2e}")
PY
