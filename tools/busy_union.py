#!/usr/bin/env python3
"""How busy is the chip during the timed steps?  From a rocprofv3 kernel trace of `bench.py` (several streams): the union of the kernels'
[start, end) intervals over the last `nsteps` steps of the trace, the time with 0 / 1 / 2 / 3+ kernels in flight, and the largest holes
(nothing in flight) with the kernels on either side.
    python tools/busy_union.py <kernel_trace.csv> [steps_in_trace=13] [timed_steps=10]"""
import csv, re, sys
path = sys.argv[1]; total_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 13; nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path)))
per = len(rows) // total_steps
seg = rows[len(rows) - nsteps * per:]
short = lambda s: re.sub(r"^void |vbg::|\(.*$", "", s)[:60]
ev = []
for s, e, _ in seg:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
t0, t1 = seg[0][0], max(e for _, e, _ in seg)
hist = {}
cur, last = 0, t0
for t, d in ev:
    hist[min(cur, 3)] = hist.get(min(cur, 3), 0) + (t - last)
    cur += d; last = t
span = t1 - t0
print(f"launches/step {per}; span/step {span / nsteps / 1e6:.2f} ms; kernel time summed/step {sum(e - s for s, e, _ in seg) / nsteps / 1e6:.2f} ms")
print("time per step with k kernels in flight: " + ", ".join(f"k={'3+' if k == 3 else k}: {v / nsteps / 1e6:.2f} ms" for k, v in sorted(hist.items())))
# holes: sweep by end time
holes = []
reach, prev = seg[0][1], seg[0][2]
for s, e, n in seg[1:]:
    if s > reach:
        holes.append((s - reach, short(prev), short(n)))
    if e > reach:
        reach, prev = e, n
holes.sort(reverse=True)
print(f"holes (nothing in flight): {len(holes) / nsteps:.0f} per step, {sum(h[0] for h in holes) / nsteps / 1e6:.2f} ms per step; > 10 us: {sum(1 for h in holes if h[0] > 10000) / nsteps:.1f} per step = {sum(h[0] for h in holes if h[0] > 10000) / nsteps / 1e6:.2f} ms")
agg = {}
for g, a, b in holes:
    if g > 5000:
        k = (a, b); agg.setdefault(k, [0, 0]); agg[k][0] += 1; agg[k][1] += g
for (a, b), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:20]:
    print(f"  {t / nsteps / 1e3:8.1f} us/step x{c / nsteps:5.1f}  {a}  ->  {b}")
