"""largest idle gaps between consecutive kernels of the timed steps in a rocprofv3 kernel trace: which kernels sit on either side
   python tools/gap_report.py gpurun_out/prof_g/g_kernel_trace.csv [nsteps_in_trace=13] [top=25] [steps after the timed ones=0]"""
import csv, re, sys, collections
path = sys.argv[1]; nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 13; top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(path))]
rows.sort()
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0
n = len(rows); per = n // nsteps; seg = rows[n - (10 + skip) * per:n - skip * per]
short = lambda s: re.sub(r"^void |vbg::|\(.*$", "", s)[:70]
gaps = [(seg[i + 1][0] - seg[i][1], short(seg[i][2]), short(seg[i + 1][2])) for i in range(len(seg) - 1)]
busy = sum(e - s for s, e, _ in seg); span = seg[-1][1] - seg[0][0]
print(f"launches/step {per}  span/step {span/10/1e6:.2f} ms  busy/step {busy/10/1e6:.2f} ms  idle/step {(span-busy)/10/1e6:.2f} ms")
hist = collections.Counter()
for g, a, b in gaps:
    hist[min(int(g / 1000) // 5 * 5, 200)] += g
print("idle ms/step by gap size (us bucket):", {k: round(v / 10 / 1e6, 3) for k, v in sorted(hist.items())})
agg = collections.defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    if g > 8000:
        agg[(a, b)][0] += 1; agg[(a, b)][1] += g
print("pairs (before -> after) by total idle time, gaps > 8 us:")
for (a, b), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"  {t/10/1e3:8.1f} us/step  x{c/10:5.1f}  {a}  ->  {b}")

# where in the step the idle time sits: the launches of ONE step (the last of the window) in groups of 40
one = seg[-per:]
print("one step, groups of 40 launches: idle us | busy us | first kernel of the group")
for g0 in range(0, len(one) - 1, 40):
    grp = one[g0:g0 + 41]
    idle = sum(grp[i + 1][0] - grp[i][1] for i in range(len(grp) - 1))
    busy = sum(e - s_ for s_, e, _ in grp[:-1])
    print(f"  {g0:5d}  idle {idle / 1e3:7.0f}  busy {busy / 1e3:7.0f}   {short(grp[0][2])}")
