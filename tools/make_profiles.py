#!/usr/bin/env python3
"""Summarise the rocprofv3 runs of gpurun_out/ into the committed profiles/ (names carry the round tag).

expects: gpurun_out/prof_e (kernel-trace + stats of `bench.py --steps 10 --warmup 3`), gpurun_out/pmc_f / pmc_w (FETCH_SIZE / WRITE_SIZE
passes of `bench.py --steps 3 --warmup 1`), gpurun_out/gemm_shapes.txt, gpurun_out/bench_line.json"""
import collections, csv, json, os, re, shutil, subprocess, sys
TAG = sys.argv[1] if len(sys.argv) > 1 else "r02"          # round tag of the output names
NSTEP = int(sys.argv[2]) if len(sys.argv) > 2 else 13       # steps in the kernel trace of prof_e (bench.py --steps 10 --warmup 3 --no-h2d-leg)
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/"
G = R + "gpurun_out/"


def agg(path):
    a = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(path)):
        m = re.search(r'gemm_kernel<(\d+), (\d+), (\d+), 256, (\d), (\d), (true|false), (\d)>', r['Kernel_Name'])
        if not m:
            continue
        k = m.groups()[:6] + (r['Grid_Size'],) + (m.group(7),)
        a[k][0] += 1; a[k][1] += float(r['Counter_Value']); a[k][2] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    return a


f, w = agg(G + 'pmc_f/f_counter_collection.csv'), agg(G + 'pmc_w/w_counter_collection.csv')
lines = ["# rocprofv3 --kernel-trace --pmc FETCH_SIZE  /  --pmc WRITE_SIZE  (two separate passes) over `python bench.py --steps 3 --warmup 1 --no-cpu-baseline`",
         "# per-launch averages for every vbg::gemm_kernel instantiation x grid.  Units: rocprofv3 reports KB; FETCH_SIZE is DOUBLED per",
         "# MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE is taken as is (it reproduces known output sizes exactly,",
         "# e.g. 131072x256 fp32 = 134.2 MB).  FETCH counts L2 misses (Infinity-Cache hits included), not DRAM reads.",
         "# columns: BM BN BK A-kind B-kind vec precision-form (0 fp32 MFMA, 3 bf16x3 split) | grid threads | launches/step | fetch MB | write MB | avg us (under the profiler)"]
nf = nw = n = 0
for k in sorted(f, key=lambda k: -f[k][1]):
    fm = 2 * f[k][1] / f[k][0] * 1024 / 1e6
    wm = w[k][1] / w[k][0] * 1024 / 1e6 if k in w else float('nan')
    lines.append(f"{k[0]:>4}{k[1]:>4}{k[2]:>3}  A{k[3]} B{k[4]} {k[5]:5s} prec{k[7]} grid {int(k[6]):>8d}  x{f[k][0] / 4:6.1f}  fetch {fm:8.1f} MB  write {wm:8.1f} MB  {f[k][2] / f[k][0]:8.1f} us")
    grouped = k[7] == '0' and k[5] == 'true'            # (fp32-form DENSE_K x DENSE_K launches are the grouped attention-score products and the unaligned stem)
    if k[3] == '0' and k[4] == '0' and not grouped:
        nf += 2 * f[k][1] * 1024; nw += w[k][1] * 1024 if k in w else 0; n += f[k][0]
tot_f = sum(2 * v[1] * 1024 for v in f.values()) / 4
tot_w = sum(v[1] * 1024 for v in w.values()) / 4
lines.append(f"# all GEMM launches: fetch {tot_f / 1e9:.1f} GB + write {tot_w / 1e9:.1f} GB per step (before the XCD-aware block->tile map: fetch 112.7 GB per step)")
lines.append(f"# ungrouped dense NT GEMM (A0 B0; the launches bench.py's roofline times): {n / 4:.0f} launches/step, fetch {nf / n / 1e6:.1f} MB + write {nw / n / 1e6:.1f} MB per launch")
open(R + 'profiles/' + TAG + '_gemm_hbm_traffic.txt', 'w').write("\n".join(lines) + "\n")
json.dump({"kernel": "vbg::gemm_kernel<*,*,*,DENSE_K,DENSE_K> (every ungrouped dense NT launch of one bench step: the launches bench.py times)",
           "launches_per_step": n / 4, "fetch_bytes_per_launch": nf / n, "write_bytes_per_launch": nw / n, "bytes_per_launch": (nf + nw) / n,
           "source": "profiles/' + TAG + '_gemm_hbm_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE x2 gfx950 correction; L2 misses incl. Infinity-Cache hits)"},
          open(R + 'profiles/' + TAG + '_gemm_hbm_traffic.json', 'w'), indent=1)
print(lines[-2]); print(lines[-1])
shutil.copy(G + 'prof_e/e_kernel_stats.csv', R + 'profiles/' + TAG + '_bench_kernel_stats.csv')
shutil.copy(G + 'prof_e/e_domain_stats.csv', R + 'profiles/' + TAG + '_bench_domain_stats.csv')
shutil.copy(G + 'gemm_shapes.txt', R + 'profiles/' + TAG + '_gemm_shapes.txt')
shutil.copy(G + 'bench_line.json', R + 'profiles/' + TAG + '_bench_line.json')
# ---- MFMA utilisation per instantiation (gpurun_out/pmc_m: SQ_VALU_MFMA_BUSY_CYCLES ... GRBM_GUI_ACTIVE pass of `bench.py --steps 3 --warmup 1`)
if os.path.exists(G + 'pmc_m/m_counter_collection.csv'):
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(G + 'pmc_m/m_counter_collection.csv')):
        if 'gemm_kernel' not in r['Kernel_Name']:
            continue
        d = disp.setdefault(r['Dispatch_Id'], {'name': r['Kernel_Name'], 'ns': int(r['End_Timestamp']) - int(r['Start_Timestamp'])})
        d[r['Counter_Name']] = float(r['Counter_Value'])
    ag = collections.defaultdict(lambda: collections.defaultdict(float))
    ctrs = ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_LDS_BANK_CONFLICT', 'GRBM_GUI_ACTIVE')
    for d in disp.values():
        k = re.search(r'gemm_kernel<(\d+), (\d+), (\d+), 256, (\d), (\d), (true|false), (\d)>', d['name']).groups()
        a_ = ag[k]; a_['n'] += 1; a_['ns'] += d['ns']
        for c in ctrs:
            a_[c] += d.get(c, 0.0)
    names = {'0': 'DENSE_K', '1': 'DENSE_R', '2': 'CONV_K', '3': 'CONV_R', '4': 'WT_R'}
    out = ["# rocprofv3 --kernel-trace --pmc " + " ".join(ctrs),
           "# over `python bench.py --steps 3 --warmup 1 --no-cpu-baseline`, summed per gemm_kernel instantiation over the 4 steps.",
           "# mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (kernel time * clock * 1024 SIMDs), clock = GRBM_GUI_ACTIVE / time / 8 XCDs;",
           "# wait/active columns are fractions of SQ_WAVE_CYCLES.",
           "tile,A,B,vec,form,launches_per_step,ms_per_step,clock_GHz,mfma_util,wait_any,wait_inst_any,active_inst_any,lds_bank_conflict_cycles"]
    tb = tc = 0
    for k, a_ in sorted(ag.items(), key=lambda kv: -kv[1]['ns']):
        clk = a_['GRBM_GUI_ACTIVE'] / a_['ns'] / 8
        cap = a_['ns'] * clk * 1024
        tb += a_['SQ_VALU_MFMA_BUSY_CYCLES']; tc += cap
        wc = a_['SQ_WAVE_CYCLES'] or 1
        out.append(f"{k[0]}x{k[1]}x{k[2]},{names[k[3]]},{names[k[4]]},{k[5]},prec{k[6]},{a_['n'] / 4:.1f},{a_['ns'] / 4 / 1e6:.3f},{clk:.2f},{a_['SQ_VALU_MFMA_BUSY_CYCLES'] / cap:.3f},"
                   f"{a_['SQ_WAIT_ANY'] / wc:.3f},{a_['SQ_WAIT_INST_ANY'] / wc:.3f},{a_['SQ_ACTIVE_INST_ANY'] / wc:.3f},{a_['SQ_LDS_BANK_CONFLICT']:.0f}")
    out.append(f"# all GEMM / conv kernels of the step: MFMA pipe busy {tb / tc:.3f} of the time they run")
    open(R + 'profiles/' + TAG + '_gemm_mfma_util.txt', 'w').write("\n".join(out) + "\n")
    print(out[-1])
tab = subprocess.run([sys.executable, R + 'tools/hbm_table.py', G + 'prof_e/e_kernel_trace.csv', str(NSTEP)], capture_output=True, text=True).stdout
open(R + 'profiles/' + TAG + '_hbm_kernels.txt', 'w').write("# tools/hbm_table.py over the kernel trace of `bench.py --steps 10 --warmup 3` (13 steps): achieved HBM-side bandwidth of the\n"
                                                    "# streaming kernels = algorithmic bytes per step (cfg2 shapes, fp32, every tensor read / written once) / kernel time per step\n" + tab)
print(tab)
rows = list(csv.DictReader(open(R + 'profiles/' + TAG + '_bench_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows) / NSTEP / 1e6
cat = collections.OrderedDict()
for r in rows:
    nme, v = r['Name'], float(r['TotalDurationNs']) / NSTEP / 1e6
    if 'gemm_kernel' in nme:
        m = re.search(r'256, (\d), (\d)', nme)
        key = {('0', '0'): 'dense NT (fwd, 1x1)', ('0', '1'): 'dense NN (dgrad)', ('1', '1'): 'dense TN (wgrad)', ('2', '0'): 'conv fwd', ('2', '4'): 'conv dgrad', ('1', '3'): 'conv wgrad'}[m.groups()]
    elif 'bn_' in nme: key = 'BatchNorm'
    elif 'ln_' in nme or 'softmax' in nme or 'gelu' in nme: key = 'LN/softmax/GELU'
    elif 'adamw' in nme or 'sgd' in nme: key = 'optimizers'
    elif 'at::native' in nme or 'rocclr' in nme: key = 'torch plumbing'
    else: key = 'other vbg kernels'
    cat[key] = cat.get(key, 0) + v
print("kernel ms/step", round(tot, 2))
for k, v in cat.items():
    print(f"{k:22s} {v:6.2f} ms {100 * v / tot:5.1f} %")
for r in rows:
    if 'gemm_kernel' in r['Name'] and ', 0, 0, ' in r['Name']:
        print(r['Name'][:64], int(r['Calls']) // NSTEP, round(float(r['AverageNs']) / 1e3, 1))
print(open(R + 'profiles/' + TAG + '_bench_line.json').read()[:2000])
