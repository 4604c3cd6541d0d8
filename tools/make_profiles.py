#!/usr/bin/env python3
"""Summarise the rocprofv3 runs of gpurun_out/ into the committed profiles/ (names carry the round tag).

    python tools/make_profiles.py [r04] [steps in the kernel trace = 23] [steps in the PMC traces = 7]

expects (tools/collect_profiles.sh): gpurun_out/prof_e (kernel-trace + stats of `bench.py --steps 10 --warmup 3 --no-h2d-leg`),
gpurun_out/pmc_f / pmc_w (FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 3 --warmup 1 --no-h2d-leg`), gpurun_out/pmc_m (SQ / GRBM
pass of the same command), gpurun_out/gemm_shapes.txt, gpurun_out/bench_line.json"""
import collections, csv, json, os, re, shutil, subprocess, sys
TAG = sys.argv[1] if len(sys.argv) > 1 else "r06"
NSTEP = int(sys.argv[2]) if len(sys.argv) > 2 else 23          # bench.py --steps 10 --warmup 3: 3 + 10 (headline) + 10 (roofline leg)
NPMC = int(sys.argv[3]) if len(sys.argv) > 3 else 7            # bench.py --steps 3 --warmup 1: 1 + 3 + 3
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/"
G = R + "gpurun_out/"
P = R + "profiles/" + TAG + "_"
MATRIX = ("gemm_kernel", "attn_kernel<", "conv3x3_kernel", "conv3x3_wgrad_kernel")          # kernels that run on the matrix cores


def short(name):
    """kernel name without namespace / argument list"""
    return re.sub(r"^void |vbg::|\(.*$", "", name)


def is_roofline_launch(name, grid):
    """the launches bench.py's `roofline` times: every UNGROUPED dense NT product (plane_gemm NT, gemm_kernel DENSE_K x DENSE_K in the
    split form)"""
    if "plane_gemm_kernel" in name:          # <BM, BN, waves M, waves N, stages, TN, ping-pong, form>: the NT launches
        return re.search(r"plane_gemm_kernel<\d+, \d+, \d+, \d+, \d+, false", name) is not None
    m = re.search(r"gemm_kernel<(\d+), (\d+), (\d+), 256, (\d), (\d), (true|false), (\d)>", name)
    return bool(m) and m.group(4) == "0" and m.group(5) == "0" and not (m.group(7) == "0" and m.group(6) == "true")


def agg(path):
    a = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(path)):
        if not any(k in r["Kernel_Name"] for k in MATRIX):
            continue
        k = (short(r["Kernel_Name"]), r["Grid_Size"])
        a[k][0] += 1; a[k][1] += float(r["Counter_Value"]); a[k][2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return a


# ---- HBM-side traffic of the matrix kernels ---------------------------------------------------------------------------------------
if os.path.exists(G + "pmc_f/f_counter_collection.csv") and os.path.exists(G + "pmc_w/w_counter_collection.csv"):
    f, w = agg(G + "pmc_f/f_counter_collection.csv"), agg(G + "pmc_w/w_counter_collection.csv")
    lines = ["# rocprofv3 --kernel-trace --pmc FETCH_SIZE  /  --pmc WRITE_SIZE  (two separate passes) over `python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-amp-leg --no-h2d-leg`",
             "# per-launch averages for every matrix-core kernel instantiation x grid.  Units: rocprofv3 reports KB; FETCH_SIZE is DOUBLED per",
             "# MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE is taken as is (it reproduces known output sizes exactly).",
             "# FETCH counts L2 misses (Infinity-Cache hits included), not DRAM reads.",
             "# columns: kernel | grid threads | launches/step | fetch MB | write MB | avg us (under the profiler)"]
    nf = nw = n = 0
    for k in sorted(f, key=lambda k: -f[k][1]):
        fm = 2 * f[k][1] / f[k][0] * 1024 / 1e6
        wm = w[k][1] / w[k][0] * 1024 / 1e6 if k in w else float("nan")
        lines.append(f"{k[0]:58s} grid {int(k[1]):>8d}  x{f[k][0] / NPMC:6.1f}  fetch {fm:8.1f} MB  write {wm:8.1f} MB  {f[k][2] / f[k][0]:8.1f} us")
        if is_roofline_launch("void vbg::" + k[0] + ("(vbg_plane_gemm_desc)" if "plane" in k[0] else "(vbg_gemm_desc)"), k[1]):
            nf += 2 * f[k][1] * 1024; nw += w[k][1] * 1024 if k in w else 0; n += f[k][0]
    # the row-reuse 3x3 convolution launches (forward + input gradient): bench.py's primary `roofline` since round 4
    cf = cw = cn = 0
    for k in f:
        if k[0].startswith("conv3x3_kernel<"):
            cf += 2 * f[k][1] * 1024; cw += w[k][1] * 1024 if k in w else 0; cn += f[k][0]
    if cn:
        json.dump({"kernel": "conv3x3_kernel<*,*,*,*> (every row-reuse 3x3 convolution launch of one bench step, forward + input gradient: the launches bench.py's `roofline` times)",
                   "launches_per_step": cn / NPMC, "fetch_bytes_per_launch": cf / cn, "write_bytes_per_launch": cw / cn, "bytes_per_launch": (cf + cw) / cn,
                   "source": "profiles/" + TAG + "_gemm_hbm_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE x2 gfx950 correction; L2 misses incl. Infinity-Cache hits)"},
                  open(P + "conv3_hbm_traffic.json", "w"), indent=1)
        lines.append(f"# row-reuse 3x3 convolution launches (forward + input gradient): {cn / NPMC:.0f} launches/step, fetch {cf / cn / 1e6:.1f} MB + write {cw / cn / 1e6:.1f} MB per launch")
    tot_f = sum(2 * v[1] * 1024 for v in f.values()) / NPMC
    tot_w = sum(v[1] * 1024 for v in w.values()) / NPMC
    lines.append(f"# all matrix-core launches: fetch {tot_f / 1e9:.1f} GB + write {tot_w / 1e9:.1f} GB per step (round 1: 45.1 + 12.6 GB; before the XCD-aware block->tile map 112.7 GB fetched)")
    if n:
        lines.append(f"# ungrouped dense NT products (the launches bench.py's roofline times): {n / NPMC:.0f} launches/step, fetch {nf / n / 1e6:.1f} MB + write {nw / n / 1e6:.1f} MB per launch")
        json.dump({"kernel": "plane_gemm_kernel<*,*,*,*,*,false> + gemm_kernel<*,*,*,DENSE_K,DENSE_K,*,3> (every ungrouped dense NT launch of one bench step: the launches bench.py times)",
                   "launches_per_step": n / NPMC, "fetch_bytes_per_launch": nf / n, "write_bytes_per_launch": nw / n, "bytes_per_launch": (nf + nw) / n,
                   "source": "profiles/" + TAG + "_gemm_hbm_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE x2 gfx950 correction; L2 misses incl. Infinity-Cache hits)"},
                  open(P + "gemm_hbm_traffic.json", "w"), indent=1)
    open(P + "gemm_hbm_traffic.txt", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[-2:]))

shutil.copy(G + "prof_e/e_kernel_stats.csv", P + "bench_kernel_stats.csv")
shutil.copy(G + "prof_e/e_domain_stats.csv", P + "bench_domain_stats.csv")
for src, dst in (("gemm_shapes.txt", "gemm_shapes.txt"), ("bench_line.json", "bench_line.json"), ("plane_gemm_shapes.txt", "plane_gemm_shapes.txt"),
                 ("gemm_shapes_amp.txt", "gemm_shapes_amp.txt"), ("conv3_shapes.txt", "conv3_shapes.txt"),
                 ("conv3_pw.txt", "conv3_pw.txt"), ("step_conv3.txt", "step_conv3.txt"), ("conv3_forms.txt", "conv3_forms.txt"),
                 ("attn_shapes.txt", "attn_shapes.txt"), ("step_plane.txt", "step_plane.txt"), ("plane_pair_shapes.txt", "plane_pair_shapes.txt"),
                 ("bench_amp_line.json", "bench_amp_line.json"), ("infer_latency.txt", "infer_latency.txt"),
                 ("stock_loop_phases.txt", "stock_loop_phases.txt"), ("forced_reducer.txt", "forced_reducer.txt"),
                 ("stream_race.txt", "stream_race.txt")):
    if os.path.exists(G + src):
        shutil.copy(G + src, P + dst)
if os.path.exists(G + "prof_amp/amp_kernel_stats.csv"):
    shutil.copy(G + "prof_amp/amp_kernel_stats.csv", P + "amp_kernel_stats.csv")

# ---- MFMA utilisation per instantiation ----------------------------------------------------------------------------------------------
if os.path.exists(G + "pmc_m/m_counter_collection.csv"):
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(G + "pmc_m/m_counter_collection.csv")):
        if not any(k in r["Kernel_Name"] for k in MATRIX):
            continue
        d = disp.setdefault(r["Dispatch_Id"], {"name": short(r["Kernel_Name"]), "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
    ag = collections.defaultdict(lambda: collections.defaultdict(float))
    ctrs = ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE")
    for d in disp.values():
        a_ = ag[d["name"]]; a_["n"] += 1; a_["ns"] += d["ns"]
        for c in ctrs:
            a_[c] += d.get(c, 0.0)
    out = ["# rocprofv3 --kernel-trace --pmc " + " ".join(ctrs),
           f"# over `python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-amp-leg --no-h2d-leg`, summed per matrix-core kernel instantiation over the {NPMC} steps.",
           "# mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (kernel time * clock * 1024 SIMDs), clock = GRBM_GUI_ACTIVE / time / 8 XCDs;",
           "# wait/active columns are fractions of SQ_WAVE_CYCLES.  gemm_kernel<BM, BN, BK, 256, A-kind, B-kind, vec, form>: kinds 0 DENSE_K, 1 DENSE_R, 2 CONV_K,",
           "# 3 CONV_R, 4 WT_R; form 0 = fp32 MFMA, 3 = in-kernel bf16x3 split.  plane_gemm_kernel<BM, BN, waves M, waves N, stages, TN, ping-pong, form>: pre-split planes,",
           "# form 0 = three bf16 planes / six piece products, 1 = two fp16 planes / three piece products (mfma_util counts BUSY cycles: half the products at equal time halve it).",
           "# attn_kernel<mode, dropout>: 0 forward, 1 dQ, 2 dK/dV.  conv3x3_kernel<pixels per tile, filters per tile, fp16 form> / conv3x3_wgrad_kernel<waves along the filters, fp16 form, 7x7 region maps>: csrc/conv3.hip.",
           "kernel,launches_per_step,ms_per_step,clock_GHz,mfma_util,wait_any,wait_inst_any,active_inst_any,lds_bank_conflict_Mcycles"]
    tb = tc = 0
    for k, a_ in sorted(ag.items(), key=lambda kv: -kv[1]["ns"]):
        clk = a_["GRBM_GUI_ACTIVE"] / a_["ns"] / 8
        cap = a_["ns"] * clk * 1024
        tb += a_["SQ_VALU_MFMA_BUSY_CYCLES"]; tc += cap
        wc = a_["SQ_WAVE_CYCLES"] or 1
        out.append(f"\"{k}\",{a_['n'] / NPMC:.1f},{a_['ns'] / NPMC / 1e6:.3f},{clk:.2f},{a_['SQ_VALU_MFMA_BUSY_CYCLES'] / cap:.3f},"
                   f"{a_['SQ_WAIT_ANY'] / wc:.3f},{a_['SQ_WAIT_INST_ANY'] / wc:.3f},{a_['SQ_ACTIVE_INST_ANY'] / wc:.3f},{a_['SQ_LDS_BANK_CONFLICT'] / 1e6:.0f}")
    out.append(f"# all matrix-core kernels of the step: MFMA pipe busy {tb / tc:.3f} of the time they run")
    open(P + "gemm_mfma_util.txt", "w").write("\n".join(out) + "\n")
    print(out[-1])

# ---- streaming kernels ------------------------------------------------------------------------------------------------------------------
if os.path.exists(G + "prof_e/e_kernel_trace.csv"):
    tab = subprocess.run([sys.executable, R + "tools/hbm_table.py", G + "prof_e/e_kernel_trace.csv", str(NSTEP)], capture_output=True, text=True).stdout
    open(P + "hbm_kernels.txt", "w").write(f"# tools/hbm_table.py over the kernel trace of `bench.py --steps 10 --warmup 3 --no-h2d-leg` ({NSTEP} steps): achieved HBM-side bandwidth of the\n"
                                          "# streaming kernels = algorithmic bytes per step (cfg2 shapes, every tensor read / written once) / kernel time per step\n" + tab)
    print(tab)

# ---- where the step goes ----------------------------------------------------------------------------------------------------------------
rows = list(csv.DictReader(open(P + "bench_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / NSTEP / 1e6
cat = collections.OrderedDict()
for r in rows:
    nme, v = r["Name"], float(r["TotalDurationNs"]) / NSTEP / 1e6
    if "plane_gemm_kernel" in nme:
        tn = re.search(r"plane_gemm_kernel<\d+, \d+, \d+, \d+, \d+, true", nme) is not None
        pair = re.search(r", 1>\(", nme) is not None
        key = "plane GEMM TN (dense wgrad)" if tn else ("plane GEMM NT, fp16-pair form (QKV / FFN1 / FFN2 forward, data gradients)" if pair else "plane GEMM NT, bf16x3 form (attention out)")
    elif "conv3x3_wgrad" in nme or "conv3_wgrad_reduce" in nme: key = "conv wgrad, row-reuse kernel (conv3.hip)"
    elif "conv3x3_kernel" in nme or "conv3_wflip" in nme or "conv3_wprep" in nme: key = "conv fwd + dgrad, row-reuse kernel (conv3.hip)"
    elif "gemm_kernel" in nme:
        m = re.search(r"256, (\d), (\d)", nme)
        key = {("0", "0"): "dense NT in-kernel split (1x1, heads, stem)", ("0", "1"): "dense NN (dgrad)", ("1", "1"): "dense TN (wgrad)", ("2", "0"): "conv fwd",
               ("2", "4"): "conv dgrad", ("1", "3"): "conv wgrad"}[m.groups()]
    elif "attn_" in nme: key = "fused attention"
    elif "split_planes" in nme: key = "plane splits"
    elif "bn_" in nme: key = "BatchNorm"
    elif "ln_" in nme or "softmax" in nme or "gelu" in nme: key = "LayerNorm / GELU"
    elif "adamw" in nme or "sgd" in nme: key = "optimizers"
    elif "at::native" in nme or "rocclr" in nme: key = "torch plumbing"
    else: key = "other vbg kernels"
    cat[key] = cat.get(key, 0) + v
summary = [f"kernel ms/step {tot:.2f}"] + [f"{k:44s} {v:6.2f} ms {100 * v / tot:5.1f} %" for k, v in sorted(cat.items(), key=lambda kv: -kv[1])]
open(P + "step_breakdown.txt", "w").write("# kernel time per step by category, from " + TAG + "_bench_kernel_stats.csv (rocprofv3 --kernel-trace --stats)\n" + "\n".join(summary) + "\n")
print("\n".join(summary))
# ---- agreement of bench.py's live event timing with the kernel trace (generated, never hand-written: VERDICT r3) ------------------------
line = json.loads(open(P + "bench_line.json").read().strip().splitlines()[-1])
agr = ["# agreement check of bench.py's roofline objects with rocprofv3's kernel trace (tools/make_profiles.py; all runs on one box).",
       "# `live` = bench.py's event timing in the committed bench line (run WITHOUT the profiler); `live, traced run` = the same timing in the",
       "# very run the trace is of (its JSON line, gpurun_out/prof_e.log).  Since the two-stream defaults (end of round 5) the launches of both",
       "# families share the chip with launches of other streams.  The event pair brackets the DISPATCH (start / stop events in the packet):",
       "# with several queues active it includes what the dispatch waits for workgroup slots held by the other streams' kernels; the trace's",
       "# begin / end timestamps are the kernel's own.  One stream (rounds 1-4, and `alone on the chip` below): the two agreed within 1-2 %;",
       "# now the event time is ~10 % above the trace in the SAME run.  bench.py reports the event time (the longer, less flattering one)."]
prof_line = {}
try:
    prof_line = next(json.loads(l) for l in open(G + "prof_e.log") if l.startswith("{"))
except Exception:
    pass
for fam, sel, key in (("row-reuse 3x3 convolutions (conv3x3_kernel, forward + input gradient)", lambda n: "conv3x3_kernel<" in n, "roofline"),
                      ("ungrouped dense NT products (plane_gemm NT + gemm_kernel DENSE_K x DENSE_K)", lambda n: is_roofline_launch(n, None), "roofline_nt")):
    tw = tn = 0.0
    agr.append(f"{fam}:")
    for r in rows:
        if sel(r["Name"]):
            agr.append(f"    {short(r['Name'])[:84]:84s} x{int(r['Calls']) / NSTEP:6.1f} / step   {float(r['AverageNs']) / 1e3:8.1f} us")
            tw += float(r["TotalDurationNs"]); tn += int(r["Calls"])
    # (since round 5 `roofline` is whichever family takes more of the step; the other one rides as roofline_conv3 / roofline_nt)
    want = "conv3x3" if key == "roofline" else "plane_gemm"
    live = next((line[k] for k in ("roofline", "roofline_conv3", "roofline_nt") if k in line and want in line[k].get("kernel", "")), {})
    if tn:
        lp = next((prof_line[k] for k in ("roofline", "roofline_conv3", "roofline_nt") if k in prof_line and want in prof_line[k].get("kernel", "")), {})
        agr.append(f"    kernel trace: {tn / NSTEP:.0f} launches per step, average {tw / tn / 1e3:.2f} us;  bench.py live, traced run: average {lp.get('avg_us')} us "
                   f"(step {prof_line.get('ms_per_step')} ms);  bench.py live ({key}, unprofiled: step {line.get('ms_per_step')} ms): "
                   f"{live.get('launches', 0) / max(line.get('steps', 1), 1):.0f} launches per step, average {live.get('avg_us')} us"
                   + (f", alone on the chip {live['single_stream']['avg_us']} us" if 'single_stream' in live else ""))
# (bench.py reads this file and emits `roofline*.frac_trace`: the same fraction with the trace's average launch duration in place of the event pairs')
tr = {}
for fam, sel in (("conv3", lambda n: "conv3x3_kernel<" in n), ("nt", lambda n: is_roofline_launch(n, None))):
    tw = sum(float(r["TotalDurationNs"]) for r in rows if sel(r["Name"])); tn = sum(int(r["Calls"]) for r in rows if sel(r["Name"]))
    if tn:
        tr[fam] = {"avg_us": tw / tn / 1e3, "launches_per_step": tn / NSTEP}
json.dump(dict(tr, source="profiles/" + TAG + "_bench_kernel_stats.csv (rocprofv3 --kernel-trace --stats of `bench.py --steps 10 --warmup 3`, streams as run)"),
          open(P + "trace_avg.json", "w"), indent=1)
open(P + "agreement.txt", "w").write("\n".join(agr) + "\n")
print("\n".join(agr))
print(open(P + "bench_line.json").read()[:1500])
