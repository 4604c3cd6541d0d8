#!/usr/bin/env python3
"""print calls / average us of the kernels of a rocprofv3 kernel_stats.csv whose name matches a regular expression:
python tools/kstat.py <dir or csv> <regex>"""
import csv, glob, os, re, sys
src = sys.argv[1]
files = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)
rx = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
for f in files:
    for r in csv.DictReader(open(f)):
        if rx.search(r["Name"]):
            print(f"{int(r['Calls']):7d} calls  {float(r['AverageNs']) / 1e3:9.2f} us avg  {float(r['TotalDurationNs']) / 1e6:9.3f} ms  {r['Name'][:110]}")
