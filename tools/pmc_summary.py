#!/usr/bin/env python3
"""sum the counters of every dispatch whose kernel name contains <pattern>: python tools/pmc_summary.py pattern dir/file_counter_collection.csv ..."""
import collections, csv, sys
pat = sys.argv[1]
for path in sys.argv[2:]:
    agg = collections.defaultdict(float); seen = set(); ns = 0
    for r in csv.DictReader(open(path)):
        if pat not in r['Kernel_Name']:
            continue
        agg[r['Counter_Name']] += float(r['Counter_Value'])
        if r['Dispatch_Id'] not in seen:
            seen.add(r['Dispatch_Id']); ns += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    n = max(len(seen), 1)
    print(path, 'dispatches', len(seen), 'avg us %.1f' % (ns / n / 1e3))
    for k, v in sorted(agg.items()):
        print('   %-40s %.4g per dispatch' % (k, v / n))
