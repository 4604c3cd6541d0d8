"""compare the per-parameter gradient errors of two full-scale runs (tests/test_gpu_full_scale.py with VBG_DUMP_DIR set)"""
import json, sys
a = json.load(open(sys.argv[1]))["errs"]; b = json.load(open(sys.argv[2]))["errs"]
rows = sorted(a, key=lambda k: -max(a[k], b[k]))[:25]
for k in rows:
    print(f"{a[k]:.2e} {b[k]:.2e}  {k}")
import statistics
qa = [a[k] for k in a if "attention.self" in k and "weight" in k]; qb = [b[k] for k in a if "attention.self" in k and "weight" in k]
print("attention.self weights: median", statistics.median(qa), statistics.median(qb), "max", max(qa), max(qb))
