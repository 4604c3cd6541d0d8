#!/usr/bin/env python3
"""Host-side profile of the bench step (cProfile over a few steps, amp on so that the GPU is not the bottleneck):
    python tools/host_profile.py [--fp32]"""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--steps", "12", "--warmup", "4", "--no-cpu-baseline", "--no-amp-leg", "--no-h2d-leg"] + ([] if "--fp32" in sys.argv else ["--amp"])
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
st.sort_stats("tottime").print_stats("vibertgrid|bench.py|ctypes|numpy|torch/autograd/function", 40)
print(s.getvalue()[:12000])
