#!/usr/bin/env python3
"""A/B of the round-6 ring guard of conv3x3_kernel's pipelined loop: tests/test_gpu_streams.py::_ring_contention_mismatches with the
guard as built (expect 0 mismatching launches) -- run with VBG_DEBUG_CONV3_NO_RING_GUARD=1 to leave the barrier out (expect > 0)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vibertgrid-pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import test_gpu_streams as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
print(f"VBG_DEBUG_CONV3_NO_RING_GUARD={os.environ.get('VBG_DEBUG_CONV3_NO_RING_GUARD', '0')}: {T._ring_contention_mismatches(n)} of {n} contended launches differ from the uncontended result")
