"""More seeds of the short-corpus scan fuzz (tests/test_gpu_scan_split.py) and of the flat operation-sequence fuzz (tests/test_gpu_fuzz.py)
than the suite runs:  python tools/soak_scan_fuzz.py [first_seed] [n_seeds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_scan_split as F, test_gpu_fuzz as G
s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
t0 = time.time(); bad = 0
for s in range(s0, s0 + n):
    try:
        F.test_split_scan_equals_the_oracle(s)
    except AssertionError as e:
        bad += 1; print("split scan seed", s, "FAILED", str(e)[:200], flush=True)
for s in range(s0, s0 + n):
    try:
        G.test_random_operation_sequences_match_the_oracle(s)
    except AssertionError as e:
        bad += 1; print("operation sequences seed", s, "FAILED", str(e)[:200], flush=True)
print("seeds %d..%d: %d failures, %.0f s" % (s0, s0 + n - 1, bad, time.time() - t0), flush=True)
