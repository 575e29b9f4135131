"""Generates tests/golden/flat_10kx128_cosine.npz from the repo's own CPU oracle
(after it has passed the reference's KATs, tests/test_oracle_kats.py).

    python tests/golden/make_golden_vectors.py

BASELINE.json configs[0]: pkg/hybrid exact flat scan, 10k x 128 fp32 cosine, k=10.
Corpus/query seeds follow SURVEY.md 8d (20260424 / 20260425)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import _oracle as O  # noqa: E402

CORPUS_SEED, QUERY_SEED, N, D, K, NQ = 20260424, 20260425, 10000, 128, 10, 32

rows = O.gen_rows(CORPUS_SEED, 0, N, D)
qs = O.gen_rows(QUERY_SEED, 0, NQ, D)
out_r = np.empty((NQ, K), np.uint32)
out_d = np.empty((NQ, K), np.float32)
for i in range(NQ):
    out_r[i], out_d[i] = O.exact_search(0, rows, qs[i], K)
np.savez(os.path.join(ROOT, "tests", "golden", "flat_10kx128_cosine.npz"),
         corpus_seed=CORPUS_SEED, query_seed=QUERY_SEED, rows=out_r, dist=out_d)
print("wrote flat_10kx128_cosine.npz", out_r[0], out_d[0])
