"""Generates tests/golden/flat_10kx128_cosine.npz and tests/golden/hnsw_3kx32_cosine.npz from the repo's own CPU oracle
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

# ---- HNSW: a graph built by the oracle's Insert (hnsw.go:266-468) and the traversal results on it ------------------
# 3000 x 32 cosine, M=8, efConstruction=40, MaxLevel=4 (multi-level: upper-level descent included), seed 11;
# 40 queries, efSearch=64, k=10.  Stored: levels, per-level links (so the device graph can be rebuilt without the
# oracle), entry point, and the oracle's Search output incl. evaluation counts.
HN, HD, HM, HEFC, HEFS, HLV, HSEED, HNQ = 3000, 32, 8, 40, 64, 4, 11, 40
hrows = O.gen_rows(CORPUS_SEED, 0, HN, HD)
h = O.HNSW(0, HD, M=HM, efConstruction=HEFC, efSearch=HEFS, maxLevel=HLV, seed=HSEED)
for r in hrows:
    h.insert(r)
levels = np.array([h.node_level(n) for n in range(HN)], np.int8)
l0_deg = np.zeros(HN, np.uint32); l0_links = np.zeros((HN, 2 * HM), np.uint32)
up_off = np.zeros(HN, np.uint32); up_blocks = []
for n in range(HN):
    l = h.links(n, 0); l0_deg[n] = l.size; l0_links[n, :l.size] = l
    up_off[n] = len(up_blocks)
    for lv in range(1, int(levels[n]) + 1):
        l = h.links(n, lv); blk = np.zeros(1 + HM, np.uint32); blk[0] = l.size; blk[1:1 + l.size] = l
        up_blocks.append(blk)
up_links = np.stack(up_blocks) if up_blocks else np.zeros((1, 1 + HM), np.uint32)
ep, cur = h.entry_point()
hqs = O.gen_rows(QUERY_SEED, 0, HNQ, HD)
hr = np.full((HNQ, K), 0xFFFFFFFF, np.uint32); hd = np.full((HNQ, K), np.inf, np.float32); he = np.zeros(HNQ, np.uint64)
for i in range(HNQ):
    r, d, e = h.search(hqs[i], K, with_evals=True)
    hr[i, :r.size], hd[i, :d.size], he[i] = r, d, e
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "hnsw_3kx32_cosine.npz"),
                    corpus_seed=CORPUS_SEED, query_seed=QUERY_SEED, n=HN, dim=HD, M=HM, efConstruction=HEFC, efSearch=HEFS, maxLevel=HLV, seed=HSEED,
                    levels=levels, l0_deg=l0_deg, l0_links=l0_links, up_off=up_off, up_links=up_links, entry=ep, cur_level=cur,
                    rows=hr, dist=hd, evals=he)
print("wrote hnsw_3kx32_cosine.npz  entry", ep, "level", cur, "upper blocks", len(up_blocks), "first result", hr[0], hd[0])
