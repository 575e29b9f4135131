"""Random shapes through the batched path at the widths k_qreg_filter takes (384 / 512 / 768; float32 rows and the bfloat16 copy), against the
exact scan: row counts around tile and grid boundaries, 130-900 queries (1-4 workgroups per row walk, three included: the placement
falls back to the plain mapping), tombstones, k from 1 to 128.  python tools/dev_qreg_fuzz.py [cases=40] [seed=1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import quiver_amd as q
from tests import _oracle as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for c in range(cases):
    dim = int(rng.choice([384, 512, 768]))
    metric = str(rng.choice(["cosine", "dot_product", "euclidean", "squared_euclidean"]))
    n = int(rng.choice([32768, 32769, 40000, 64 * 777 + 1, 64 * 1024, 64 * 1025 - 1, 100_003, 256 * 64 * 2 + 63]))
    nq = int(rng.choice([130, 256, 257, 300, 512, 600, 768, 900]))
    k = int(rng.choice([1, 7, 10, 33, 64, 65, 100, 128]))
    bf = bool(rng.integers(0, 2))
    idx = q.DeviceIndex(dim, metric, bf16_rows=bf)
    idx.add_synthetic(1000 + c, 0, n)
    qs = O.gen_rows(2000 + c, 0, nq, dim)
    dead = rng.choice(n, size=int(rng.integers(0, 200)), replace=False).astype(np.uint32)
    if len(dead):
        idx.remove(dead)
    exact = [np.concatenate([idx.search(qs[i:i + 8], k)[j] for i in range(0, nq, 8)]) for j in range(3)]
    got = idx.search(qs, k, batched=True)
    same = np.array_equal(exact[0], got[0]) and np.array_equal(exact[1].view(np.uint32), got[1].view(np.uint32)) and np.array_equal(exact[2], got[2])
    bad += not same
    print("case %2d: %-17s dim %3d rows %6d queries %3d k %3d copy %d removed %3d: %s" % (c, metric, dim, n, nq, k, bf, len(dead), "same" if same else "DIFFERENT"), flush=True)
    idx.close()
print("%d of %d cases differ" % (bad, cases))
sys.exit(1 if bad else 0)
