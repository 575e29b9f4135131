"""Randomised stress of the batched paths against the exact scans of the same index (filter off), bit for bit: larger corpora, more queries and
wider rows than the -m gpu suite affords.  python tools/stress_selection_path.py [seconds] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import quiver_amd

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
runs = bad = flagged_total = 0
while time.time() < t_end:
    metric = str(rng.choice(["cosine", "dot_product", "euclidean", "squared_euclidean"]))
    dim = int(rng.choice([32, 64, 96, 100, 128, 200, 256, 384, 768, 1000, 1536, 2048]))
    n = int(rng.choice([40_000, 131_072, 200_000, 500_000, 1_000_000]))
    if n * dim * 4 > 3.5e9: n = int(3.5e9 / (dim * 4))
    rowmajor, bf16 = bool(rng.integers(2)), bool(rng.integers(2))
    idx = quiver_amd.DeviceIndex(dim, metric, rowmajor=rowmajor, bf16_rows=bf16)
    style = int(rng.integers(3))
    if style == 0:
        idx.add_synthetic(int(rng.integers(1 << 30)), 0, n)
    else:                                                      # clustered rows stored cluster by cluster (style 2: tiny clusters), built on the device
        g = torch.Generator(device="cuda"); g.manual_seed(int(rng.integers(1 << 30)))
        per = 4096 if style == 1 else 192
        for s0 in range(0, n, 250_000):
            m = min(250_000, n - s0)
            c = torch.randn(((m + per - 1) // per, dim), generator=g, device="cuda").repeat_interleave(per, 0)[:m]
            x = (c + 0.03 * torch.randn((m, dim), generator=g, device="cuda")).contiguous()
            idx.add_device(x.data_ptr(), m, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    dead = int(rng.choice([0, 0, n // 50, n // 3]))
    if dead: idx.remove(rng.choice(n, dead, replace=False).astype(np.uint32))
    for _ in range(3):
        nq = int(rng.choice([9, 33, 64, 100, 256, 300, 700]))
        k = int(rng.choice([1, 10, 15, 16, 17, 40, 64, 65, 100, 129, 300, 1000, 2048, 2049, 4096]))
        qrow = rng.integers(0, n, size=nq)
        qs = np.stack([idx.get_row(int(r)) for r in qrow[:8]] + [rng.standard_normal(dim).astype(np.float32) for _ in range(nq - min(nq, 8))])[:nq].astype(np.float32)
        flt = str(rng.choice(["auto", "auto", "fp32", "bf16x3", "bf16x1"]))   # the index's choice of filter kernel (qv_index_set_filter)
        idx.set_filter(flt)
        got = idx.search(qs, k, batched=True)
        idx.set_filter("off"); want = idx.search(qs, k); idx.set_filter(quiver_amd.DeviceIndex.default_filter)
        ok = np.array_equal(got[0], want[0]) and got[1].tobytes() == want[1].tobytes() and np.array_equal(got[2], want[2])
        runs += 1
        if not ok:
            bad += 1
            print("MISMATCH metric=%s dim=%d n=%d rowmajor=%s bf16=%s style=%d dead=%d nq=%d k=%d filter=%s" % (metric, dim, n, rowmajor, bf16, style, dead, nq, k, flt), flush=True)
    idx.close()
print("%d batches, %d mismatches" % (runs, bad))
sys.exit(1 if bad else 0)
