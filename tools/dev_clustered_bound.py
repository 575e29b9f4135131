"""How many queries of a batch the filter + re-score path hands back (candidate overflow -> exact scan) on corpora stored cluster after
cluster: 30 x 10 000, 300 x 1000, 3000 x 100 rows at 768 dimensions, 30 x 10 000 at 128.  The bound of a query's k-th distance comes from a
sample; how it is selected (QV_MFMA_SAMPLE_GROUP_MIN) decides whether such corpora fall off the fast path.  python tools/dev_clustered_bound.py"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, quiver_amd as q
rng = np.random.default_rng(11)
for dim, n_clusters, per, nq, k in ((768, 30, 10_000, 256, 10), (768, 30, 10_000, 256, 64), (768, 300, 1_000, 256, 10), (768, 3000, 100, 256, 10), (128, 30, 10_000, 256, 10)):
    centres = rng.standard_normal((n_clusters, dim))
    rows = np.concatenate([c + 0.3 * rng.standard_normal((per, dim)) for c in centres]).astype(np.float32)
    qs = (centres[rng.integers(0, n_clusters, nq)] + 0.3 * rng.standard_normal((nq, dim))).astype(np.float32)
    idx = q.DeviceIndex(dim, "cosine"); idx.add(rows)
    dq = torch.from_numpy(qs).cuda()
    dr = torch.empty((nq, k), dtype=torch.int32, device="cuda"); dd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    fl = torch.full((nq,), 7, dtype=torch.int32, device="cuda")
    idx.search_batched_device(dq.data_ptr(), nq, k, dr.data_ptr(), dd.data_ptr(), fl.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    print("dim %d clusters %d x %d, nq %d k %d: handed back %d of %d (QV_MFMA_SAMPLE_GROUP_MIN=%s)" % (dim, n_clusters, per, nq, k, int(fl.abs().sum().item()), nq, os.environ.get("QV_MFMA_SAMPLE_GROUP_MIN", "1")), flush=True)
