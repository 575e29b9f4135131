"""Small collections (BASELINE configs[0] is 10k x 128; the reference's own ExactIndex.Search bench is 1000 x 64,
final_bench.txt:28): up to 256 tiles and 16 results qv_index_search* runs scan + merge as ONE launch (k_flat_scan_small: the last
workgroup to finish merges the others' lists) and the host-pointer entry point polls a sequence number the kernel writes behind
its results instead of waiting for the stream.  Same rows, same bits as the oracle; back-to-back calls must never see the
previous call's results."""
import threading

import numpy as np
import pytest

from tests import _oracle as O

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("metric", range(9))
@pytest.mark.parametrize("n,dim", [(1, 3), (64, 16), (65, 7), (1000, 64), (10_000, 128), (16_384, 32), (16_385, 32)])
def test_small_collections_equal_the_oracle(metric, n, dim):
    import quiver_amd as q
    rng = np.random.default_rng(1000 * metric + n + dim)
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    if n > 10:
        rows[3] = 0.0; rows[5] = rows[4]
    idx = q.DeviceIndex(dim, metric)
    idx.add(rows)
    alive = None
    if n > 200:
        dead = rng.choice(n, n // 7, replace=False).astype(np.uint32)
        idx.remove(dead)
        alive = np.ones(n, bool); alive[dead] = False
    qs = rng.standard_normal((6, dim)).astype(np.float32)
    qs[1] = rows[min(4, n - 1)]
    for k in (1, 10, 16, 17):
        for i in range(6):                                            # one query per call: the polled path; consecutive calls must not see stale results
            r, d, c = idx.search(qs[i], k)
            er, ed = O.exact_search(metric, rows, qs[i], k, alive=alive)
            assert int(c[0]) == len(er) and np.array_equal(r[0, :len(er)], er) and np.array_equal(_bits(d[0, :len(er)]), _bits(ed)), (metric, n, k, i)
        r, d, c = idx.search(qs[:3], k)                               # up to four queries share the launch (no polling)
        for i in range(3):
            er, ed = O.exact_search(metric, rows, qs[i], k, alive=alive)
            assert np.array_equal(r[i, :len(er)], er) and np.array_equal(_bits(d[i, :len(er)]), _bits(ed))


def test_small_collection_many_callers_and_device_entry():
    import torch
    import quiver_amd as q
    n, dim, k = 10_000, 128, 10
    rows = O.gen_rows(20260424, 0, n, dim)
    idx = q.DeviceIndex(dim, "cosine")
    idx.add(rows)
    qs = O.gen_rows(20260425, 0, 64, dim)
    want = [O.exact_search(0, rows, x, k) for x in qs]
    errs = []

    def worker(t):
        try:
            for rep in range(300):
                i = (t * 11 + rep) % 64
                r, d, _ = idx.search(qs[i], k)
                if not (np.array_equal(r[0], want[i][0]) and np.array_equal(_bits(d[0]), _bits(want[i][1]))):
                    errs.append((t, rep))
        except Exception as ex:  # noqa: BLE001
            errs.append(repr(ex))

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not errs, errs[:5]
    dq = torch.from_numpy(qs).cuda()
    dr = torch.empty((64, k), dtype=torch.int32, device="cuda"); dd = torch.empty((64, k), dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for i in range(64):                                               # 64 single-query launches back to back on one stream: the tickets are left zero each time
        idx.search_device(dq[i].data_ptr(), 1, k, dr[i].data_ptr(), dd[i].data_ptr(), s)
    torch.cuda.synchronize()
    for i in range(64):
        assert np.array_equal(dr[i].cpu().numpy().view(np.uint32), want[i][0]) and np.array_equal(_bits(dd[i].cpu().numpy()), _bits(want[i][1])), i
