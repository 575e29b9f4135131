"""qv_sharded_* — the multi-GPU flat scan behind the C ABI (SURVEY.md 8e, 8b's device_count): one handle, a shard per device,
ONE RCCL all-gather of the per-shard top-k, merge on the first device.  On the 1-GPU box the RCCL path runs with a single
shard (a real ncclCommInitAll + ncclAllGather over one rank), and the multi-shard bookkeeping (bases, placement, merge,
empty shards, removes) runs with several shards CO-LOCATED on device 0 through the point-to-point exchange mode.  The
oracle is the checker: same rows, same (distance, global row) order, same float32 bits."""
import numpy as np
import pytest

import quiver_amd
from quiver_amd import ShardedIndex
from tests import _oracle as O

pytestmark = pytest.mark.gpu


def _check(idx, rows_by_global, qs, k, metric_id):
    gids = np.array(sorted(rows_by_global), dtype=np.uint32)
    mat = np.stack([rows_by_global[g] for g in gids])
    r, d, c = idx.search(qs, k)
    for i, q in enumerate(qs):
        er, ed = O.exact_search(metric_id, mat, q, k)
        assert c[i] == len(er)
        assert gids[er].tolist() == r[i, :c[i]].tolist(), i
        assert np.array_equal(ed.view(np.uint32), d[i, :c[i]].view(np.uint32)), i
        assert (r[i, c[i]:] == 0xFFFFFFFF).all() and np.isinf(d[i, c[i]:]).all()


@pytest.mark.parametrize("metric", ["cosine", "l2", "dot"])
def test_rccl_path_single_shard_equals_oracle(metric):
    """n = 1: the exchange is still a real RCCL all-gather (communicator over one device)"""
    mid = quiver_amd.metric_id(metric)
    rows = O.gen_rows(1001, 0, 5000, 96)
    idx = ShardedIndex(96, metric, devices=[0])
    gids = idx.add(rows)
    assert gids.tolist() == list(range(5000)) and idx.size() == 5000
    qs = O.gen_rows(1002, 0, 9, 96)
    _check(idx, dict(zip(gids.tolist(), rows)), qs, 10, mid)
    _check(idx, dict(zip(gids.tolist(), rows)), qs[:1], 64, mid)


@pytest.mark.parametrize("metric,n_shards", [("cosine", 3), ("l2", 4), ("cosine_f32", 2)])
def test_co_located_shards_equal_oracle(metric, n_shards):
    mid = quiver_amd.metric_id(metric)
    dim = 64
    idx = ShardedIndex(dim, metric, devices=[0] * n_shards, peer_copy=True)
    span = quiver_amd.lib().qv_sharded_span(n_shards)
    rows_by_g = {}
    rows = O.gen_rows(7, 0, 4001, dim)
    g1 = idx.add(rows[:3000]); rows_by_g.update(zip(g1.tolist(), rows[:3000]))            # one batch, cut over the shards
    g2 = idx.add(rows[3000:3001]); rows_by_g.update(zip(g2.tolist(), rows[3000:3001]))    # a single Insert
    g3 = idx.add(rows[3001:]); rows_by_g.update(zip(g3.tolist(), rows[3001:]))
    assert len(rows_by_g) == 4001 and idx.size() == 4001
    fills = [idx.shard_info(g)["rows"] for g in range(n_shards)]
    assert max(fills) - min(fills) <= 1
    for g in range(n_shards):
        assert idx.shard_info(g)["base"] == g * span
    assert all(int(gid) // span < n_shards for gid in rows_by_g)
    qs = O.gen_rows(8, 0, 12, dim)
    _check(idx, rows_by_g, qs, 10, mid)
    _check(idx, rows_by_g, qs[:1], 1, mid)
    # removes (tombstones in several shards)
    dead = list(rows_by_g)[::7]
    idx.remove(dead)
    for gid in dead:
        del rows_by_g[gid]
    assert idx.size() == len(rows_by_g)
    _check(idx, rows_by_g, qs, 10, mid)


def test_fewer_rows_than_shards_and_k_larger_than_corpus():
    idx = ShardedIndex(8, "l2", devices=[0, 0, 0, 0], peer_copy=True)
    r, d, c = idx.search(np.ones(8, np.float32), 5)
    assert c[0] == 0                                              # empty index: no results, no error (exact.go:96-98)
    rows = O.gen_rows(5, 0, 2, 8)
    gids = idx.add(rows)                                          # two shards stay empty
    _check(idx, dict(zip(gids.tolist(), rows)), O.gen_rows(6, 0, 3, 8), 5, 1)
    with pytest.raises(quiver_amd.QvError) as e:
        idx.search(rows[0], 0)
    assert "k must be positive" in str(e.value)
    r, d, c = idx.search(rows[0], 65)                            # any k: clamped to the corpus, the rest padded
    assert c[0] == 2 and (r[0, 2:] == 0xFFFFFFFF).all() and sorted(r[0, :2].tolist()) == sorted(gids.tolist())


def test_rccl_refuses_a_device_listed_twice():
    with pytest.raises(quiver_amd.QvError, match="listed twice"):
        ShardedIndex(8, "cosine", devices=[0, 0])


def test_device_pointer_form_and_synthetic_blocks_equal_single_index():
    """the shape bench.py times: synthetic corpus in contiguous blocks, queries / results resident on the first device"""
    import torch
    n, dim, k = 200_000, 128, 10
    sh = ShardedIndex(dim, "cosine", devices=[0, 0, 0], peer_copy=True)
    sh.add_synthetic(20260424, 0, n)
    one = quiver_amd.DeviceIndex(dim, "cosine")
    one.add_synthetic(20260424, 0, n)
    qs = O.gen_rows(20260425, 0, 16, dim)
    dq = torch.from_numpy(qs).cuda()
    dr = torch.empty((16, k), dtype=torch.int32, device="cuda"); dd = torch.empty((16, k), dtype=torch.float32, device="cuda")
    span = quiver_amd.lib().qv_sharded_span(3)
    bounds = [g * n // 3 for g in range(4)]
    for nq in (1, 16):
        sh.search_device(dq.data_ptr(), nq, k, dr.data_ptr(), dd.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        r1, d1, _ = one.search(qs[:nq], k)
        got = dr[:nq].cpu().numpy().view(np.uint32)
        back = np.array([[bounds[int(x) // span] + int(x) % span for x in row] for row in got], dtype=np.uint32)   # global id -> generator row
        assert np.array_equal(back, r1) and np.array_equal(dd[:nq].cpu().numpy().view(np.uint32), d1.view(np.uint32))
    sh.profile(True)
    sh.search(qs[:1], k)
    p = sh.profile_read()
    assert p["searches"] == 1 and p["scan_ms"] > 0


@pytest.mark.parametrize("nq", [40, 300])
def test_sharded_batches_take_the_filter_path_and_equal_the_exact_scan(nq, monkeypatch):
    """a batch of 9+ queries goes through qv_index_search_batched_device on every shard that qualifies (the batched
    MFMA path shards like the flat scan); the result equals the exact scan of one unsharded index"""
    n, dim, k = 1_200_000, 64, 10
    sh = ShardedIndex(dim, "cosine", devices=[0, 0, 0], peer_copy=True)
    sh.add_synthetic(20260424, 0, n)
    one = quiver_amd.DeviceIndex(dim, "cosine")
    one.add_synthetic(20260424, 0, n)
    qs = O.gen_rows(20260425, 0, nq, dim)
    one.set_filter("off")                                          # the reference: the exact multi-query scan
    r1, d1, _ = one.search(qs, k)
    r, d, c = sh.search(qs, k)
    span = quiver_amd.lib().qv_sharded_span(3)
    bounds = [g * n // 3 for g in range(4)]
    back = np.array([[bounds[int(x) // span] + int(x) % span for x in row] for row in r], dtype=np.uint32)
    assert (c == k).all() and np.array_equal(back, r1) and np.array_equal(d.view(np.uint32), d1.view(np.uint32))


def test_two_devices_over_rccl_equal_one_index():
    """the multi-DEVICE code paths (hipMemcpyPeerAsync between devices, cross-device events, a grouped ncclAllGather over two
    communicators): runs only where two GPUs are visible — the 1-GPU test box skips it"""
    if quiver_amd.lib().qv_device_count() < 2:
        pytest.skip("needs two GPUs")
    n, dim, k = 100_003, 96, 10
    for peer in (False, True):
        sh = ShardedIndex(dim, "cosine", devices=[0, 1], peer_copy=peer)
        sh.add_synthetic(20260424, 0, n)
        one = quiver_amd.DeviceIndex(dim, "cosine")
        one.add_synthetic(20260424, 0, n)
        qs = O.gen_rows(20260425, 0, 12, dim)
        span = quiver_amd.lib().qv_sharded_span(2)
        bounds = [0, n // 2, n]
        for kk in (k, 300):
            r, d, c = sh.search(qs, kk)
            r1, d1, _ = one.search(qs, kk)
            back = np.array([[bounds[int(x) // span] + int(x) % span for x in row] for row in r], dtype=np.uint32)
            assert np.array_equal(back, r1) and np.array_equal(d.view(np.uint32), d1.view(np.uint32)), (peer, kk)
        sh.close(); one.close()


def test_concurrent_callers_on_one_rccl_handle_over_every_visible_device():
    """ADVICE (round 3): include/qv.h promises concurrent searches on one handle.  On the RCCL exchange every call context drives
    the shared per-device communicators from streams of its own (only the group enqueue is serialised): eight threads of
    qv_sharded_search and qv_sharded_search_device on ONE handle over all visible GPUs against a single index.  Needs >= 2 GPUs
    (the 1-GPU test box skips it; with one rank the same test runs in test_gpu_sharded_index.py)."""
    import threading
    import torch
    ndev = quiver_amd.lib().qv_device_count()
    if ndev < 2:
        pytest.skip("needs two or more GPUs")
    n, dim, k = 200_003, 64, 10
    sh = ShardedIndex(dim, "cosine", devices=list(range(ndev)))            # RCCL exchange
    sh.add_synthetic(20260424, 0, n)
    one = quiver_amd.DeviceIndex(dim, "cosine")
    one.add_synthetic(20260424, 0, n)
    qs = O.gen_rows(20260425, 0, 32, dim)
    r1, d1, _ = one.search(qs, k)
    span = quiver_amd.lib().qv_sharded_span(ndev)
    bounds = [g * n // ndev for g in range(ndev + 1)]
    errs = []

    def back(r):
        return np.array([bounds[int(x) // span] + int(x) % span for x in r], dtype=np.uint32)

    def worker(t):
        try:
            torch.cuda.set_device(0)
            st = torch.cuda.Stream(device=0)
            dq = torch.from_numpy(qs).cuda()
            dr = torch.empty((1, k), dtype=torch.int32, device="cuda"); dd = torch.empty((1, k), dtype=torch.float32, device="cuda")
            for rep in range(40):
                i = (t * 7 + rep) % len(qs)
                if rep % 2:
                    r, d, _ = sh.search(qs[i], k)
                    r, d = r[0], d[0]
                else:
                    sh.search_device(dq[i].data_ptr(), 1, k, dr.data_ptr(), dd.data_ptr(), st.cuda_stream)
                    st.synchronize()
                    r, d = dr.cpu().numpy().view(np.uint32)[0], dd.cpu().numpy()[0]
                if not (np.array_equal(back(r), r1[i]) and np.array_equal(d.view(np.uint32), d1[i].view(np.uint32))):
                    errs.append((t, rep))
        except Exception as ex:  # noqa: BLE001
            errs.append(repr(ex))

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs[:5]
    sh.close(); one.close()
