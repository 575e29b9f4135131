"""qv_sharded_* as a full core.Index (pkg/core/collection.go:78-96) over several shards: any k (a filtered Collection.Search
asks for k = Index.Size(), collection.go:679-682), filtered search, search with a negative example (hybrid_index.go:517-570),
listed-row distances, update, get — each compared bit for bit with the CPU oracle and with ONE unsharded qv_index holding
the same rows.  Shards are co-located on device 0 (point-to-point exchange) except the 1-shard case, which runs RCCL."""
import threading

import numpy as np
import pytest

import quiver_amd
from quiver_amd import DeviceIndex, ShardedIndex
from tests import _oracle as O

pytestmark = pytest.mark.gpu

SHARDS = [1, 3, 8]


def _make(n_shards, dim, metric, **kw):
    if n_shards == 1:
        return ShardedIndex(dim, metric, devices=[0], **kw)                       # the RCCL exchange, one rank
    return ShardedIndex(dim, metric, devices=[0] * n_shards, peer_copy=True, **kw)


def _oracle_topk(mid, mat, gids, q, k):
    er, ed = O.exact_search(mid, mat, q, min(k, len(mat)))
    return gids[er], ed


@pytest.mark.parametrize("n_shards", SHARDS)
@pytest.mark.parametrize("metric", ["cosine", "l2sq", "dot_f32"])
def test_any_k_equals_oracle(n_shards, metric):
    mid = quiver_amd.metric_id(metric)
    dim, n = 48, 3001
    rows = O.gen_rows(31, 0, n, dim)
    rows[100] = rows[7]; rows[2000] = rows[7]                                      # equal distances across shards: ties go by global row
    sh = _make(n_shards, dim, metric)
    gids = sh.add(rows)
    order = np.argsort(gids, kind="stable")
    sg, mat = gids[order], rows[order]                                             # ascending global row == the oracle's row order
    qs = O.gen_rows(32, 0, 3, dim)
    for k in (65, 100, 1500, n, n + 50):
        r, d, c = sh.search(qs, k)
        for i, q in enumerate(qs):
            er, ed = _oracle_topk(mid, mat, sg, q, k)
            assert c[i] == len(er) == min(k, n)
            assert np.array_equal(r[i, :c[i]], er), (k, i)
            assert np.array_equal(d[i, :c[i]].view(np.uint32), ed.view(np.uint32)), (k, i)
            assert (r[i, c[i]:] == 0xFFFFFFFF).all() and np.isinf(d[i, c[i]:]).all()
    # tombstones in several shards, then the full ranking again
    dead = gids[::5]
    sh.remove(dead)
    keep = ~np.isin(sg, dead)
    r, d, c = sh.search(qs[:1], n)
    er, ed = _oracle_topk(mid, mat[keep], sg[keep], qs[0], n)
    assert c[0] == keep.sum() and np.array_equal(r[0, :c[0]], er) and np.array_equal(d[0, :c[0]].view(np.uint32), ed.view(np.uint32))
    sh.close()


@pytest.mark.parametrize("k", [40, 100])
def test_batches_on_shards_large_enough_for_a_guessed_bound(k):
    """three co-located shards of 150 000 rows: each shard's batch takes the selection path with a guessed bound (16 or more results per query
    over 131 072 rows or more); half of the rows sit in clusters of 256 stored one after the other, so some guesses fail and their queries come
    back flagged — redone on the device (k <= 64) or from the host's flag read (k > 64).  Equal to one index's exact scan."""
    dim, per, nq = 64, 150_000, 48
    rng = np.random.default_rng(17)
    centers = rng.standard_normal((per * 3 // 2 // 256, dim)).astype(np.float32)
    clustered = np.concatenate([c + 0.02 * rng.standard_normal((256, dim)).astype(np.float32) for c in centers])
    rows = np.concatenate([clustered, O.gen_rows(61, 0, 3 * per - len(clustered), dim)])
    one = quiver_amd.DeviceIndex(dim, "cosine", filter="off")
    one.add(rows)
    qs = np.concatenate([centers[0:32:2], centers[1:33:2], O.gen_rows(62, 0, nq - 32, dim)]).astype(np.float32)
    er, ed, _ = one.search(qs, k)
    sh = quiver_amd.ShardedIndex(dim, "cosine", devices=[0, 0, 0], peer_copy=True)
    gids = sh.add(rows)
    r, d, c = sh.search(qs, k)
    pos = {int(g): i for i, g in enumerate(gids)}
    back = np.vectorize(lambda g: pos[int(g)])(r)                                 # global row ids back to positions in `rows`
    assert (c == k).all()
    assert np.array_equal(back, er.astype(back.dtype)) and np.array_equal(d.view(np.uint32), ed.view(np.uint32))
    # the device-pointer call: no host synchronisation at either k (the hand-backs are listed and redone on the device)
    import torch
    dq = torch.from_numpy(qs).cuda()
    dr = torch.empty((nq, k), dtype=torch.int32, device="cuda"); dd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    sh.search_device(dq.data_ptr(), nq, k, dr.data_ptr(), dd.data_ptr(), torch.cuda.current_stream().cuda_stream)
    sh.sync(); torch.cuda.synchronize()
    assert np.array_equal(dr.cpu().numpy().view(np.uint32), r) and np.array_equal(dd.cpu().numpy().view(np.uint32), d.view(np.uint32))
    sh.close(); one.close()


@pytest.mark.parametrize("n_shards", SHARDS)
def test_masked_search_equals_oracle_over_the_selected_rows(n_shards):
    mid, dim, n = 0, 40, 2500
    rows = O.gen_rows(41, 0, n, dim)
    sh = _make(n_shards, dim, "cosine")
    gids = sh.add(rows)
    order = np.argsort(gids, kind="stable")
    sg, mat = gids[order], rows[order]
    sh.remove(sg[::9])                                                             # dead rows listed among the candidates are ignored
    alive = np.ones(n, bool); alive[::9] = False
    rng = np.random.default_rng(5)
    qs = O.gen_rows(42, 0, 4, dim)
    for frac, k in ((0.5, 10), (0.02, 10), (0.3, 200), (0.001, 7), (1.0, n)):
        sel = rng.random(n) < frac
        sel_ids = sg[sel]
        rng.shuffle(sel_ids)                                                       # any order, duplicates allowed
        sel_ids = np.concatenate([sel_ids, sel_ids[:3]])
        r, d, c = sh.search_masked(qs, k, sel_ids)
        cand = sel & alive
        for i, q in enumerate(qs):
            if cand.sum() == 0:
                assert c[i] == 0
                continue
            er, ed = _oracle_topk(mid, mat[cand], sg[cand], q, k)
            assert c[i] == len(er) == min(k, cand.sum())
            assert np.array_equal(r[i, :c[i]], er) and np.array_equal(d[i, :c[i]].view(np.uint32), ed.view(np.uint32)), (frac, k, i)
            assert (r[i, c[i]:] == 0xFFFFFFFF).all()
    r, d, c = sh.search_masked(qs, 5, np.zeros(0, np.uint32))                      # nothing selected: no results, no error
    assert (c == 0).all() and (r == 0xFFFFFFFF).all()
    with pytest.raises(quiver_amd.QvError) as e:
        sh.search_masked(qs, 5, np.array([0xFFFFFF00], np.uint32))
    assert e.value.code == quiver_amd._lib.QV_ERR_OUT_OF_RANGE
    sh.close()


@pytest.mark.parametrize("n_shards", SHARDS)
@pytest.mark.parametrize("k_fetch", [30, 64, 150])
def test_negative_example_fetch_equals_single_index(n_shards, k_fetch):
    """hybrid_index.go:517-546: the retrieveK nearest rows of the query and, for exactly those, the distance to the negative"""
    mid, dim, n = 0, 56, 1800
    rows = O.gen_rows(51, 0, n, dim)
    sh = _make(n_shards, dim, "cosine")
    gids = sh.add(rows)
    order = np.argsort(gids, kind="stable")
    sg, mat = gids[order], rows[order]
    q, neg = O.gen_rows(52, 0, 2, dim)
    r, d, nd, c = sh.search_negative(q, neg, k_fetch)
    er, ed = _oracle_topk(mid, mat, sg, q, k_fetch)
    assert c == len(er) and np.array_equal(r[:c], er) and np.array_equal(d[:c].view(np.uint32), ed.view(np.uint32))
    row_of = {int(g): i for i, g in enumerate(sg)}
    want = O.all_distances(mid, mat[[row_of[int(g)] for g in r[:c]]], neg)
    assert np.array_equal(nd[:c].view(np.uint32), want.view(np.uint32))
    # fewer rows than k_fetch: clamped, padded
    small = _make(n_shards, dim, "cosine")
    g2 = small.add(rows[:5])
    r, d, nd, c = small.search_negative(q, neg, k_fetch)
    assert c == 5 and sorted(r[:5].tolist()) == sorted(g2.tolist()) and (r[5:] == 0xFFFFFFFF).all() and np.isinf(nd[5:]).all()
    sh.close(); small.close()


@pytest.mark.parametrize("n_shards", SHARDS)
def test_distance_rows_update_get(n_shards):
    mid, dim, n = 1, 33, 700
    rows = O.gen_rows(61, 0, n, dim)
    sh = _make(n_shards, dim, "l2")
    gids = sh.add(rows)
    assert sh.rows() == n and sh.size() == n
    q = O.gen_rows(62, 0, 1, dim)[0]
    pick = np.array([0, 699, 5, 5, 350, 123], dtype=np.int64)
    got = sh.distance_rows(q, gids[pick])
    assert np.array_equal(got.view(np.uint32), O.all_distances(mid, rows[pick], q).view(np.uint32))
    assert np.array_equal(sh.get_rows(gids[pick]).view(np.uint32), rows[pick].view(np.uint32))
    assert np.array_equal(sh.get_row(gids[77]).view(np.uint32), rows[77].view(np.uint32))
    # update: overwrite a live row and revive a dead one (Collection.Update = delete + insert under one lock, collection.go:417-465)
    new = O.gen_rows(63, 0, 2, dim)
    sh.update(gids[10], new[0]); rows[10] = new[0]
    sh.remove(gids[20:21]); assert sh.size() == n - 1
    sh.update(gids[20], new[1]); rows[20] = new[1]; assert sh.size() == n
    order = np.argsort(gids, kind="stable")
    r, d, c = sh.search(q, 12)
    er, ed = _oracle_topk(mid, rows[order], gids[order], q, 12)
    assert np.array_equal(r[0], er) and np.array_equal(d[0].view(np.uint32), ed.view(np.uint32))
    with pytest.raises(quiver_amd.QvError):
        sh.update(0xFFFFFF00, new[0])
    with pytest.raises(quiver_amd.QvError):
        sh.get_row(0xFFFFFF00)
    with pytest.raises(quiver_amd.QvError):
        sh.distance_rows(q, np.array([0xFFFFFF00], np.uint32))
    sh.close()


def test_index_get_rows_equals_get_row():
    rows = O.gen_rows(71, 0, 300, 19)
    idx = DeviceIndex(19, "cosine")
    idx.add(rows)
    pick = np.array([299, 0, 64, 63, 64, 128], np.uint32)
    assert np.array_equal(idx.get_rows(pick).view(np.uint32), rows[pick].view(np.uint32))
    assert idx.get_rows(np.zeros(0, np.uint32)).shape == (0, 19)
    with pytest.raises(quiver_amd.QvError):
        idx.get_rows(np.array([300], np.uint32))


@pytest.mark.parametrize("n_shards", [1, 3])
def test_concurrent_searches_on_one_handle(n_shards):
    """the reference runs Index.Search under a read lock (collection.go:647): many callers at once on one handle, each in a
    call context of its own; a writer (add) interleaves exclusively"""
    mid, dim, n = 0, 64, 20000
    rows = O.gen_rows(81, 0, n, dim)
    sh = _make(n_shards, dim, "cosine")
    gids = sh.add(rows)
    order = np.argsort(gids, kind="stable")
    sg, mat = gids[order], rows[order]
    qs = O.gen_rows(82, 0, 24, dim)
    want = [_oracle_topk(mid, mat, sg, q, 10) for q in qs]
    errs = []

    def worker(t):
        try:
            for rep in range(6):
                i = (t * 5 + rep) % len(qs)
                if rep % 3 == 2:
                    r, d, c = sh.search_masked(qs[i], 10, sg)                      # every row selected: same answer
                elif rep % 3 == 1:
                    r, d, c = sh.search(qs[i], 100); r, d = r[:, :10], d[:, :10]   # the ranked path
                else:
                    r, d, c = sh.search(qs[i], 10)
                if not (np.array_equal(r[0], want[i][0]) and np.array_equal(d[0].view(np.uint32), want[i][1].view(np.uint32))):
                    errs.append((t, rep))
        except Exception as ex:  # noqa: BLE001
            errs.append(repr(ex))

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert errs == []
    sh.close()


def test_batch_with_queries_the_filter_hands_back_is_redone_on_the_device_without_a_host_round_trip():
    """A corpus stored cluster by cluster makes the matrix-core filter hand some queries back (candidate overflow); on a sharded handle
    those are listed and re-scanned exactly ON THE DEVICE (k_redo_compact + k_flat_scan_redo) behind the batch — the device-pointer call
    stays asynchronous.  Results must equal the exact scan of one index over the same rows, handed-back queries included."""
    import torch
    rng = np.random.default_rng(12)
    dim, n_clusters, per = 768, 30, 4_000
    centres = rng.standard_normal((n_clusters, dim)).astype(np.float32)
    rows = np.concatenate([c + 0.3 * rng.standard_normal((per, dim)).astype(np.float32) for c in centres])
    nq, k = 256, 10
    qs = (centres[rng.integers(0, n_clusters, nq)] + 0.3 * rng.standard_normal((nq, dim))).astype(np.float32)
    n = rows.shape[0]
    one = quiver_amd.DeviceIndex(dim, "cosine", filter="off")
    one.add(rows)
    er, ed, _ = one.search(qs, k)
    # the shards' own filter runs do hand queries back: shard 0's rows in an index of their own, through the batched device entry point
    G = 3
    sh = quiver_amd.ShardedIndex(dim, "cosine", devices=[0] * G, peer_copy=True)
    gids = sh.add(rows)
    info = [sh.shard_info(g) for g in range(G)]
    first = quiver_amd.DeviceIndex(dim, "cosine")
    first.add(rows[:info[0]["rows"]])
    dq = torch.from_numpy(qs).cuda()
    fr = torch.empty((nq, k), dtype=torch.int32, device="cuda"); fd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    fl = torch.zeros((nq,), dtype=torch.int32, device="cuda")
    first.search_batched_device(dq.data_ptr(), nq, k, fr.data_ptr(), fd.data_ptr(), fl.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int((fl != 0).sum().item()) >= 1, "this corpus no longer makes the filter hand anything back: the test exercises nothing"
    # the sharded batch, device pointers in and out
    dr = torch.empty((nq, k), dtype=torch.int32, device="cuda"); dd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    sh.search_device(dq.data_ptr(), nq, k, dr.data_ptr(), dd.data_ptr(), torch.cuda.current_stream().cuda_stream)
    sh.sync(); torch.cuda.synchronize()
    got = dr.cpu().numpy().view(np.uint32)
    want = gids[er]                                        # global ids of the single index's rows
    assert np.array_equal(got, want)
    assert np.array_equal(dd.cpu().numpy().view(np.uint32), ed.view(np.uint32))
    # and through host pointers
    hr, hd, hc = sh.search(qs, k)
    assert np.array_equal(hr, want) and np.array_equal(hd.view(np.uint32), ed.view(np.uint32))
    # a batch large enough that one shard hands back MORE than 64 queries: the redo kernel serves 64 list entries per launch, so this
    # one goes through several of them
    nq2 = 2048
    qs2 = (centres[rng.integers(0, n_clusters, nq2)] + 0.3 * rng.standard_normal((nq2, dim))).astype(np.float32)
    dq2 = torch.from_numpy(qs2).cuda()
    fr2 = torch.empty((nq2, k), dtype=torch.int32, device="cuda"); fd2 = torch.empty((nq2, k), dtype=torch.float32, device="cuda")
    fl2 = torch.zeros((nq2,), dtype=torch.int32, device="cuda")
    first.search_batched_device(dq2.data_ptr(), nq2, k, fr2.data_ptr(), fd2.data_ptr(), fl2.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    handed_back = int((fl2 != 0).sum().item())
    er2, ed2, _ = one.search(qs2, k)
    dr2 = torch.empty((nq2, k), dtype=torch.int32, device="cuda"); dd2 = torch.empty((nq2, k), dtype=torch.float32, device="cuda")
    sh.search_device(dq2.data_ptr(), nq2, k, dr2.data_ptr(), dd2.data_ptr(), torch.cuda.current_stream().cuda_stream)
    sh.sync(); torch.cuda.synchronize()
    assert np.array_equal(dr2.cpu().numpy().view(np.uint32), gids[er2]) and np.array_equal(dd2.cpu().numpy().view(np.uint32), ed2.view(np.uint32))
    assert handed_back > 64, "only %d of %d queries were handed back by shard 0's filter: the multi-launch redo was not exercised" % (handed_back, nq2)
