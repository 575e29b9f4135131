"""Differential fuzzing of the flat index through the C ABI: seeded random sequences of add / remove / update / search /
masked search / full ranking / distance_rows / get_row against a host model that asks the CPU oracle
(exact.go:56-133 semantics: copy on insert, tombstones, (distance, row) order).  Every comparison is bit-exact.

The vectors are drawn from a small alphabet of values so that equal distances, duplicates, zero vectors and sign
cancellations happen all the time — the cases a seeded Gaussian never produces."""
import numpy as np
import pytest

import quiver_amd
from tests import _oracle as O

pytestmark = pytest.mark.gpu

METRICS = ["cosine", "l2", "dot", "l1", "l2sq", "cosine_f32", "l2_f32", "dot_f32", "l2sq_f64"]


def _vectors(rng, n, dim, style):
    if style == 0:                                   # few distinct values per component: many exact ties
        return rng.choice(np.array([-2.0, -1.0, -0.5, 0.0, 0.0, 0.5, 1.0, 3.0], np.float32), size=(n, dim))
    if style == 1:                                   # wide dynamic range: cancellation and rounding in every accumulate
        return (rng.standard_normal((n, dim)) * np.exp2(rng.integers(-20, 20, size=(n, dim)))).astype(np.float32)
    return rng.standard_normal((n, dim)).astype(np.float32)


class Model:
    def __init__(self, metric, dim):
        self.mid = quiver_amd.metric_id(metric)
        self.rows = np.zeros((0, dim), np.float32)
        self.alive = np.zeros(0, np.uint8)

    def add(self, x):
        first = self.rows.shape[0]
        self.rows = np.concatenate([self.rows, x]); self.alive = np.concatenate([self.alive, np.ones(len(x), np.uint8)])
        return first

    def search(self, q, k, sel=None):
        a = self.alive if sel is None else (self.alive & sel.astype(np.uint8))
        return O.exact_search(self.mid, self.rows, q, k, alive=a)


def _compare(model, got, qs, k, sel=None):
    r, d, c = got
    for i in range(qs.shape[0]):
        ro, do = model.search(qs[i], k, sel)
        assert int(c[i]) == ro.size, (i, int(c[i]), ro.size)
        assert r[i, :ro.size].tolist() == ro.tolist(), i
        assert d[i, :ro.size].tobytes() == do.tobytes(), i


@pytest.mark.parametrize("seed", range(27))
def test_random_operation_sequences_match_the_oracle(seed):
    rng = np.random.default_rng(1000 + seed)
    metric = METRICS[seed % len(METRICS)]
    dim = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 16, 31, 33, 64, 70]))
    style = seed % 3
    idx = quiver_amd.DeviceIndex(dim, metric)
    m = Model(metric, dim)
    for step in range(40):
        n = m.rows.shape[0]
        op = rng.choice(["add", "add", "remove", "update", "search", "search", "multi", "masked", "rank", "rows", "get"])
        if op == "add" or n == 0:
            cnt = int(rng.choice([1, 2, 63, 64, 65, 130, 700]))
            x = _vectors(rng, cnt, dim, style)
            if n and rng.random() < 0.5:
                x[0] = m.rows[rng.integers(n)]                         # a duplicate of an existing row (possibly a dead one)
            assert idx.add(x) == m.add(x)
        elif op == "remove":
            who = rng.integers(0, n, size=int(rng.integers(1, 40))).astype(np.uint32)      # repeats and already-dead rows included
            idx.remove(who); m.alive[who] = 0
        elif op == "update":
            row = int(rng.integers(n)); v = _vectors(rng, 1, dim, style)[0]
            idx.update(row, v); m.rows[row] = v; m.alive[row] = 1      # update revives (qv_index_update contract)
        elif op in ("search", "multi"):
            nq = 1 if op == "search" else int(rng.choice([2, 3, 5, 8, 9, 17]))
            qs = _vectors(rng, nq, dim, style)
            if rng.random() < 0.4:
                qs[0] = m.rows[rng.integers(n)]
            k = int(rng.choice([1, 2, 10, 64, 65, 200]))
            _compare(m, idx.search(qs, k), qs, k)
        elif op == "masked":
            sel = rng.random(n) < rng.choice([0.02, 0.3, 0.9])
            qs = _vectors(rng, int(rng.choice([1, 4])), dim, style)
            k = int(rng.choice([1, 10, 100]))
            _compare(m, idx.search_masked(qs, k, sel), qs, k, sel)
        elif op == "rank":
            qs = _vectors(rng, 1, dim, style)
            k = idx.size()
            assert k == int(m.alive.sum())
            if k:
                _compare(m, idx.search(qs, k), qs, k)
        elif op == "rows":
            q = _vectors(rng, 1, dim, style)[0]
            who = rng.integers(0, n, size=int(rng.integers(1, 70))).astype(np.uint32)
            got = idx.distance_rows(q, who)
            want = np.array([O.distance(m.mid, q, m.rows[w]) for w in who], np.float32)
            assert got.tobytes() == want.tobytes()
        else:
            row = int(rng.integers(n))
            assert idx.get_row(row).tobytes() == m.rows[row].tobytes()
        assert idx.rows() == m.rows.shape[0] and idx.size() == int(m.alive.sum())
    idx.close()


def _graphs(h, o):
    hg = [(h.node_level(i), [h.links(i, l).tolist() for l in range(h.node_level(i) + 1)]) for i in range(h.nodes())]
    og = [(o.node_level(i), [o.links(i, l).tolist() for l in range(o.node_level(i) + 1)]) for i in range(o.nodes())]
    return hg, og


@pytest.mark.parametrize("seed", range(16))
def test_random_hnsw_histories_match_the_oracle(seed):
    """Insert / InsertBatch (device build, several batch schedules) / Delete / Search interleaved at random: after every
    mutation the host mirror's graph equals the CPU restatement's link for link, and every search returns its rows and
    float32 bits (hnsw.go:266-468, 471-713, 745-842).  Tie-heavy vectors on the even seeds.  Ids are zero-padded so that their
    string order (the top-up's tie-break, hnsw.go:699-704) is the node order the oracle uses."""
    from quiver_amd import hnsw
    from quiver_amd.device_index import graph_batch_size
    rng = np.random.default_rng(5000 + seed)
    metric = [6, 5, 7, 0, 1, 3, 4, 2][seed % 8]
    dim = int(rng.choice([4, 8, 24, 32]))
    style = 0 if seed % 2 == 0 else 2
    cfg = dict(M=int(rng.choice([4, 8, 16])), EfConstruction=int(rng.choice([16, 40, 100])), EfSearch=int(rng.choice([8, 32, 64])),
               MaxLevel=int(rng.choice([1, 4, 16])))
    h = hnsw.HNSW(hnsw.Config(DistanceFunc=metric, Seed=seed + 1, **cfg))
    o = O.HNSW(metric, dim, seed=seed + 1, M=cfg["M"], efConstruction=cfg["EfConstruction"], efSearch=cfg["EfSearch"], maxLevel=cfg["MaxLevel"])
    live, nxt = [], 0
    for step in range(24):
        op = rng.choice(["insert", "batch", "batch", "delete", "search", "search"]) if nxt else "batch"
        if op == "insert":
            v = _vectors(rng, 1, dim, style)[0]
            h.Insert("n%06d" % nxt, v); o.insert(v); live.append(nxt); nxt += 1
        elif op == "batch":
            m = int(rng.choice([3, 40, 150]))
            x = _vectors(rng, m, dim, style)
            bm, rd = int(rng.choice([1, 16, 64])), int(rng.choice([0, 8]))
            h.InsertBatch(["n%06d" % (nxt + i) for i in range(m)], x, bm, rd)
            done = 0
            while done < m:
                b = min(graph_batch_size(nxt + done, bm, rd), m - done)
                o.insert_batch(x[done:done + b]); done += b
            live.extend(range(nxt, nxt + m)); nxt += m
        elif op == "delete" and live:
            who = live.pop(int(rng.integers(len(live))))
            h.Delete("n%06d" % who); assert o.delete(who) == 0
        else:
            qs = _vectors(rng, 4, dim, style)
            k = int(rng.choice([1, 5, 20]))
            res = h.SearchBatch(qs, k)
            for i in range(4):
                er, ed = o.search(qs[i], k)
                assert len(res[i]) == len(er), (step, i)
                assert np.asarray([r.Distance for r in res[i]], np.float32).tobytes() == ed.tobytes(), (step, i)
                assert [r.VectorIndex for r in res[i]] == er.tolist(), (step, i)
            continue
        assert h.Size() == o.size() == len(live)
        hg, og = _graphs(h, o)
        assert hg == og, step
        assert h.entry_point() == o.entry_point()


@pytest.mark.parametrize("seed", range(8))
def test_random_sharded_histories_match_the_oracle(seed):
    """qv_sharded_* with 1..8 co-located shards (point-to-point exchange): add / remove / update / search (any k) / filtered search /
    negative-example fetch / listed-row distances in random order; the answer is the oracle's over the live rows keyed by global
    row id, (distance, global row) order, float32 bits"""
    from quiver_amd import ShardedIndex
    rng = np.random.default_rng(9000 + seed)
    metric = METRICS[seed % len(METRICS)]
    mid = quiver_amd.metric_id(metric)
    dim = int(rng.choice([3, 8, 20, 64]))
    shards = int(rng.integers(1, 9))
    style = seed % 3
    idx = ShardedIndex(dim, metric, devices=[0] * shards, peer_copy=True)
    live, dead = {}, []
    for step in range(40):
        op = rng.choice(["add", "add", "remove", "search", "search", "update", "masked", "ranked", "negative", "rows"]) if live else "add"
        if op == "update":                                       # overwrite a live row, or revive a tombstoned one
            g = dead.pop() if dead and rng.random() < 0.5 else list(live)[int(rng.integers(len(live)))]
            x = _vectors(rng, 1, dim, style)[0]
            idx.update(g, x)
            live[g] = x
        elif op in ("masked", "ranked", "negative", "rows"):
            gids = np.array(sorted(live), dtype=np.uint32)
            mat = np.stack([live[g] for g in gids])
            q = _vectors(rng, 1, dim, style)[0]
            if op == "masked":
                sel = rng.random(gids.size) < rng.choice([0.05, 0.5, 1.0])
                extra = np.array(dead[:3], dtype=np.uint32)        # tombstoned rows among the candidates are ignored
                k = int(rng.choice([1, 10, 100]))
                r, d, c = idx.search_masked(q, k, np.concatenate([gids[sel], extra]))
                if sel.sum() == 0:
                    assert c[0] == 0
                else:
                    er, ed = O.exact_search(mid, mat[sel], q, k)
                    assert int(c[0]) == er.size and r[0, :er.size].tolist() == gids[sel][er].tolist() and d[0, :er.size].tobytes() == ed.tobytes(), step
            elif op == "ranked":
                k = int(rng.choice([65, 200, gids.size, gids.size + 7]))
                r, d, c = idx.search(q, k)
                er, ed = O.exact_search(mid, mat, q, k)
                assert int(c[0]) == er.size and r[0, :er.size].tolist() == gids[er].tolist() and d[0, :er.size].tobytes() == ed.tobytes(), step
            elif op == "negative":
                neg = _vectors(rng, 1, dim, style)[0]
                kf = int(rng.choice([30, 64, 90]))
                r, d, nd, c = idx.search_negative(q, neg, kf)
                er, ed = O.exact_search(mid, mat, q, kf)
                assert c == er.size and r[:c].tolist() == gids[er].tolist() and d[:c].tobytes() == ed.tobytes(), step
                assert nd[:c].tobytes() == O.all_distances(mid, mat[er], neg).tobytes(), step
            else:
                pick = rng.integers(0, gids.size, size=int(rng.choice([1, 5, 40])))
                assert idx.distance_rows(q, gids[pick]).tobytes() == O.all_distances(mid, mat[pick], q).tobytes(), step
                assert idx.get_rows(gids[pick]).tobytes() == mat[pick].tobytes(), step
        elif op == "add":
            x = _vectors(rng, int(rng.choice([1, 2, 7, 64, 300])), dim, style)
            gids = idx.add(x)
            assert len(set(gids.tolist()) & set(live)) == 0
            live.update(zip(gids.tolist(), x))
        elif op == "remove":
            who = [g for g in live if rng.random() < 0.1][:50] or [next(iter(live))]
            idx.remove(who)
            for g in who:
                del live[g]
            dead.extend(who)
        elif live:
            gids = np.array(sorted(live), dtype=np.uint32)
            mat = np.stack([live[g] for g in gids])
            qs = _vectors(rng, int(rng.choice([1, 3, 9])), dim, style)
            k = int(rng.choice([1, 10, 64]))
            r, d, c = idx.search(qs, k)
            for i in range(qs.shape[0]):
                er, ed = O.exact_search(mid, mat, qs[i], k)
                assert int(c[i]) == er.size
                assert r[i, :er.size].tolist() == gids[er].tolist(), (step, i)
                assert d[i, :er.size].tobytes() == ed.tobytes(), (step, i)
        assert idx.size() == len(live)
    idx.close()


@pytest.mark.parametrize("seed", range(8))
def test_random_hybrid_histories(seed):
    """hybrid.HybridIndex (hybrid_index.go:224-811) under random Insert / InsertBatch / Delete / DeleteBatch: forced-exact
    searches equal the oracle's exact search over the live vectors (ids, float32 bits), single and batched; forced-HNSW
    batched searches (device traversal) equal the single-query form (host-driven traversal) result for result."""
    from quiver_amd import hybrid
    rng = np.random.default_rng(12000 + seed)
    metric = ["cosine", "euclidean", "dot_product", "manhattan"][seed % 4]
    mid = quiver_amd.metric_id(metric)
    dim = int(rng.choice([4, 16, 40]))
    cfg = hybrid.IndexConfig(DistanceFunc=metric, Seed=seed + 1, ExplorationFactor=0.0)
    cfg.HNSWConfig = hybrid.HNSWConfig(M=int(rng.choice([4, 16])), MaxM0=0, EfConstruction=int(rng.choice([20, 100])), EfSearch=int(rng.choice([16, 64])))
    cfg.HNSWConfig.MaxM0 = 2 * cfg.HNSWConfig.M
    n_shards = (1, 3, 8)[seed % 3]                                # the exact index on one device, or sharded (co-located shards)
    if n_shards > 1:
        cfg.Devices, cfg.PeerCopy = [0] * n_shards, True
    idx = hybrid.HybridIndex(cfg)
    live, nxt = {}, 0
    for step in range(22):
        op = rng.choice(["insert", "batch", "batch", "delete", "delbatch", "exact", "exact", "hnsw"]) if live else "batch"
        if op == "insert":
            v = _vectors(rng, 1, dim, 2)[0]
            idx.Insert("id%06d" % nxt, v); live["id%06d" % nxt] = v; nxt += 1
        elif op == "batch":
            x = _vectors(rng, int(rng.choice([2, 30, 200])), dim, 2)
            new = {"id%06d" % (nxt + i): x[i] for i in range(len(x))}
            idx.InsertBatch(new); live.update(new); nxt += len(x)
        elif op == "delete":
            who = list(live)[int(rng.integers(len(live)))]
            idx.Delete(who); del live[who]
        elif op == "delbatch":
            who = [i for i in live if rng.random() < 0.15][:40]
            if who:
                idx.DeleteBatch(who)
                for i in who:
                    del live[i]
        elif op == "exact" and live:
            ids = sorted(live)
            mat = np.stack([live[i] for i in ids])
            qs = _vectors(rng, int(rng.choice([1, 5, 12])), dim, 2)
            k = int(rng.choice([1, 7, 30]))
            resp = idx.BatchSearch(hybrid.BatchSearchRequest(Queries=list(qs), K=k, ForceStrategy="exact"))
            for i in range(qs.shape[0]):
                er, ed = O.exact_search(mid, mat, qs[i], k)
                one = idx.SearchWithRequest(hybrid.HybridSearchRequest(Query=qs[i], K=k, ForceStrategy="exact")).Results
                for got in (resp.Results[i], one):
                    assert [r.ID for r in got] == [ids[j] for j in er], (step, i)
                    assert np.asarray([r.Distance for r in got], np.float32).tobytes() == ed.tobytes(), (step, i)
        elif live:
            qs = _vectors(rng, 6, dim, 2)
            k = int(rng.choice([1, 5, 15]))
            resp = idx.BatchSearch(hybrid.BatchSearchRequest(Queries=list(qs), K=k, ForceStrategy="hnsw"))
            assert resp.StrategiesUsed == ["hnsw"] * 6
            for i in range(6):
                one = idx.SearchWithRequest(hybrid.HybridSearchRequest(Query=qs[i], K=k, ForceStrategy="hnsw")).Results
                assert [(r.ID, np.float32(r.Distance).tobytes()) for r in resp.Results[i]] == [(r.ID, np.float32(r.Distance).tobytes()) for r in one], (step, i)
                assert len(one) == min(k, len(live))
        assert idx.Size() == len(live)


@pytest.mark.parametrize("seed", range(24))
def test_filter_path_under_random_mutations_equals_the_exact_scan(seed, monkeypatch):
    """the matrix-core filter + re-score (qv_index_search's route for 9+ queries over >= 32768 rows) on indexes that grow, lose and
    replace rows between batches, with and without the bfloat16 row copy: every batch equals the exact multi-query scan of the same
    index (the filter switched off: qv_index_set_filter), and a few queries per seed equal the CPU oracle"""
    rng = np.random.default_rng(5000 + seed)
    metric = ["cosine", "dot_product", "euclidean", "squared_euclidean"][seed % 4]
    dim = int(rng.choice([32, 64, 128, 200, 256, 768]))
    idx = quiver_amd.DeviceIndex(dim, metric, bf16_rows=bool(seed & 1))
    m = Model(metric, dim)
    style = seed % 3
    x = _vectors(rng, 45_000, dim, style)
    assert idx.add(x) == m.add(x)
    for step in range(6):
        n = m.rows.shape[0]
        op = rng.choice(["add", "remove", "update"])
        if op == "add":
            x = _vectors(rng, int(rng.choice([1, 64, 1000, 9000])), dim, style)
            assert idx.add(x) == m.add(x)
        elif op == "remove":
            lo = int(rng.integers(0, n - 3000))
            who = np.arange(lo, lo + int(rng.integers(1, 3000)), dtype=np.uint32)
            idx.remove(who); m.alive[who] = 0
        else:
            for row in rng.integers(0, n, size=20):
                v = _vectors(rng, 1, dim, style)[0]
                idx.update(int(row), v); m.rows[row] = v; m.alive[row] = 1
        n = m.rows.shape[0]
        nq = int(rng.choice([9, 40, 64, 100, 300, 600]))
        k = int(rng.choice([1, 10, 64]))
        qs = _vectors(rng, nq, dim, style)
        qs[0] = m.rows[rng.integers(n)]
        got = idx.search(qs, k)                                    # the filter path (>= 1 M query-rows) or the exact scan below it
        idx.set_filter("off")                                      # the exact multi-query scans of the same index
        want = idx.search(qs, k)
        idx.set_filter(quiver_amd.DeviceIndex.default_filter)
        assert np.array_equal(got[0], want[0]) and got[1].tobytes() == want[1].tobytes() and np.array_equal(got[2], want[2]), (seed, step)
        _compare(m, (got[0][:2], got[1][:2], got[2][:2]), qs[:2], k)
    idx.close()


@pytest.mark.parametrize("seed", range(32))
def test_selection_path_under_random_mutations_equals_the_exact_scan(seed):
    """the same for batches of 16 or more results per query over 131 072 rows or more (round 6's selection path: a guessed bound checked after
    the filter, k_cand_narrow, a wave per 32 survivors or the pass over the tiles, the in-LDS selection), with and without the row-major
    and bfloat16 copies, on value alphabets that make exact ties, overflowing candidate lists and failing guesses routine"""
    rng = np.random.default_rng(7000 + seed)
    metric = ["cosine", "dot_product", "euclidean", "squared_euclidean"][seed % 4]
    dim = int(rng.choice([32, 64, 100, 128, 256, 768]))
    idx = quiver_amd.DeviceIndex(dim, metric, rowmajor=bool(seed & 2), bf16_rows=bool(seed & 1))
    m = Model(metric, dim)
    style = seed % 3
    x = _vectors(rng, 140_000, dim, style)
    assert idx.add(x) == m.add(x)
    for step in range(4):
        n = m.rows.shape[0]
        op = rng.choice(["add", "remove", "update"])
        if op == "add":
            x = _vectors(rng, int(rng.choice([1, 64, 9000, 30000])), dim, style)
            assert idx.add(x) == m.add(x)
        elif op == "remove":
            lo = int(rng.integers(0, n - 6000))
            who = np.arange(lo, lo + int(rng.integers(1, 6000)), dtype=np.uint32)
            idx.remove(who); m.alive[who] = 0
        else:
            for row in rng.integers(0, n, size=20):
                v = _vectors(rng, 1, dim, style)[0]
                idx.update(int(row), v); m.rows[row] = v; m.alive[row] = 1
        n = m.rows.shape[0]
        nq = int(rng.choice([9, 40, 100, 256, 300]))
        k = int(rng.choice([16, 40, 64, 100, 300, 1000]))
        qs = _vectors(rng, nq, dim, style)
        qs[0] = m.rows[rng.integers(n)]
        got = idx.search(qs, k)
        idx.set_filter("off")
        want = idx.search(qs, k)
        idx.set_filter(quiver_amd.DeviceIndex.default_filter)
        assert np.array_equal(got[0], want[0]) and got[1].tobytes() == want[1].tobytes() and np.array_equal(got[2], want[2]), (seed, step)
        _compare(m, (got[0][:2], got[1][:2], got[2][:2]), qs[:2], k)
    idx.close()
