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
