"""Host-side mirrors of the reference's Go callers (pkg/hybrid, pkg/hnsw, pkg/core,
pkg/vectortypes) driving the HIP path; written to read like the reference's own tests
(the cited *_test.go) and checked against the CPU oracle."""
import json
import os

import numpy as np
import pytest

from tests import _oracle as O

pytestmark = pytest.mark.gpu

KATS = json.load(open(os.path.join(O.ROOT, "tests", "golden", "ref_kats.json")))


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


# The exact index behind ExactIndex / HybridIndex / Collection on one device (a qv_index) and sharded over 3 and 8 row shards
# (a qv_sharded handle; co-located on device 0 with the point-to-point exchange, which is how a 1-GPU box runs several
# shards): the host mirror and its reference tables must not notice the difference.
PLACEMENTS = {"1dev": None, "3shards": {"Devices": [0, 0, 0], "PeerCopy": True}, "8shards": {"Devices": [0] * 8, "PeerCopy": True}}


@pytest.fixture(params=list(PLACEMENTS))
def placement(request, monkeypatch):
    from quiver_amd import hybrid
    monkeypatch.setattr(hybrid, "DEFAULT_PLACEMENT", PLACEMENTS[request.param])
    return request.param


# ------------------------------------------------------------------ vectortypes ---

def test_vectortypes_distance_funcs_kats():                     # distances_test.go:9-204
    from quiver_amd import vectortypes as vt
    fn = {0: vt.CosineDistance, 1: vt.EuclideanDistance, 2: vt.SquaredEuclideanDistance, 3: vt.DotProductDistance, 4: vt.ManhattanDistance}
    for kat in KATS["distance"]:
        if kat["metric"] in fn:
            assert abs(float(fn[kat["metric"]](kat["a"], kat["b"])) - kat["want"]) <= kat["tol"], kat


def test_vectortypes_panics_on_length_mismatch():               # distances_test.go:206-230
    from quiver_amd import vectortypes as vt
    for f in (vt.CosineDistance, vt.EuclideanDistance, vt.SquaredEuclideanDistance, vt.DotProductDistance, vt.ManhattanDistance):
        with pytest.raises(ValueError, match="vectors must have the same length"):
            f([1, 2, 3], [1, 2])


def test_vectortypes_lookup_and_surfaces():                     # types_test.go:9-60; surface_test.go:7-208
    from quiver_amd import vectortypes as vt
    assert vt.GetDistanceFuncByType("invalid") is vt.CosineDistance          # types.go:46-47
    assert vt.GetDistanceFuncByType(vt.Manhattan) is vt.ManhattanDistance
    assert float(vt.ComputeDistance([1, 1, 0], [0, 0, 0], vt.Manhattan)) == 2.0   # types_test.go:44-49
    with pytest.raises(ValueError):
        vt.ComputeDistance([1, 2], [1], vt.Cosine)
    assert float(vt.DotProductSurface.Distance([2, 0, 0], [2, 0, 0])) == -3.0     # surface_test.go:117-123
    assert float(vt.GetSurfaceByType("nope").Distance([1, 0, 0], [0, 1, 0])) == 1.0
    cm = vt.ContraMap(vt.CosineSurface, lambda s: [float(ord(c)) for c in s])     # surface_test.go:160-192
    assert abs(float(cm.Distance("abc", "xyz")) - 0.0) < 0.1
    rng = np.random.default_rng(0)
    a, b = rng.standard_normal((50, 33)).astype(np.float32), rng.standard_normal((50, 33)).astype(np.float32)
    got = vt.batch_distances("euclidean", a, b)
    want = np.array([O.distance(1, a[i], b[i]) for i in range(50)], np.float32)
    assert np.array_equal(_bits(got), _bits(want))


# ------------------------------------------------------------------ hybrid.ExactIndex ---

@pytest.mark.usefixtures("placement")
def test_exact_index_insert_delete_semantics():                 # exact_test.go:13-95
    from quiver_amd import hybrid, vectortypes as vt
    from quiver_amd._host import GoError
    idx = hybrid.ExactIndex(vt.CosineDistance)
    assert idx.Size() == 0
    v = np.array([0.1, 0.2], np.float32)
    idx.Insert("vec1", v)
    v[0] = 99                                                    # copy-on-insert, exact_test.go:46-60
    with pytest.raises(GoError) as e:
        idx.Insert("vec1", [0.3, 0.4])
    assert str(e.value) == "vector with ID vec1 already exists"
    with pytest.raises(GoError) as e:
        idx.Insert("vec2", [0.3])                                # exact_test.go:314-322
    assert str(e.value) == "vector dimension mismatch: expected 2, got 1"
    with pytest.raises(GoError) as e:
        idx.Search([0.3], 1)                                     # exact_test.go:324-330
    assert str(e.value) == "query dimension mismatch: expected 2, got 1"
    with pytest.raises(GoError) as e:
        idx.Search([0.1, 0.2], 0)
    assert str(e.value) == "k must be positive"
    r = idx.Search([0.1, 0.2], 5)
    assert [x.ID for x in r] == ["vec1"] and r[0].Distance <= 1e-6
    idx.Delete("nonexistent")                                    # exact_test.go:90-94: not an error
    idx.Delete("vec1")
    assert idx.Size() == 0
    assert idx.Search([1, 2, 3], 3) == []                        # empty -> empty, nil (exact.go:96-98)
    idx.Insert("a", [1, 2, 3])                                   # dimension resets when empty (exact.go:66-68)
    assert idx.Size() == 1


@pytest.mark.parametrize("kat", KATS["exact_search"], ids=lambda k: k["src"])
@pytest.mark.usefixtures("placement")
def test_exact_index_search_kats(kat):                          # exact_test.go:97-205
    from quiver_amd import hybrid
    idx = hybrid.ExactIndex(kat["metric"])
    for i, r in zip(kat["ids"], kat["rows"]):
        idx.Insert(i, r)
    res = idx.Search(kat["query"], kat["k"])
    ids = [r.ID for r in res]
    assert len(ids) == min(kat["k"], len(kat["ids"]))
    if kat["exact_order"]:
        assert ids[: len(kat["want_ids"])] == kat["want_ids"]
    else:
        assert set(kat["want_ids"]) <= set(ids)
    assert all(res[i].Distance <= res[i + 1].Distance for i in range(len(res) - 1))


# ------------------------------------------------------------------ hybrid.HybridIndex ---

def _hybrid(metric="cosine", **kw):
    from quiver_amd import hybrid
    cfg = hybrid.IndexConfig(DistanceFunc=metric, ExplorationFactor=kw.pop("exploration", 0.0), Seed=kw.pop("seed", 3))
    for k, v in kw.items():
        setattr(cfg, k, v)
    return hybrid.HybridIndex(cfg)


@pytest.mark.usefixtures("placement")
def test_hybrid_forced_exact_and_batch_insert():                # hybrid_index_test.go:270-311
    from quiver_amd import hybrid
    idx = _hybrid()
    idx.InsertBatch({"vec1": [1, 0, 0], "vec2": [0, 1, 0], "vec3": [0, 0, 1]})
    assert idx.Size() == 3
    resp = idx.SearchWithRequest(hybrid.HybridSearchRequest(Query=[0.9, 0.1, 0.0], K=1, ForceStrategy=hybrid.ExactIndexType, IncludeStats=True))
    assert resp.StrategyUsed == "exact" and len(resp.Results) == 1 and resp.Results[0].ID == "vec1"
    resp = idx.SearchWithRequest(hybrid.HybridSearchRequest(Query=[0.1, 0.9, 0.0], K=1, ForceStrategy=hybrid.ExactIndexType))
    assert resp.Results[0].ID == "vec2"                          # hybrid_index_test.go:349-360


@pytest.mark.usefixtures("placement")
def test_hybrid_errors_and_rollback():
    from quiver_amd import hybrid
    from quiver_amd._host import GoError
    idx = _hybrid()
    idx.Insert("a", [1, 0, 0])
    with pytest.raises(GoError, match="vector with ID a already exists"):
        idx.Insert("a", [0, 1, 0])
    with pytest.raises(GoError, match="vector dimension mismatch: expected 3, got 2"):
        idx.Insert("b", [0, 1])
    with pytest.raises(GoError, match="vector dimension mismatch: expected 3, got 2"):
        idx.InsertBatch({"c": [0, 1, 0], "d": [1, 1]})
    assert idx.Size() == 1                                       # all-or-nothing
    with pytest.raises(GoError, match="vector with ID a already exists"):
        idx.InsertBatch({"e": [0, 1, 0], "a": [1, 1, 1]})
    with pytest.raises(GoError, match="vector with ID zz not found"):
        idx.Delete("zz")
    with pytest.raises(GoError, match="some vectors not found"):
        idx.DeleteBatch(["a", "zz"])
    with pytest.raises(GoError, match="k must be positive"):
        idx.SearchWithRequest(hybrid.HybridSearchRequest(Query=[1, 0, 0], K=0))
    with pytest.raises(GoError, match="query dimension mismatch: expected 3, got 2"):
        idx.Search([1, 0], 1)
    with pytest.raises(GoError, match="invalid search strategy: bogus"):
        idx.SearchWithRequest(hybrid.HybridSearchRequest(Query=[1, 0, 0], K=1, ForceStrategy="bogus"))
    idx.DeleteBatch(["a"])
    assert idx.Size() == 0
    idx.Insert("n", [1, 2])                                      # dimension resets (hybrid_index.go:282-284)


@pytest.mark.usefixtures("placement")
def test_hybrid_exact_distances_l2_unit_axes():                 # hybrid_property_test.go:443-461
    idx = _hybrid("euclidean")
    idx.Insert("a", [1, 0, 0]); idx.Insert("b", [0, 1, 0]); idx.Insert("c", [0, 0, 1])
    from quiver_amd import hybrid
    res = idx.SearchWithRequest(hybrid.HybridSearchRequest(Query=[0, 0, 0], K=3, ForceStrategy="exact")).Results
    assert len(res) == 3 and all(abs(r.Distance - 1.0) < 1e-3 for r in res)


@pytest.mark.parametrize("kat", KATS["negative_rerank"], ids=lambda k: k["src"])
@pytest.mark.usefixtures("placement")
def test_hybrid_negative_example_kats(kat):                     # hybrid_index_test.go:541-656; hybrid_index_rerank_test.go:9-47
    from quiver_amd import hybrid
    idx = _hybrid(kat["metric"])
    for i, r in zip(kat["ids"], kat["rows"]):
        idx.Insert(i, r)
    req = hybrid.HybridSearchRequest(Query=kat["query"], K=kat["k"], NegativeExample=kat["negative"], NegativeWeight=kat["weight"],
                                     ForceStrategy="exact")
    res = idx.SearchWithRequest(req).Results
    assert len(res) == kat["want_count"]
    assert not any(np.isnan(r.Distance) for r in res)
    if kat.get("want_all_equal"):
        assert res[0].Distance == res[1].Distance == res[2].Distance
    if "not_first" in kat:
        assert res[0].ID != kat["not_first"]
    # identical to the oracle's restatement of hybrid_index.go:517-570
    rows = np.array(kat["rows"], np.float32)
    order = sorted(range(len(kat["ids"])), key=lambda i: kat["ids"][i])
    rank = np.empty(len(order), np.uint32); rank[order] = np.arange(len(order), dtype=np.uint32)
    er, ed = O.exact_search_negative(kat["metric"], rows, kat["query"], kat["negative"], kat["weight"], kat["k"], id_rank=rank)
    assert [r.ID for r in res] == [kat["ids"][i] for i in er]
    assert np.array_equal(_bits([r.Distance for r in res]), _bits(ed))
    # fluent form, hybrid_index_test.go:637-655
    fres = idx.FluentSearch(kat["query"]).WithK(kat["k"]).WithNegativeExample(kat["negative"]).WithNegativeWeight(kat["weight"]).WithForceStrategy("exact").Execute()
    assert [r.ID for r in fres.Results] == [r.ID for r in res]


@pytest.mark.usefixtures("placement")
def test_hybrid_negative_rerank_random_vs_oracle():
    from quiver_amd import hybrid
    rows = O.gen_rows(61, 0, 500, 24)
    ids = [f"v{i}" for i in range(500)]
    idx = _hybrid("cosine")
    idx.InsertBatch({i: r for i, r in zip(ids, rows)})
    order = sorted(range(500), key=lambda i: ids[i])
    rank = np.empty(500, np.uint32); rank[order] = np.arange(500, dtype=np.uint32)
    qs, negs = O.gen_rows(62, 0, 5, 24), O.gen_rows(63, 0, 5, 24)
    for q, n in zip(qs, negs):
        for k, w in ((5, 0.5), (20, 0.9), (40, 0.1)):
            res = idx.SearchWithRequest(hybrid.HybridSearchRequest(Query=q, K=k, NegativeExample=n, NegativeWeight=w, ForceStrategy="exact")).Results
            er, ed = O.exact_search_negative(0, rows, q, n, w, k, id_rank=rank)
            assert [r.ID for r in res] == [ids[i] for i in er]
            assert np.array_equal(_bits([r.Distance for r in res]), _bits(ed))


@pytest.mark.usefixtures("placement")
def test_hybrid_batch_search_equals_single_searches():          # hybrid_index.go:677-811
    from quiver_amd import hybrid
    from quiver_amd._host import GoError
    rows = O.gen_rows(71, 0, 3000, 32)
    idx = _hybrid("cosine")
    idx.InsertBatch({f"v{i}": r for i, r in enumerate(rows)})
    qs = O.gen_rows(72, 0, 9, 32)
    resp = idx.BatchSearch(hybrid.BatchSearchRequest(Queries=list(qs), K=7, ForceStrategy="exact"))
    assert resp.StrategiesUsed == ["exact"] * 9
    for i, q in enumerate(qs):
        er, ed = O.exact_search(0, rows, q, 7)
        assert [r.ID for r in resp.Results[i]] == [f"v{j}" for j in er]
        assert np.array_equal(_bits([r.Distance for r in resp.Results[i]]), _bits(ed))
    with pytest.raises(GoError, match="no queries provided"):
        idx.BatchSearch(hybrid.BatchSearchRequest(Queries=[], K=3))
    with pytest.raises(GoError, match="query 1 dimension mismatch"):
        idx.BatchSearch(hybrid.BatchSearchRequest(Queries=[qs[0], qs[1][:5]], K=3))


def test_selector_threshold_overwrite_quirk():                  # adaptive.go:226-231; adaptive_test.go:44-104
    idx = _hybrid("cosine", exploration=0.0)
    assert idx.SelectStrategy(10, 3, 5) == "exact"              # fresh: 10 < ExactThreshold=1000
    assert idx.SelectStrategy(5000, 200, 10) == "hnsw"          # dim 200 > 100, k < 50
    assert idx.SelectStrategy(5000, 200, 60) == "exact"
    assert idx.SelectStrategy(5000, 50, 10) == "hnsw"
    for i in range(5):
        idx.Insert(f"v{i}", np.eye(4, dtype=np.float32)[i % 4] + i)
    # Insert overwrote exactThreshold := VectorCount and dimThreshold := AvgDimension, so both
    # comparisons are now always false for the index's own stats and un-forced searches go HNSW
    assert idx.SelectStrategy(5, 4, 10) == "hnsw"


# ------------------------------------------------------------------ hnsw.HNSW (host graph, GPU distances) ---

def _build_both(metric, rows, seed, **cfg):
    from quiver_amd import hnsw
    h = hnsw.HNSW(hnsw.Config(DistanceFunc=metric, Seed=seed, **cfg))
    o = O.HNSW(metric, rows.shape[1], seed=seed, M=cfg.get("M", 16), maxM0=cfg.get("MaxM0", 0),
               efConstruction=cfg.get("EfConstruction", 200), efSearch=cfg.get("EfSearch", 100), maxLevel=cfg.get("MaxLevel", 16))
    for i, r in enumerate(rows):
        h.Insert(f"v{i}", r)
        o.insert(r)
    return h, o


@pytest.mark.parametrize("metric", [6, 0, 5])
def test_hnsw_graph_and_search_identical_to_oracle(metric):
    """bit-identical distances => the host-driven graph (one device batch per hop) is the
    SAME graph the CPU oracle builds from the same seed, and searches return the same
    rows, distances and order"""
    rows = O.gen_rows(81 + metric, 0, 400, 48)
    h, o = _build_both(metric, rows, seed=11, EfConstruction=40, EfSearch=32)
    assert h.nodes() == o.nodes() == 400 and h.entry_point() == o.entry_point()
    for n in range(400):
        assert h.node_level(n) == o.node_level(n)
        for l in range(h.node_level(n) + 1):
            assert np.array_equal(h.links(n, l), o.links(n, l)), (n, l)
    for q in O.gen_rows(91, 0, 10, 48):
        res = h.Search(q, 10)
        er, ed = o.search(q, 10)
        assert [r.VectorIndex for r in res] == er.tolist()
        assert np.array_equal(_bits([r.Distance for r in res]), _bits(ed))
        assert [r.VectorID for r in res] == [f"v{i}" for i in er]
    assert h.distance_evals() > h.distance_calls() > 0            # batched: many evaluations per device call


def test_hnsw_delete_topup_identical_to_oracle():
    rows = O.gen_rows(95, 0, 150, 16)
    h, o = _build_both(6, rows, seed=5, EfConstruction=30, EfSearch=20)
    for n in range(0, 150, 3):
        h.Delete(f"v{n}")
        assert o.delete(n) == 0
    assert h.Size() == o.size() == 100
    from quiver_amd._host import GoError
    with pytest.raises(GoError, match="vector with ID v0 not found"):
        h.Delete("v0")
    for q in O.gen_rows(96, 0, 5, 16):
        res = h.Search(q, 100)                                    # under-filled -> exact top-up (hnsw.go:676-710)
        er, ed = o.search(q, 100)
        assert len(res) == 100
        assert np.array_equal(_bits([r.Distance for r in res]), _bits(ed))
        assert sorted(r.VectorIndex for r in res) == sorted(er.tolist())


@pytest.mark.parametrize("kat", KATS["hnsw_properties"], ids=lambda k: k["src"])
def test_hnsw_reference_tables(kat):                             # hnsw_test.go:154-256
    from quiver_amd import hnsw
    rows = np.array(kat["rows"], np.float32)
    h = hnsw.HNSW(hnsw.Config(DistanceFunc=kat["metric"], Seed=1))
    ids = kat.get("ids", [f"v{i}" for i in range(len(rows))])
    for i, r in zip(ids, rows):
        h.Insert(i, r)
    res = h.Search(kat["query"], kat["k"])
    assert 0 < len(res) <= kat["k"]
    assert all(res[i].Distance <= res[i + 1].Distance for i in range(len(res) - 1))
    if "must_contain_row" in kat:
        assert ids[kat["must_contain_row"]] in [r.VectorID for r in res]


def test_hnsw_adapter_fill_and_negative():                       # adapter.go:41-95, 345-437; adapter_test.go:136-248
    from quiver_amd import hybrid
    rows = O.gen_rows(101, 0, 60, 8)
    a = hybrid.HNSWAdapter("euclidean", hybrid.HNSWConfig(EfConstruction=20, EfSearch=10), seed=2)
    for i, r in enumerate(rows):
        a.Insert(f"id{i}", r)
    res = a.Search(rows[6], 5)
    assert res[0].ID == "id6" and res[0].Distance == 0.0          # adapter_test.go: q == id6 -> first result id6
    res = a.Search(rows[1], 60)                                   # k == size: the fill pass guarantees k results
    assert len(res) == 60 and "id1" in [r.ID for r in res]
    neg = a.SearchWithNegative(rows[3], rows[4], 0.5, 5)
    assert len(neg) == 5 and all(neg[i].Distance <= neg[i + 1].Distance for i in range(4))
    # weight 0 -> plain results truncated to k (adapter.go:366-372)
    assert [r.ID for r in a.SearchWithNegative(rows[3], rows[4], 0.0, 5)] == [r.ID for r in a.Search(rows[3], 30)][:5]


# ------------------------------------------------------------------ core.Collection surface ---

@pytest.mark.usefixtures("placement")
def test_collection_add_search_fluent_filters():                # collection_test.go; collection.go:133-331, 637-807, 886-1108
    from quiver_amd import core, hybrid
    idx = _hybrid("cosine")
    c = core.Collection("docs", 4, idx)
    with pytest.raises(core.CoreError, match="vector ID cannot be empty"):
        c.Add("", [1, 0, 0, 0])
    with pytest.raises(core.ErrInvalidDimension, match="invalid vector dimension: expected 4, got 3"):
        c.Add("x", [1, 0, 0])
    with pytest.raises(core.ErrInvalidMetadata):
        c.Add("x", [1, 0, 0, 0], "not json")
    c.Add("x", [1, 0, 0, 0], json.dumps({"kind": "a", "n": 1}))
    with pytest.raises(core.ErrVectorAlreadyExist, match="vector with the same ID already exists: x"):
        c.Add("x", [0, 1, 0, 0])
    with pytest.raises(core.CoreError, match="no vectors provided for batch insert"):
        c.AddBatch([])
    with pytest.raises(core.ErrInvalidDimension, match="for vector bad: expected 4, got 2"):
        c.AddBatch([core.Vector("ok", [0, 1, 0, 0]), core.Vector("bad", [0, 1])])
    assert c.Count() == 1
    rows = O.gen_rows(111, 0, 200, 4)
    c.AddBatch([core.Vector(f"v{i}", rows[i], json.dumps({"kind": "a" if i % 2 else "b", "n": i})) for i in range(200)])
    assert c.Count() == 201
    # un-filtered: TopK results, Score = 1 - Distance (collection.go:763)
    q = rows[17]
    resp = c.Search(core.SearchRequest(Vector=q, TopK=5, Options=core.SearchOptions(IncludeMetadata=True)))
    assert len(resp.Results) <= 5 and resp.Metadata.IndexSize == 201
    for it in resp.Results:
        assert it.Score == float(np.float32(1.0) - np.float32(it.Distance))
    with pytest.raises(core.CoreError, match="top_k must be greater than 0"):
        c.Search(core.SearchRequest(Vector=q, TopK=0))
    with pytest.raises(core.ErrInvalidDimension):
        c.Search(core.SearchRequest(Vector=q[:3], TopK=1))


@pytest.mark.usefixtures("placement")
def test_collection_filtered_search_is_a_full_ranking():        # collection.go:679-682: searchK = Index.Size()
    from quiver_amd import core, hybrid
    rows = O.gen_rows(121, 0, 300, 8)
    ids = [f"v{i}" for i in range(300)]
    ex = hybrid.ExactIndex("cosine")                              # a plain core.Index (no BatchIndex): AddBatch falls back to Insert
    c = core.Collection("c", 8, ex)
    c.AddBatch([core.Vector(ids[i], rows[i], json.dumps({"group": i % 7, "tag": "t%d" % (i % 3)})) for i in range(300)])
    q = O.gen_rows(122, 0, 1, 8)[0]
    resp = c.FluentSearch(q).WithK(10).Filter("group", 3).Execute()
    er, ed = O.exact_search(0, rows, q, 300)                      # the full ranking the index is asked for
    want = [(ids[r], d) for r, d in zip(er, ed) if r % 7 == 3][:10]
    assert [it.ID for it in resp.Results] == [w[0] for w in want]
    assert np.array_equal(_bits([it.Distance for it in resp.Results]), _bits([w[1] for w in want]))
    assert all(json.loads(it.Metadata)["group"] == 3 for it in resp.Results)
    resp2 = c.FluentSearch(q).WithK(4).FilterIn("tag", ["t0", "t2"]).FilterGreaterThan("group", 4).Execute()
    want2 = [ids[r] for r in er if (r % 3) in (0, 2) and (r % 7) > 4][:4]
    assert [it.ID for it in resp2.Results] == want2
    # sticky builder errors (collection.go:932-1091)
    with pytest.raises(core.CoreError, match="k must be greater than 0"):
        c.FluentSearch(q).WithK(0).Filter("group", 1).Execute()
    with pytest.raises(core.CoreError, match="filter field cannot be empty"):
        c.FluentSearch(q).Filter("", 1).Execute()
    with pytest.raises(core.ErrInvalidDimension):
        c.FluentSearch(q[:2]).Execute()
    assert len(c.FluentSearch(q).WithK(1000).Execute().Results) == 300   # k clamped to Count() (:924-926)
    # update / delete keep the index and the maps in step (collection.go:356-465)
    c.Update("v5", vector=q)
    assert c.FluentSearch(q).WithK(1).Execute().Results[0].ID == "v5"
    c.Delete("v5")
    with pytest.raises(core.ErrVectorNotFound):
        c.Get("v5")
    c.DeleteBatch(["v6", "v7"])
    assert c.Count() == 297


# ------------------------------------------------------------------ Arrow columnar load -> device ---

def test_arrow_ipc_load_to_device_and_search(tmp_path):          # index/arrow_hnsw_test.go:33-60; arrow_hnsw_property_test.go:142-171
    import pyarrow as pa
    from quiver_amd import arrowindex as ai
    rows = O.gen_rows(131, 0, 700, 32)
    ids = [f"id{i}" for i in range(700)]
    p = str(tmp_path / "idx.arrow")
    ai.save_ipc(p, ids, rows)
    idx = ai.ArrowFlatIndex(32)
    idx.Load(p)                                                   # FixedSizeList child buffer -> qv_index_add, no per-row work
    assert idx.Len() == 700
    q = O.gen_rows(132, 0, 1, 32)[0]
    res = idx.Search(pa.array(q, type=pa.float32()), 10)
    er, ed = O.exact_search(8, rows, q, 10)                       # metric 8 = arrow_hnsw.go:124-132 re-score
    assert [r.ID for r in res] == [ids[i] for i in er]
    assert np.array_equal(_bits([r.Distance for r in res]), _bits(ed))
    p2 = str(tmp_path / "copy.arrow")
    idx.Save(p2)                                                  # round trip (arrow_hnsw_test.go:33-60)
    idx2 = ai.ArrowFlatIndex(32)
    idx2.Load(p2)
    assert [r.ID for r in idx2.Search(q, 10)] == [r.ID for r in res]
    with pytest.raises(ValueError, match="already exists"):
        idx2.Add(rows[0], "id0")
    with pytest.raises(ValueError, match="dimension mismatch: got 3 want 32"):
        idx2.Add(np.zeros(3, np.float32), "new")
    with pytest.raises(ValueError, match="k must be positive"):
        idx2.Search(q, 0)
    axes = ai.ArrowFlatIndex(3)                                   # arrow_hnsw_property_test.go:142-171: unit axes -> 1.0
    for i, name in enumerate("abc"):
        axes.Add(np.eye(3, dtype=np.float32)[i], name)
    assert all(abs(r.Distance - 1.0) < 1e-6 for r in axes.Search(np.zeros(3, np.float32), 3))


# ------------------------------------------------------------------ device-resident HNSW traversal ---

@pytest.mark.parametrize("metric", [6, 0, 5, 3])
@pytest.mark.parametrize("max_level", [16, 1])
def test_hnsw_device_traversal_identical_to_oracle(metric, max_level):
    """qv_graph_search: whole queries walked on the GPU, one wavefront each — same rows, order,
    float32 bits and distance-evaluation counts as the CPU restatement of hnsw.go:602-713.
    With the reference's default levels many level-0 searches under-fill (multi-level nodes keep
    only self-links below their top level, hnsw.go:463-467) and are topped up by a brute-force pass
    (hnsw.go:676-710): those queries come back through the host path.  MaxLevel=1 gives a
    single-layer graph where that never happens, so every query stays on the device."""
    rows = O.gen_rows(141 + metric, 0, 3000, 64)
    h, o = _build_both(metric, rows, seed=21, EfConstruction=60, EfSearch=48, MaxLevel=max_level)
    qs = O.gen_rows(151, 0, 100, 64)
    res, ev = h.SearchBatch(qs, 10, with_evals=True)
    full = 0
    for i, q in enumerate(qs):
        er, ed, ne = o.search(q, 10, with_evals=True)
        assert [r.VectorIndex for r in res[i]] == er.tolist(), i
        assert np.array_equal(_bits([r.Distance for r in res[i]]), _bits(ed))
        if ne < 3000:                          # no brute-force top-up in the oracle: evaluation counts must agree
            full += 1
            assert int(ev[i]) == ne - 1        # the reference also evaluates the entry point once more up front (hnsw.go:637)
    assert h.device_fallbacks() == 0 and full + h.topups() == 100 and full > 0
    if max_level == 1:
        assert h.topups() == 0
    one = h.Search(qs[0], 10)                  # the host-driven traversal agrees too
    assert [r.VectorIndex for r in one] == [r.VectorIndex for r in res[0]]


def test_hnsw_device_traversal_after_deletes_falls_back_for_topup():
    rows = O.gen_rows(161, 0, 400, 16)
    h, o = _build_both(6, rows, seed=3, EfConstruction=30, EfSearch=20)
    for n in range(0, 400, 2):
        h.Delete(f"v{n}"); o.delete(n)
    qs = O.gen_rows(162, 0, 20, 16)
    res = h.SearchBatch(qs, 150)                                  # k > reachable: under-filled -> host top-up (hnsw.go:676-710)
    for i, q in enumerate(qs):
        er, ed = o.search(q, 150)
        assert len(res[i]) == len(er) == 150
        assert np.array_equal(_bits([r.Distance for r in res[i]]), _bits(ed))
        assert sorted(r.VectorIndex for r in res[i]) == sorted(er.tolist())
    res5 = h.SearchBatch(qs, 5)
    for i, q in enumerate(qs):
        er, ed = o.search(q, 5)
        assert [r.VectorIndex for r in res5[i]] == er.tolist()


def test_hnsw_device_traversal_with_equal_distances_uses_the_exact_heap_kernel():
    """duplicate vectors give exactly equal distances; with ties the reference's pop order depends
    on its binary heaps' layout, so the wave-resident kernel flags those queries and the exact-heap
    kernel re-runs them: still identical to the CPU restatement"""
    rows = O.gen_rows(171, 0, 1500, 32)
    rows[100:600] = rows[1000:1500]                 # 500 exact duplicates
    h, o = _build_both(6, rows, seed=9, EfConstruction=40, EfSearch=64, MaxLevel=1)
    qs = O.gen_rows(172, 0, 60, 32)
    qs[:10] = rows[100:110]                         # queries equal to duplicated rows: distance-0 ties
    res = h.SearchBatch(qs, 10)
    for i, q in enumerate(qs):
        er, ed = o.search(q, 10)
        assert [r.VectorIndex for r in res[i]] == er.tolist(), i
        assert np.array_equal(_bits([r.Distance for r in res[i]]), _bits(ed))
    assert h.device_fallbacks() == 0


# ------------------------------------------------------------------ round 2: batched build, row reuse, locking ---

def _host_graph(h):
    return [(h.node_level(i), [h.links(i, l).tolist() for l in range(h.node_level(i) + 1)]) for i in range(h.nodes())], h.entry_point()


def _oracle_graph(o):
    return [(o.node_level(i), [o.links(i, l).tolist() for l in range(o.node_level(i) + 1)]) for i in range(o.nodes())], o.entry_point()


@pytest.mark.parametrize("batch_max,ramp_div", [(1, 0), (64, 8)])
def test_hnsw_insert_batch_builds_on_device_and_equals_oracle(batch_max, ramp_div):
    """HNSW.InsertBatch: n Inserts connected by qv_graph_insert; the adjacency pulled back to the host equals the oracle's
    (sequential Insert for batch_max = 1, qvo_hnsw_insert_batch otherwise), and both search paths agree with it"""
    from quiver_amd import hnsw
    from quiver_amd.device_index import graph_batch_size
    n, dim = 1500, 32
    rows = O.gen_rows(515, 0, n, dim)
    h = hnsw.HNSW(hnsw.Config(M=8, EfConstruction=40, EfSearch=48, MaxLevel=6, DistanceFunc="hnsw_cosine", Seed=17))
    ids = ["n%d" % i for i in range(n)]
    h.InsertBatch(ids[:900], rows[:900], batch_max, ramp_div)
    h.InsertBatch(ids[900:], rows[900:], batch_max, ramp_div)         # a second call continues the same device graph
    assert h.built_on_device() and h.Size() == n
    o = O.HNSW(5, dim, M=8, efConstruction=40, efSearch=48, maxLevel=6, seed=17)
    done = 0
    for seg in (900, n):
        while done < seg:
            b = min(graph_batch_size(done, batch_max, ramp_div), seg - done)
            o.insert_batch(rows[done:done + b]); done += b
    assert _host_graph(h) == _oracle_graph(o)
    qs = O.gen_rows(516, 0, 24, dim)
    batch = h.SearchBatch(qs, 5)
    for i, q in enumerate(qs):
        ro, do = o.search(q, 5)
        single = h.Search(q, 5)                                       # host-driven walk over the pulled-back adjacency
        assert [r.VectorIndex for r in single] == ro.tolist() and np.array_equal(_bits([r.Distance for r in single]), _bits(do))
        assert [r.VectorIndex for r in batch[i]] == ro.tolist() and np.array_equal(_bits([r.Distance for r in batch[i]]), _bits(do))
    # a host-driven Insert and a Delete afterwards still work (and end the device-built state) ...
    extra = O.gen_rows(517, 0, 1, dim)[0]
    h.Insert("extra", extra); o.insert(extra)
    h.Delete("n17"); o.delete(17)
    assert not h.built_on_device() and _host_graph(h) == _oracle_graph(o)
    # ... and the next InsertBatch uploads the host graph, has the device score its links (qv_graph_make_buildable) and goes on
    # building on the device: still the oracle's graph
    more = O.gen_rows(518, 0, 300, dim)
    h.InsertBatch(["m%d" % i for i in range(300)], more, batch_max, ramp_div)
    done = 0
    while done < 300:
        b = min(graph_batch_size(n + 1 + done, batch_max, ramp_div), 300 - done)
        o.insert_batch(more[done:done + b]); done += b
    assert h.built_on_device() and h.Size() == n + 300 and _host_graph(h) == _oracle_graph(o)


def test_hnsw_insert_batch_duplicate_id_is_rejected_before_anything_changes():
    from quiver_amd import hnsw
    from quiver_amd._host import GoError
    rows = O.gen_rows(9, 0, 10, 8)
    h = hnsw.HNSW(hnsw.Config(M=4, DistanceFunc="hnsw_euclidean"))
    h.InsertBatch(["a%d" % i for i in range(6)], rows[:6])
    with pytest.raises(GoError, match="vector with ID a3 already exists"):
        h.InsertBatch(["b0", "a3"], rows[6:8])
    assert h.Size() == 6 and h.nodes() == 6


@pytest.mark.usefixtures("placement")
def test_hybrid_insert_batch_is_two_device_calls_and_searches_agree():
    from quiver_amd import hybrid, hnsw
    n, dim = 1200, 24
    rows = O.gen_rows(77, 0, n, dim)
    idx = hybrid.HybridIndex(hybrid.IndexConfig(DistanceFunc="euclidean", HNSWConfig=hnsw.Config(M=8, EfConstruction=60, EfSearch=64), ExplorationFactor=0.0))
    idx.InsertBatch({"v%d" % i: rows[i] for i in range(n)})
    from quiver_amd._host import hlib
    assert hlib().qvh_hybrid_hnsw_built_on_device(idx._h) == 1
    qs = O.gen_rows(78, 0, 16, dim)
    resp = idx.BatchSearch(hybrid.BatchSearchRequest(Queries=list(qs), K=5, ForceStrategy="hnsw"))     # one graph-traversal call
    for i, q in enumerate(qs):
        one = idx.SearchWithRequest(hybrid.HybridSearchRequest(Query=q, K=5, ForceStrategy="hnsw"))
        assert [(r.ID, r.Distance) for r in resp.Results[i]] == [(r.ID, r.Distance) for r in one.Results]
    ex = idx.BatchSearch(hybrid.BatchSearchRequest(Queries=list(qs), K=5, ForceStrategy="exact"))
    for i, q in enumerate(qs):
        er, ed = O.exact_search(1, rows, q, 5)
        assert [r.ID for r in ex.Results[i]] == ["v%d" % j for j in er]


@pytest.mark.usefixtures("placement")
def test_exact_index_reuses_tombstoned_rows_under_churn():
    """Collection.Update = Delete + Insert (collection.go): the device index must not grow with every update"""
    from quiver_amd import hybrid
    from quiver_amd._host import hlib
    rows = O.gen_rows(3, 0, 600, 16)
    e = hybrid.ExactIndex("cosine")
    for i in range(200):
        e.Insert("v%d" % i, rows[i])
    for rnd in range(5):
        for i in range(0, 200, 2):
            e.Delete("v%d" % i)
        for i in range(0, 200, 2):
            e.Insert("v%d" % i, rows[200 + (rnd * 100 + i // 2) % 400])
    assert e.Size() == 200 and hlib().qvh_exact_device_rows(e._h) == 200
    live = {("v%d" % i): (rows[i] if i % 2 else rows[200 + (4 * 100 + i // 2) % 400]) for i in range(200)}
    q = O.gen_rows(4, 0, 1, 16)[0]
    got = e.Search(q, 7)
    names = list(live.keys()); mat = np.stack([live[n_] for n_ in names])
    er, ed = O.exact_search(0, mat, q, 7)
    assert np.array_equal(_bits([r.Distance for r in got]), _bits(ed))
    assert len(set(ed.tolist())) < 7 or [r.ID for r in got] == [names[j] for j in er]


@pytest.mark.usefixtures("placement")
def test_concurrent_searches_on_one_hybrid_index_are_consistent():
    """Collection.Search holds only a read lock (collection.go:647): many Index.Search calls run at once.  ctypes drops the GIL,
    so Python threads really do enter the C++ mirror concurrently."""
    import threading
    from quiver_amd import hybrid, hnsw
    n, dim = 800, 20
    rows = O.gen_rows(21, 0, n, dim)
    idx = hybrid.HybridIndex(hybrid.IndexConfig(DistanceFunc="cosine", HNSWConfig=hnsw.Config(M=8, EfConstruction=40, EfSearch=40), ExplorationFactor=0.0))
    for i in range(n):
        idx.Insert("v%d" % i, rows[i])                                # host-driven graph: Search walks it with per-thread visited stamps
    qs = O.gen_rows(22, 0, 40, dim)

    def serial(force):
        return [[(r.ID, r.Distance) for r in idx.SearchWithRequest(hybrid.HybridSearchRequest(Query=q, K=6, ForceStrategy=force)).Results] for q in qs]
    want = {"hnsw": serial("hnsw"), "exact": serial("exact")}
    errors = []

    def worker(force, rounds):
        try:
            for _ in range(rounds):
                if serial(force) != want[force]:
                    errors.append("mismatch in " + force)
        except Exception as ex:                                       # noqa: BLE001
            errors.append(repr(ex))
    ts = [threading.Thread(target=worker, args=(f, 3)) for f in ("hnsw", "exact", "hnsw", "exact", "hnsw", "hnsw")]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:3]


def test_hnsw_degree_bound_above_64_searches_through_the_host_walk():
    """M = 48 -> MaxM0 = 96: the device traversal does not take it; SearchBatch must answer through the host-driven Search
    (like k > 512), not fail"""
    from quiver_amd import hnsw
    rows = O.gen_rows(61, 0, 300, 12)
    h = hnsw.HNSW(hnsw.Config(M=48, EfConstruction=60, EfSearch=40, MaxLevel=1, DistanceFunc="hnsw_euclidean", Seed=2))
    o = O.HNSW(6, 12, M=48, efConstruction=60, efSearch=40, maxLevel=1, seed=2)
    for i in range(300):
        h.Insert("p%d" % i, rows[i]); o.insert(rows[i])
    qs = O.gen_rows(62, 0, 5, 12)
    out = h.SearchBatch(qs, 4)
    for i, q in enumerate(qs):
        ro, do = o.search(q, 4)
        assert [r.VectorIndex for r in out[i]] == ro.tolist()
    assert h.device_fallbacks() == 5


# ------------------------------------------------------------------ round 2: the remaining reference tables ---

@pytest.mark.parametrize("kat", KATS["hnsw_adapter_tables"], ids=lambda k: k["src"])
def test_hnsw_adapter_reference_table(kat):                      # adapter_test.go:136-248
    from quiver_amd import hybrid
    from quiver_amd._host import GoError
    a = hybrid.HNSWAdapter("hnsw_cosine", hybrid.HNSWConfig(M=kat["config"]["M"], EfConstruction=kat["config"]["EfConstruction"]), seed=3)
    for id_, row in zip(kat["ids"], kat["rows"]):
        a.Insert(id_, row)
    for case in kat["cases"]:
        if "want_error" in case:
            with pytest.raises(GoError, match=case["want_error"]):
                a.Search(case["query"], case["k"])
            continue
        res = a.Search(case["query"], case["k"])
        assert len(res) > 0
        if "must_contain" in case:
            assert case["must_contain"] in [r.ID for r in res]
        if "want_first" in case:
            assert res[0].ID == case["want_first"]
        if "max_results" in case:
            assert case["min_results"] <= len(res) <= case["max_results"]


@pytest.mark.parametrize("kat", KATS["hnsw_edge_cases"], ids=lambda k: k["src"])
def test_hnsw_edge_case_tables(kat):                             # hnsw_property_test.go:397-462
    from quiver_amd import hnsw
    from quiver_amd._host import GoError
    h = hnsw.HNSW(hnsw.Config(DistanceFunc=kat["metric"]))
    if "want_insert_error" in kat:
        h.Insert(*kat["inserts"][0])
        with pytest.raises(GoError, match=kat["want_insert_error"]):
            h.Insert(*kat["inserts"][1])
        return
    for id_, v in kat["inserts"]:
        h.Insert(id_, v)
    for id_ in kat["deletes"]:
        h.Delete(id_)
    if "want_error" in kat:
        with pytest.raises(GoError, match=kat["want_error"]):
            h.Search(kat["query"], kat["k"])
    else:
        assert len(h.Search(kat["query"], kat["k"])) == kat["want_count"]


@pytest.mark.parametrize("kat", KATS["arrow_graph"], ids=lambda k: k["src"])
def test_arrow_graph_search_table(kat):                          # arrowindex/graph_test.go:10-28
    from quiver_amd import arrowindex
    g = arrowindex.Graph(kat["dim"], kat["m"], kat["efConstruction"], kat["efSearch"])
    for id_, v in kat["adds"]:
        g.Add(id_, v)
    assert g.Len() == len(kat["adds"]) <= g.m                    # the exhaustive branch (graph.go:482-484)
    assert g.Search(kat["query"], kat["k"]) == kat["want_ids"]
    assert g.Search(kat["query"], 100) == kat["want_ids"]        # k is clamped to the node count (:478-480)
    with pytest.raises(ValueError, match="query dimension mismatch: got 3, want 2"):
        g.Search([0.0, 0.0, 0.0], 1)
    assert arrowindex.Graph(2).Search([0.0, 0.0], 3) == []       # empty graph: nil, nil (:474-476)
    with pytest.raises(ValueError, match="not float32-representable"):
        g.Add(9, [0.1, 0.2])
    with pytest.raises(ValueError, match="not float32-representable"):
        g.Search([0.1, 0.2], 1, strict=True)                     # strict: a query is refused by the same rule as a vector
    with pytest.warns(RuntimeWarning, match="rounded to float32"):
        g.Search([0.1, 0.2], 1)                                  # default: rounded, and said so (the reference's own table queries need it)


def test_persistence_collection_search_tables():                 # persistence/collection_test.go:259-325, 355-383
    from quiver_amd import persistence as ps
    kat, sort_kat = KATS["persistence_collection"]
    c = ps.Collection("test", kat["dimension"], kat["metric"])
    for id_, v in kat["vectors"]:
        c.AddVector(id_, v, None)
    for case in kat["cases"]:
        res = c.Search(kat["query"], case["limit"])
        assert len(res) == case["want_count"]
        assert all(res[i].Distance <= res[i + 1].Distance for i in range(len(res) - 1))
        if "want_first" in case:
            assert res[0].ID == case["want_first"]
    with pytest.raises(ps.GoError, match=kat["bad_query_error"]):
        c.Search(kat["bad_query"], 2)
    rs = [ps.SearchResult(i, d) for i, d in sort_kat["sort_in"]]
    ps.SortSearchResults(rs)
    assert [r.ID for r in rs] == sort_kat["sort_want"]
    # against the oracle on random data, incl. overwrite-on-AddVector, delete + row reuse, limit <= 0 = full ranking
    rows = O.gen_rows(13, 0, 400, 24)
    c2 = ps.Collection("r", 24, "cosine")
    for i in range(300):
        c2.AddVector("v%d" % i, rows[i], {"n": str(i)})
    c2.AddVector("v7", rows[350])                                # overwrite in place
    c2.DeleteVector("v9"); c2.AddVector("w", rows[351])          # the tombstoned row is reused
    live = {("v%d" % i): rows[i] for i in range(300) if i not in (7, 9)}
    live["v7"] = rows[350]; live["w"] = rows[351]
    names = list(live); mat = np.stack([live[n_] for n_ in names])
    q = O.gen_rows(14, 0, 1, 24)[0]
    full = c2.Search(q, 0)
    assert len(full) == c2.Count() == 300
    want = np.sort(O.all_distances(0, mat, q))
    assert np.array_equal(_bits([r.Distance for r in full]), _bits(want))
    top = c2.Search(q, 5)
    er, ed = O.exact_search(0, mat, q, 5)
    assert np.array_equal(_bits([r.Distance for r in top]), _bits(ed)) and (len(set(ed.tolist())) < 5 or [r.ID for r in top] == [names[j] for j in er])
    with pytest.raises(ps.GoError, match="vector with ID nope not found"):
        c2.DeleteVector("nope")


def test_hnsw_search_wider_than_the_device_traversal_uses_one_distance_table():
    """k above 512 (a filtered Collection.Search over an HNSW-backed collection asks k = Size(), collection.go:679-682 ->
    adapter.go:41-52): the host-driven walk takes its distances from ONE device call over every row instead of one call per hop —
    same results as the oracle's walk, and far fewer device calls than hops"""
    import time
    from quiver_amd import hnsw
    n, dim = 2500, 24
    rows = O.gen_rows(71, 0, n, dim)
    h = hnsw.HNSW(hnsw.Config(M=8, EfConstruction=60, EfSearch=50, MaxLevel=2, DistanceFunc="hnsw_cosine", Seed=5))
    o = O.HNSW(5, dim, M=8, efConstruction=60, efSearch=50, maxLevel=2, seed=5)
    for i in range(n):                                                # node by node on both sides: the same graph
        h.Insert("p%05d" % i, rows[i]); o.insert(rows[i])            # (zero-padded ids: the top-up's (Distance, VectorID) order, hnsw.go:699-704, compares
                                                                      #  the id STRINGS — "p1765" < "p24" — and the oracle's ids are the indices)
    qs = O.gen_rows(72, 0, 3, dim)
    for k in (600, n):
        t0 = time.perf_counter()
        out = h.SearchBatch(qs, k)
        dt = time.perf_counter() - t0
        for i, q in enumerate(qs):
            ro, do = o.search(q, k)
            assert [r.VectorIndex for r in out[i]] == ro.tolist(), (k, i)
            assert np.array_equal(np.array([r.Distance for r in out[i]], np.float32).view(np.uint32), do.view(np.uint32))
        assert dt < 20.0
