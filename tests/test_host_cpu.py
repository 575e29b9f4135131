"""CPU-side checks of the host layer: the C++ host library loads and exports its flat
surface; pkg/core's Collection bookkeeping (validation order, sentinel errors, filters,
searchK = Size() under filters) with a canned index, the way the reference's own
collection tests use MockIndex (collection_test.go:13-80, 424-428)."""
import ctypes as C
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_library_loads_and_exports_its_surface():
    from quiver_amd import _host
    assert os.path.exists(_host.HOST_LIB_PATH)
    h = _host.hlib()
    for name in _host._SIGS:
        assert hasattr(h, name)
    # constructing host objects needs no GPU; the first device call would fail loudly
    p = h.qvh_exact_new(0, 0)
    assert h.qvh_exact_size(p) == 0
    r = h.qvh_results_new()
    assert h.qvh_exact_search(p, None, 3, 5, r) == 0 and h.qvh_results_count(r) == 0   # empty -> empty, nil error
    h.qvh_results_free(r)
    h.qvh_exact_free(p)


def test_host_insert_fails_loudly_without_gpu():
    import numpy as np
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from quiver_amd import hybrid
    from quiver_amd._host import GoError
    idx = hybrid.ExactIndex("cosine")
    with pytest.raises(GoError, match="no CPU path"):
        idx.Insert("a", np.ones(3, np.float32))


class MockIndex:                                   # collection_test.go:13-80
    def __init__(self):
        self.vectors, self.searchResults, self.asked = {}, [], []

    def Insert(self, id, v):
        self.vectors[id] = v

    def Delete(self, id):
        self.vectors.pop(id, None)

    def Search(self, v, k):
        self.asked.append(k)
        return self.searchResults[:k]

    def Size(self):
        return len(self.vectors)


class MockBatchIndex(MockIndex):
    def __init__(self):
        super().__init__()
        self.batches = 0

    def InsertBatch(self, vs):
        self.batches += 1
        self.vectors.update(vs)

    def DeleteBatch(self, ids):
        for i in ids:
            self.vectors.pop(i, None)


def test_collection_validation_order_and_sentinels():
    from quiver_amd import core
    from quiver_amd.hybrid import BasicSearchResult
    m = MockBatchIndex()
    c = core.Collection("t", 3, m)
    with pytest.raises(core.CoreError, match="vector ID cannot be empty"):
        c.Add("", [1, 2])                           # id check precedes the dimension check
    with pytest.raises(core.ErrInvalidDimension, match="expected 3, got 2"):
        c.Add("a", [1, 2], "not json")              # dimension precedes metadata
    with pytest.raises(core.ErrInvalidMetadata):
        c.Add("a", [1, 2, 3], "[1,2]")              # must be a JSON object
    c.Add("a", [1, 2, 3], json.dumps({"x": 1}))
    with pytest.raises(core.ErrVectorAlreadyExist):
        c.Add("a", [1, 2, 3])
    c.AddBatch([core.Vector("b", [0, 0, 1], json.dumps({"x": 2})), core.Vector("c", [0, 1, 0], json.dumps({"x": 3}))])
    assert m.batches == 1 and c.Count() == 3        # BatchIndex type-assert path (collection.go:267)
    with pytest.raises(core.ErrVectorAlreadyExist, match=": b"):
        c.AddBatch([core.Vector("z", [0, 0, 1]), core.Vector("b", [0, 0, 1])])
    assert "z" not in m.vectors                      # validated before anything is inserted
    assert c.Get("a").Metadata == json.dumps({"x": 1})
    with pytest.raises(core.ErrVectorNotFound):
        c.Get("nope")
    with pytest.raises(core.ErrVectorNotFound, match=": nope"):
        c.DeleteBatch(["a", "nope"])
    m.searchResults = [BasicSearchResult("a", 0.1), BasicSearchResult("b", 0.2), BasicSearchResult("c", 0.3)]
    r = c.Search(core.SearchRequest(Vector=[1, 2, 3], TopK=2))
    assert [x.ID for x in r.Results] == ["a", "b"] and m.asked[-1] == 2
    assert abs(r.Results[0].Score - 0.9) < 1e-6     # Score = 1 - Distance
    r = c.Search(core.SearchRequest(Vector=[1, 2, 3], TopK=1, Filters=[core.Filter("x", core.GreaterThan, 1)]))
    assert m.asked[-1] == 3                          # filters -> searchK = Index.Size() (collection.go:679-682)
    assert [x.ID for x in r.Results] == ["b"]
    r = c.FluentSearch([1, 2, 3]).WithK(5).FilterIn("x", [1, 3]).Execute()
    assert [x.ID for x in r.Results] == ["a", "c"]
    r = c.FluentSearch([1, 2, 3]).WithK(5).FilterNotEquals("x", 1).FilterLessThan("x", 3).Execute()
    assert [x.ID for x in r.Results] == ["b"]
    c.Update("a", metadata=json.dumps({"x": 9}))
    assert json.loads(c.Get("a").Metadata)["x"] == 9
    c.Delete("a")
    assert c.Count() == 2
    empty = core.Collection("e", 3, MockIndex())
    assert empty.Search(core.SearchRequest(Vector=[1, 2, 3], TopK=3)).Results == []   # collection.go:665-676
    with pytest.raises(core.CoreError, match="top_k must be greater than 0"):
        empty.FluentSearch([1, 2, 3]).Execute()      # k clamps to Count() = 0, then TopK <= 0 (collection.go:924-926, 659)


def test_filter_value_semantics():                  # collection.go:530-632
    from quiver_amd.core import Filter, matchesFilter
    md = {"n": 5, "s": "abc", "b": True, "f": 2.5}
    assert matchesFilter(md, Filter("n", "=", 5.0)) and not matchesFilter(md, Filter("n", "=", 6))
    assert matchesFilter(md, Filter("s", "=", "abc")) and matchesFilter(md, Filter("s", "<", "abd"))
    assert matchesFilter(md, Filter("f", ">=", 2.5)) and matchesFilter(md, Filter("f", "<=", 2.5))
    assert matchesFilter(md, Filter("b", "=", True))
    assert not matchesFilter(md, Filter("missing", "=", 1))
    assert matchesFilter(md, Filter("n", "not_in", [1, 2])) and not matchesFilter(md, Filter("n", "not_in", [5]))
    assert matchesFilter(md, Filter("n", "not_in", "oops"))       # not a list -> true (collection.go:562-571)
    assert not matchesFilter(md, Filter("n", "in", "oops"))
    assert not matchesFilter(md, Filter("n", "~", 1))


def test_persistence_collection_without_a_distance_function_still_stores_vectors():
    """pkg/persistence/collection.go:99-208: AddVector / GetVector / DeleteVector / Count do not involve distanceFunc; only Search
    errors with "distance function is not set" (:227-229).  No device index exists then (nothing to offload): runs without a GPU."""
    import numpy as np
    import pytest
    from quiver_amd import persistence as ps
    c = ps.Collection("plain", 3, None)
    c.AddVector("a", [1, 2, 3], {"k": "v"})
    c.AddVector("b", [4, 5, 6])
    assert c.Count() == 2
    v, meta = c.GetVector("a")
    assert np.array_equal(v, np.array([1, 2, 3], np.float32)) and meta == {"k": "v"}
    c.AddVector("a", [7, 8, 9])                                   # an existing id is overwritten (c.vectors[id] = vecCopy)
    assert c.Count() == 2 and np.array_equal(c.GetVector("a")[0], np.array([7, 8, 9], np.float32))
    with pytest.raises(ps.GoError, match="distance function is not set"):
        c.Search([1, 2, 3], 1)
    c.DeleteVector("a")
    assert c.Count() == 1
    with pytest.raises(ps.GoError, match="vector with ID a not found"):
        c.DeleteVector("a")
    with pytest.raises(ps.GoError, match="vector with ID a not found"):
        c.GetVector("a")
    with pytest.raises(ps.GoError, match="vector dimension mismatch: got 2, expected 3"):
        c.AddVector("z", [1, 2])
