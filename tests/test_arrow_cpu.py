"""Arrow IPC plumbing without a GPU: the schema the reference writes
(index/arrow_hnsw.go:153-156), zero-copy access to the FixedSizeList child buffer
(arrow_hnsw.go:222-225), multi-batch files, offsets."""
import numpy as np
import pyarrow as pa
import pyarrow.ipc as ipc
import pytest

from quiver_amd import arrowindex as ai


def test_schema_matches_reference():
    s = ai.schema_for(8)
    assert s.names == ["id", "vector"]
    assert pa.types.is_string(s.field("id").type)
    t = s.field("vector").type
    assert pa.types.is_fixed_size_list(t) and t.list_size == 8 and pa.types.is_float32(t.value_type)


def test_save_load_roundtrip_and_zero_copy(tmp_path):
    rng = np.random.default_rng(0)
    v = rng.standard_normal((300, 16)).astype(np.float32)
    ids = [f"id{i}" for i in range(300)]
    p = str(tmp_path / "a.arrow")
    ai.save_ipc(p, ids, v)
    got_blocks, got_ids = [], []

    def sink(vals, bids):
        assert vals.dtype == np.float32 and vals.flags["C_CONTIGUOUS"] and not vals.flags["OWNDATA"]   # a view of the Arrow buffer
        got_blocks.append(vals.copy())
        got_ids.extend(bids)

    all_ids = ai.load_ipc(p, 16, sink)
    assert all_ids == ids == got_ids
    assert np.array_equal(np.concatenate(got_blocks), v)


def test_multi_batch_file_and_sliced_batches(tmp_path):
    rng = np.random.default_rng(1)
    v = rng.standard_normal((100, 4)).astype(np.float32)
    ids = [f"k{i}" for i in range(100)]
    p = str(tmp_path / "m.arrow")
    arr = pa.FixedSizeListArray.from_arrays(pa.array(v.reshape(-1), type=pa.float32()), 4)
    full = pa.record_batch([pa.array(ids), arr], schema=ai.schema_for(4))
    with ipc.new_file(p, ai.schema_for(4)) as w:
        w.write_batch(full.slice(0, 33))                     # sliced batches carry a list offset
        w.write_batch(full.slice(33, 50))
        w.write_batch(full.slice(83))
    blocks = []
    got = ai.load_ipc(p, 4, lambda vals, bids: blocks.append(vals.copy()))
    assert got == ids and [b.shape[0] for b in blocks] == [33, 50, 17]
    assert np.array_equal(np.concatenate(blocks), v)
    assert np.array_equal(ai.batch_values(full.slice(10, 5), 4), v[10:15])


def test_wrong_schema_is_rejected(tmp_path):
    p = str(tmp_path / "bad.arrow")
    arr = pa.FixedSizeListArray.from_arrays(pa.array(np.zeros(12, np.float64)), 4)
    sch = pa.schema([pa.field("id", pa.string()), pa.field("vector", pa.list_(pa.float64(), 4))])
    with ipc.new_file(p, sch) as w:
        w.write_batch(pa.record_batch([pa.array(["a", "b", "c"]), arr], schema=sch))
    with pytest.raises(ValueError, match="FixedSizeList<float32>"):
        ai.load_ipc(p, 4, lambda *_: None)
