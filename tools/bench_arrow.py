"""Arrow IPC -> device ingest rate (north star: "pkg/arrowindex columnar load -> device";
index/arrow_hnsw.go:201-241).  python tools/bench_arrow.py [--rows 500000] [--dim 768] [--batch-rows 50000]

Writes an IPC file with the reference's schema {id: utf8, vector: FixedSizeList<float32>[dim]} in record batches,
then times (a) the vector path alone: memory-mapped child buffer -> qv_index_add (H2D + tile transpose + norms),
and (b) ArrowFlatIndex.Load including the id bookkeeping."""
import argparse, json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyarrow as pa, pyarrow.ipc as ipc
import quiver_amd
from quiver_amd import arrowindex as A

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=500_000); ap.add_argument("--dim", type=int, default=768); ap.add_argument("--batch-rows", type=int, default=50_000)
a = ap.parse_args()
path = os.path.join(tempfile.gettempdir(), "qv_bench_arrow.ipc")
rng = np.random.default_rng(0)
with ipc.new_file(path, A.schema_for(a.dim)) as w:
    for s in range(0, a.rows, a.batch_rows):
        n = min(a.batch_rows, a.rows - s)
        v = rng.standard_normal((n, a.dim), dtype=np.float32)
        arr = pa.FixedSizeListArray.from_arrays(pa.array(v.reshape(-1), type=pa.float32()), a.dim)
        w.write_batch(pa.record_batch([pa.array(["v%d" % (s + i) for i in range(n)], type=pa.string()), arr], schema=A.schema_for(a.dim)))
size = os.path.getsize(path)

def vectors_only():
    idx = quiver_amd.DeviceIndex(a.dim, "arrow_squared_euclidean"); idx.reserve(a.rows)
    t0 = time.perf_counter()
    with pa.memory_map(path, "r") as src:
        r = ipc.open_file(src)
        for i in range(r.num_record_batches):
            idx.add(A.batch_values(r.get_batch(i), a.dim))        # the contiguous child buffer, as is
    dt = time.perf_counter() - t0
    assert idx.size() == a.rows
    idx.close()
    return dt
vectors_only()                                                    # page cache + allocator warm-up
t_vec = min(vectors_only() for _ in range(3))
t0 = time.perf_counter()
ai = A.ArrowFlatIndex(a.dim); ai.Load(path)
t_load = time.perf_counter() - t0
q = rng.standard_normal(a.dim, dtype=np.float32)
res = ai.Search(q, 3)
os.remove(path)
print(json.dumps({"workload": "Arrow IPC %d x %d float32 in %d-row record batches (%.2f GB file)" % (a.rows, a.dim, a.batch_rows, size / 1e9),
                  "vectors_to_device_s": t_vec, "vectors_GBps": a.rows * a.dim * 4 / t_vec / 1e9, "rows_per_s": a.rows / t_vec,
                  "ArrowFlatIndex_Load_s_incl_ids": t_load, "first_hit": [res[0].ID if hasattr(res[0], "ID") else str(res[0])]}))
