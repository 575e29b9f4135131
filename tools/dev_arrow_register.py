"""Arrow ingest without the staging copy?  (the survey's "next" #3 / index/arrow_hnsw.go:201-241)
A record batch's FixedSizeList<float32> child buffer is pageable host memory (a memory-mapped IPC file).  Today qv_index_add copies it
to a device staging buffer (hipMemcpy, <= 256 MiB at a time) and runs the ingest kernel (tile transpose + norms).  The alternative:
hipHostRegister the Arrow buffer, hand the ingest kernel its device-visible address (qv_index_add_device) — one pass over PCIe, no
staging buffer — and unregister.  This measures both, and registration on its own.
    python tools/dev_arrow_register.py [rows=500000] [dim=768]"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import quiver_amd

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
torch.cuda.init()
hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
x = np.random.default_rng(0).standard_normal((rows, dim), dtype=np.float32)
nbytes = x.nbytes

def staged():
    idx = quiver_amd.DeviceIndex(dim, "cosine"); idx.reserve(rows)
    t0 = time.perf_counter(); idx.add(x); dt = time.perf_counter() - t0
    idx.close(); return dt

def registered():
    idx = quiver_amd.DeviceIndex(dim, "cosine"); idx.reserve(rows)
    t0 = time.perf_counter()
    rc = hip.hipHostRegister(C.c_void_p(x.ctypes.data), C.c_size_t(nbytes), C.c_uint(2))       # hipHostRegisterMapped
    t1 = time.perf_counter()
    if rc != 0:
        idx.close(); return None
    dptr = C.c_void_p()
    assert hip.hipHostGetDevicePointer(C.byref(dptr), C.c_void_p(x.ctypes.data), C.c_uint(0)) == 0
    idx.add_device(dptr.value, rows)                                                           # the ingest kernel reads the host buffer over PCIe
    t2 = time.perf_counter()
    hip.hipHostUnregister(C.c_void_p(x.ctypes.data))
    t3 = time.perf_counter()
    chk = idx.get_row(rows - 1)
    idx.close()
    assert np.array_equal(chk, x[rows - 1])
    return t1 - t0, t2 - t1, t3 - t2

staged(); s = min(staged() for _ in range(3))
r = registered(); r = registered()
out = {"workload": "%d x %d float32 rows from pageable host memory (%.2f GB)" % (rows, dim, nbytes / 1e9),
       "staged_copy_then_ingest_s": round(s, 4), "staged_GBps": round(nbytes / s / 1e9, 1)}
if r:
    out.update({"hipHostRegister_s": round(r[0], 4), "ingest_reading_the_registered_buffer_s": round(r[1], 4), "hipHostUnregister_s": round(r[2], 4),
                "registered_total_s": round(sum(r), 4), "registered_GBps_incl_registration": round(nbytes / sum(r) / 1e9, 1),
                "registered_GBps_ingest_only": round(nbytes / r[1] / 1e9, 1)})
else:
    out["hipHostRegister"] = "refused"
print(json.dumps(out))
