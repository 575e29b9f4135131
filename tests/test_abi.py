"""CPU-side checks of the drop-in boundary: libqv.so loads, exports every symbol
include/qv.h declares, agrees with the Python prototypes, and refuses to compute
without a GPU (no fallback).  No compute calls here."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "qv.h")


def _declared_symbols():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(qv_[a-z_0-9]+)\s*\(", src)))


def test_header_declares_the_expected_surface():
    syms = _declared_symbols()
    for must in ("qv_index_create", "qv_index_add", "qv_index_remove", "qv_index_search", "qv_distance_rows",
                 "qv_distance_pairs", "qv_index_size", "qv_index_destroy", "qv_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from quiver_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "libqv.so not built: python -c 'import __graft_entry__ as g; g.build()'"
    handle = C.CDLL(_lib.LIB_PATH)
    for s in _declared_symbols():
        assert hasattr(handle, s), f"{s} declared in include/qv.h but not exported"


def test_python_prototypes_cover_the_header():
    from quiver_amd import _lib
    assert sorted(_lib.PROTOTYPES) == _declared_symbols()
    assert _lib.lib().qv_abi_version() == 4


def test_header_compiles_as_c_and_cpp(tmp_path):
    c = tmp_path / "t.c"
    c.write_text('#include "qv.h"\nint main(void){return QV_ABI_VERSION - 4;}\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(c), "-o", str(tmp_path / "t.o")])
    cpp = tmp_path / "t.cpp"
    cpp.write_text('#include "qv.h"\nint main(){return QV_ABI_VERSION - 4;}\n')
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(cpp), "-o", str(tmp_path / "t2.o")])


def test_no_torch_or_cxx_types_in_the_abi():
    src = re.sub(r"/\*.*?\*/", "", open(HDR).read(), flags=re.S)   # declarations only, comments stripped
    assert "torch" not in src and "std::" not in src and "at::" not in src and "Tensor" not in src


def test_product_never_touches_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/"""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "quiver_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".c")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"(qv_oracle|oracle_np|libqvoracle|qvo_)", txt):
                    bad.append(os.path.join(base, f))
    assert not bad, bad
    assert "qvo_" not in open(HDR).read()


def test_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import quiver_amd as q
    with pytest.raises(q.QvError) as e:
        q.DeviceIndex(8)
    assert e.value.code == -5 and "no CPU path" in str(e.value)
    import numpy as np
    from quiver_amd.device_index import distance_pairs
    with pytest.raises(q.QvError):
        distance_pairs("cosine", np.ones((1, 3), np.float32), np.ones((1, 3), np.float32))


def _build_c_smoke(tmp_path):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.join(root, "quiver_amd", "lib")
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "c", "abi_smoke.c"),
                    "-L", libdir, "-lqv", "-lm", "-Wl,-rpath," + libdir, "-o", exe], check=True, capture_output=True, text=True)
    return exe


def test_header_is_plain_c_and_the_library_fails_loudly_without_a_gpu(tmp_path):
    """include/qv.h compiled by a C11 compiler (what a cgo preamble does), linked against libqv.so alone; on a box
    without a GPU every entry point refuses with QV_ERR_NO_DEVICE — there is no CPU path to fall back to"""
    import subprocess
    import quiver_amd
    exe = _build_c_smoke(tmp_path)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    if quiver_amd.lib().qv_device_count() <= 0:
        assert "no device" in p.stdout and "no CPU path" in p.stdout
    else:
        assert p.stdout.startswith("ok:")


def test_sharded_placement_rule_and_span():
    """pure host arithmetic of qv_sharded_* (no device needed): the id space per shard and how a batch is cut over the shards"""
    import numpy as np
    from quiver_amd import _lib
    from quiver_amd.device_index import sharded_plan_add
    L = _lib.lib()
    assert L.qv_sharded_span(8) == (1 << 29) - 64 and L.qv_sharded_span(3) % 64 == 0 and 3 * L.qv_sharded_span(3) <= 1 << 32
    assert L.qv_sharded_span(1) % 64 == 0 and L.qv_sharded_span(1) > (1 << 31)
    assert sharded_plan_add([0, 0, 0, 0], 10).tolist() == [3, 3, 2, 2]
    assert sharded_plan_add([10, 0, 5], 9).tolist() == [0, 8, 1]
    assert sharded_plan_add([100, 0, 0], 1).tolist() == [0, 1, 0]           # a single Insert goes to an emptiest shard
    for have, n in (([7, 7, 7], 30), ([1000, 10, 10, 10], 5), ([0], 12345)):
        give = sharded_plan_add(have, n)
        assert int(give.sum()) == n
        after = np.array(have) + give
        assert after.max() - after.min() <= max(1, max(have) - min(have))
