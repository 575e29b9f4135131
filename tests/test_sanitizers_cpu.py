"""Everything that runs on the HOST under the sanitizers (CPU builds only: the pool has no GPU sanitizer), `make -C tests/c san`:

  coalesce_tsan / coalesce_asan   quiver_amd/csrc/qv_coalesce.h — the futex / spin-lock front that lets concurrent callers share device
                                  passes — 1 / 8 / 64 / 256 callers x 1 / 2 / 4 lanes, with and without think time, with the early round
  host_tsan / host_asan           quiver_amd/csrc/host/qvhost.cpp (the mirror of the reference's Go callers: searches under a read lock
                                  beside mutations under the write lock, exact.go:25, hnsw.go:58, hybrid_index.go:25) against
                                  tests/c/qv_stub.cpp, a CPU stand-in for libqv answered by the oracle (test infrastructure only)
  oracle_asan                     oracle/*.c through its own known-answer tests

-fsanitize=thread and -fsanitize=address,undefined are separate builds; halt_on_error: any report fails the target.  What round 6 found
and fixed this way: the delivered[] race of the early round (qv_coalesce.h), the unsynchronised read of HNSW::dg_ in HNSW::Search
(qvhost.cpp), an uninitialised `borrowed` flag and two memcpy(…, NULL, 0) in the oracle."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _san(target, tmp_path, timeout=900):
    if not shutil.which("g++"):
        pytest.skip("no g++")
    p = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "c"), target, "OUT=%s" % tmp_path], capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, (p.stdout[-3000:] + "\n" + p.stderr[-6000:])
    return p.stdout


@pytest.mark.parametrize("target", ["coalesce_tsan", "coalesce_asan"])
def test_coalescing_front_is_clean(target, tmp_path):
    out = _san(target, tmp_path)
    assert "FAILED" not in out and "callers 256 lanes 4" in out


@pytest.mark.parametrize("target", ["host_tsan", "host_asan"])
def test_host_layer_is_clean_against_the_stub(target, tmp_path):
    out = _san(target, tmp_path)
    assert "hybrid index: 6 searchers beside a mutator, failures so far 0" in out


def test_oracle_is_clean_under_its_own_kats(tmp_path):
    out = _san("oracle_asan", tmp_path)
    assert " passed" in out
