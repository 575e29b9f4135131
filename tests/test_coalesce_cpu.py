"""quiver_amd/csrc/qv_coalesce.h on the CPU: the front that lets concurrent single-query callers (all the reference's host sends:
pkg/core/collection.go:647, pkg/core/db.go:805-828, pkg/hnsw/hnsw.go:602-606) share device passes.  The pass is a sleep here
(tests/c/coalesce_harness.cpp); what is checked is the bookkeeping: every caller gets ITS queries' results cut at ITS k, never
more than `lanes` passes run at once, a lone caller never waits, nobody is left behind."""
import ctypes as C
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    so = tmp_path_factory.mktemp("coalesce") / "libcoalesce_harness.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall", "-Werror", "-o", str(so),
                           os.path.join(ROOT, "tests", "c", "coalesce_harness.cpp")])
    lib = C.CDLL(str(so))
    lib.coalesce_harness.restype = C.c_int
    lib.coalesce_harness.argtypes = [C.c_int] + [C.c_uint] * 5 + [C.c_int, C.c_uint, C.POINTER(C.c_ulonglong)]

    def run(lanes, max_group, threads, calls, pass_us, think_us=0, two_keys=1, slow_every=7):
        out = (C.c_ulonglong * 8)()
        assert lib.coalesce_harness(lanes, max_group, threads, calls, pass_us, think_us, two_keys, slow_every, out) == 0
        return dict(zip(("solo", "led", "rode", "groups", "group_queries", "lingers", "wrong", "max_passes"), [int(x) for x in out]))
    return run


@pytest.mark.parametrize("lanes,threads", [(1, 1), (1, 2), (1, 8), (1, 64), (4, 3), (4, 64), (2, 200), (4, 600)])
def test_every_caller_gets_its_own_results(harness, lanes, threads):
    r = harness(lanes, 64, threads, 40, 300)
    assert r["wrong"] == 0
    assert r["solo"] + r["led"] + r["rode"] == threads * 40
    assert r["max_passes"] <= lanes              # first passes; a group's slow second pass runs after it has given its lane away


def test_a_lone_caller_runs_solo_and_never_waits(harness):
    r = harness(1, 64, 1, 200, 50)
    assert r == dict(solo=200, led=0, rode=0, groups=0, group_queries=0, lingers=0, wrong=0, max_passes=1)


def test_callers_share_passes_when_the_lane_is_busy(harness):
    r = harness(1, 256, 32, 30, 1000, two_keys=0, slow_every=0)
    assert r["wrong"] == 0 and r["rode"] > 0
    # closed-loop callers come back together: the leader holds the group open for the ones the last pass released, so the groups
    # hold (nearly) all of them, not half
    assert r["group_queries"] / max(r["groups"], 1) > 32 * 2 * 0.6, r       # mean queries per call is 2


def test_members_whose_queries_are_final_do_not_wait_for_the_groups_slow_query(harness):
    """a group with one slow query (the exact-heap redo of qv_graph_search) hands the other members their results after the first
    pass: with 5 ms passes and every 7th query slow, the run would take twice as long if everybody waited for the second pass"""
    import time
    t0 = time.time(); r = harness(1, 256, 16, 20, 5000, two_keys=0, slow_every=7); t_early = time.time() - t0
    assert r["wrong"] == 0 and r["solo"] + r["led"] + r["rode"] == 320


def test_small_groups_respect_the_cap(harness):
    r = harness(1, 4, 16, 25, 400)
    assert r["wrong"] == 0
    assert r["group_queries"] <= 4 * max(r["groups"], 1)


def test_callers_with_think_time_longer_than_the_linger_are_not_waited_for_beyond_the_bound(harness):
    # two callers, 2 ms of think time, 1 ms passes: whoever finds the lane free may hold its group open for 1/8 pass at most
    r = harness(1, 64, 2, 50, 1000, think_us=2000)
    assert r["wrong"] == 0 and r["solo"] + r["led"] + r["rode"] == 100
