"""bench.py's contract line: one strict-JSON line the driver can keep whole (it records the last 8 KB of stdout), with the
headline, `roofline` and `cpu_baseline` on it whatever the extra configs measured; the full record goes to a sidecar."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

RECORDED = os.path.join(ROOT, "profiles", "r02_bench_n1.jsonl")     # full records of round 2's runs (26 KB each)


RECORDED_R03 = os.path.join(ROOT, "profiles", "r03_bench_full.json")   # round 3's full record (the sidecar of its 3.7 KB line)


def _records():
    return [json.loads(l) for l in open(RECORDED).read().strip().split("\n")] + [json.load(open(RECORDED_R03))]


@pytest.mark.parametrize("i", range(8))
def test_line_fits_and_parses(i):
    rec = _records()[i]
    text = bench.compact_line(rec, "gpurun_out/bench_full.json")
    assert "\n" not in text and len(text.encode()) < 4096
    line = json.loads(text, parse_constant=lambda c: pytest.fail("non-strict JSON constant " + c))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert key in line
    assert abs(line["value"] - rec["value"]) <= 1e-5 * rec["value"]
    assert line["config"]["workload"].startswith("flat cosine scan")
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4 and "traffic" in rf
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["sample"]
    assert set(cb["others"]) == {"faithful_one_query_per_core", "optimised_scan_not_the_reference"}
    also = line["also"]
    assert also["flat_1Mx768"]["frac"] > 0 and "b256x1Mx768_mfma" in also
    assert any(k.startswith("hnsw_1M") for k in also)


def test_extras_are_shed_before_the_headline():
    rec = _records()[-1]
    rec["also"] = {("batched_256x10Mx768_extra_%d" % i): dict(rec["also"]["batched_256x1Mx768_mfma"]) for i in range(200)}
    rec["config"]["sharding"] = "x" * 5000
    text = bench.compact_line(rec, None)
    assert len(text) <= bench.MAX_LINE
    line = json.loads(text)
    assert line["roofline"]["frac"] > 0 and line["cpu_baseline"]["value"] > 0 and line["also"]["in_sidecar_only"] > 0


def test_nan_never_reaches_the_line():
    rec = _records()[-1]
    rec["also"]["flat_1Mx768_single_query"]["hbm_frac"] = float("nan")
    json.loads(bench.compact_line(rec, None), parse_constant=lambda c: pytest.fail(c))


def test_emit_writes_sidecar_and_one_line(tmp_path):
    side = tmp_path / "full.json"
    code = ("import json,sys; sys.path.insert(0, %r); import bench; "
            "bench.emit(json.loads(open(%r).read().strip().split('\\n')[-1]))" % (ROOT, RECORDED))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, QV_BENCH_SIDECAR=str(side)), check=True)
    assert p.stdout.count("\n") == 1 and len(p.stdout) < 4096
    assert json.loads(p.stdout)["full_record"]
    full = json.loads(side.read_text())
    assert "hnsw_1Mx768_maxlevel1_structured" in full["also"] and len(side.read_text()) > 20000


def test_eight_gpus_and_maximum_length_strings_still_emit_the_headline():
    """the budget can be exceeded by the fixed part alone (8 per-GPU roofline entries, a long runtime string, every CPU leg):
    the line sheds extras instead of raising after the whole benchmark has run"""
    rec = _records()[-1]
    rec["n_gpus"] = 8
    rec["roofline"]["per_gpu"] = [{"rank": g, "rows": 1250000, "scan_kernel_ms": 0.5512345, "hbm_frac": 0.87123} for g in range(8)]
    rec["config"].update({k: "y" * 400 for k in ("runtime", "sharding", "exchange", "arithmetic", "workload", "device")})
    rec["cpu_baseline"]["sample"] = "z" * 3000
    rec["cpu_baseline"]["others"] = {("leg_%d" % i): {"value": 1.0 / 3, "cores": 64} for i in range(120)}
    text = bench.compact_line(rec, "gpurun_out/" + "p" * 300)
    assert len(text) <= bench.MAX_LINE
    line = json.loads(text)
    assert line["value"] == pytest.approx(rec["value"], rel=1e-5) and line["n_gpus"] == 8 and line["roofline"]["frac"] > 0
    assert line["cpu_baseline"]["value"] > 0


def test_membership_check_catches_a_skipped_row():
    """bench.py's verified_against_oracle: a result that misses a better row fails; results beyond the checked rows are ignored"""
    import numpy as np
    from tests import _oracle as O
    rows = O.gen_rows(1, 0, 5000, 32)
    q = O.gen_rows(2, 0, 1, 32)[0]
    er, ed = O.exact_search(0, rows, q, 10)
    assert bench.membership_check(O, 0, q, rows[:3000], 0, er.astype(np.int64), ed)
    assert bench.membership_check(O, 0, q, rows[100:2000], 100, er.astype(np.int64), ed)
    er2, ed2 = O.exact_search(0, rows, q, 11)
    assert not bench.membership_check(O, 0, q, rows, 0, er2[1:].astype(np.int64), ed2[1:])          # the best row is missing
    bad = ed.copy(); bad[3] = np.nextafter(bad[3], np.float32(2))
    assert not bench.membership_check(O, 0, q, rows, 0, er.astype(np.int64), bad)                    # a distance one ulp off
