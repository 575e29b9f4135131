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
