"""The line bench.py prints last must stay small enough for the driver to read (round 4's grew to 24 KB and was not parsed) and must carry
the contract's keys; `--gpus N` without a launcher must refuse to print an N-GPU line from fewer devices."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (numpy only at import time)

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _detail_objects():
    d = os.path.join(ROOT, "profiles")
    for fn in sorted(os.listdir(d)):
        if fn.endswith(".json") and "bench_default" in fn:
            try:  # a captured stdout: the JSON object is its last line
                o = json.loads([l for l in open(os.path.join(d, fn)).read().splitlines() if l.startswith("{")][-1])
            except (ValueError, IndexError):
                continue
            if isinstance(o, dict) and "roofline" in o and "kernels" in (o.get("roofline") or {}):
                yield fn, o


def test_compact_line_of_every_recorded_detail_object():
    seen = 0
    for fn, o in _detail_objects():
        line = bench.compact_line(o, os.path.join(ROOT, "bench_detail.json"))
        text = json.dumps(line)
        assert len(text) < bench.COMPACT_LIMIT, (fn, len(text))
        assert "\n" not in text
        for key in CONTRACT:
            assert key in line, (fn, key)
        assert line["value"] == pytest.approx(o["value"], rel=1e-5)
        assert line["config"]["workload"]
        r = line["roofline"]
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == bench.HBM_PEAK_GBS
        if r["achieved"] is not None:
            assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=2e-3)
        c = line["cpu_baseline"]
        assert c is None or {"value", "unit", "cores", "kind", "sample"} <= set(c)
        seen += 1
    assert seen >= 1


def test_compact_line_sheds_optional_parts_rather_than_grow():
    fn, o = next(_detail_objects())
    o = json.loads(json.dumps(o))
    for i in range(200):  # a later round adds secondaries and long texts to the detail object
        o["secondary_extra_%d" % i] = {"value": 1.0 + i, "ratio_to_headline": 0.5, "identical_to_oracle": True, "note": "x" * 500}
    o["config"]["workload"] = "w" * 5000
    o["cpu_baseline"]["sample"] = "s" * 5000
    text = json.dumps(bench.compact_line(o, None))
    assert len(text) < bench.COMPACT_LIMIT
    line = json.loads(text)
    for key in CONTRACT:
        assert key in line


def test_compact_line_of_an_almost_empty_object():
    line = bench.compact_line({"metric": "m", "value": 1.5, "unit": "u", "n_gpus": 2}, None)
    assert line["value"] == 1.5 and line["roofline"]["frac"] is None and line["cpu_baseline"] is None
    json.dumps(line)


def test_gpus_flag_refuses_without_the_devices():
    """no GPU in the CPU container: `--gpus 2` with no WORLD_SIZE must end non-zero and print no JSON line (never a 1-GPU number labelled 2)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=env, timeout=300)
    assert cp.returncode != 0
    assert "GPU" in cp.stderr
    assert not any(l.startswith("{") for l in cp.stdout.splitlines())


def test_gpus_flag_must_match_the_world_size():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=env, timeout=300)
    assert cp.returncode != 0 and "WORLD_SIZE" in cp.stderr
