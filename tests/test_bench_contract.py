"""The bench line's contract (driver + judge read it): the last recorded line under profiles/ must carry every
field with the right type, and bench.py must still produce those keys (static check of its source)."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_recorded_bench_line_has_the_contract_fields():
    import glob
    newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench.json")))[-1]      # this round's recorded line
    d = json.load(open(newest))
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                 ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert k in d and isinstance(d[k], t), k
    assert "vs_baseline" in d and d["vs_baseline"] is None            # BASELINE.md holds no published number for this metric
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] > 0
    if os.path.basename(newest) >= "r03_bench.json":
        assert "traffic_source" in r and "overflow" in d and "operator_api" in d and "generator_order" in d
        assert "split-fp16" in d["dtype"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)
    assert abs(d["value"] - 1e6 / (d["ms_per_step"] * 1e-3) / 1e6) < 0.05 * d["value"]      # Mpoints/s of the 1 M-point cycle


def test_bench_source_still_emits_the_fields():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "bound", "achieved", "peak", "frac",
              "traffic", "traffic_source", "overflow", "operator_api", "generator_order", "cores", "kind", "sample"):
        assert re.search(r'"%s"\s*[:\]]' % k, src), k
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in src
