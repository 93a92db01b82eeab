"""The bench line the driver parses must stay ONE compact JSON object (round 5's 21 KB line did not fit the driver's capture and the round's
headline went unmeasured). `bench.headline()` is fed the fullest record a default run has ever produced (profiles/r5e_bench.json: every side block present, eight
per-rank objects) plus the keys added since, and an adversarial one whose strings are far too long."""
import copy
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def canned():
    out = json.load(open(os.path.join(REPO, "profiles", "r5e_bench.json")))
    out["roofline"]["step"] = {"what": "all SURVEY 8d bytes of a step / ms_per_step / 8 TB/s",
                               "yeast-like-2.5M": {"frac": 0.1, "bytes": 4.0e9, "ms": 5.0},
                               "config3-full-200M": {"frac": 0.1, "bytes": 5.349e11, "ms": 666.0},
                               "genome3g-300M": {"frac": 0.076, "bytes": 3.789e11, "ms": 620.0}}
    out["scaling_model_8_ranks"]["busiest_owner_share"] = 0.2344
    return out


def test_headline_is_one_compact_json_object():
    out = canned()
    assert len(json.dumps(out)) > 15000                      # the record itself is what no longer fitted
    line = bench.headline(out)
    assert "\n" not in line and len(line) < bench.HEADLINE_MAX_BYTES
    h = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in h, k
    assert h["config"]["workload"] == out["config"]["workload"] and "model" not in h["config"]
    r = h["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "at_scale", "genome3g", "step"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["frac"] - out["roofline"]["frac"]) < 1e-4 and abs(h["value"] - out["value"]) / out["value"] < 1e-4
    assert r["at_scale"]["frac"] and r["genome3g"]["frac"] and set(r["step"]) >= {"yeast-like-2.5M", "config3-full-200M", "genome3g-300M"}
    c = h["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] == 1 and c["value"] > 0 and c["unit"] and c["sample"]
    assert h["parity_vs_cpu_on_bench_input"] == "identical" and h["build_id"]
    s = h["summary"]
    assert s["scaling_model_8_ranks"]["modelled_speedup_vs_1_gpu"] == out["scaling_model_8_ranks"]["modelled_speedup_vs_1_gpu"]
    assert s["full_config3"]["end_to_end"]["wall_seconds"] and s["genome3g"]["config5"]["purity"]
    assert "per_rank" not in line and "traffic_note" not in line


def test_headline_sheds_weight_rather_than_overflow():
    out = canned()
    out["cpu_baseline"]["sample"] = "x" * 5000
    out["device_ms_per_step"] = {"timer_%03d" % i: 1.2345678 for i in range(300)}
    out["parity_vs_cpu_on_bench_input"] = "DIFFERENT: " + "y" * 4000
    line = bench.headline(out)
    assert len(line) < bench.HEADLINE_MAX_BYTES
    h = json.loads(line)
    assert h["roofline"]["frac"] and h["cpu_baseline"]["value"] and h["value"]


def test_headline_of_a_run_without_side_blocks():
    out = canned()
    for k in ("secondary", "full_config3", "genome3g", "scaling_model_8_ranks", "cpu_baseline_omp", "end_to_end"):
        out.pop(k)
    out["roofline"].pop("at_scale"); out["roofline"].pop("genome3g")
    out["full_config3"] = {"skipped": "fewer than 32 host threads"}
    h = json.loads(bench.headline(copy.deepcopy(out)))
    assert h["summary"]["full_config3"]["skipped"] and h["roofline"]["frac"]
