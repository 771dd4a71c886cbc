"""bench.py's contract line (CPU): the full record of a run is folded into ONE compact JSON line (< 4 KB) + an extras file.  Round 4's driver record
had `parsed: null` because the line had grown to 20 KB; profiles/r04ar_bench.json is that very line and serves as the input here."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline", "cpu_baseline")


def _bench():
    sys.path.insert(0, ROOT)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        import bench
    finally:
        sys.argv = argv
    return bench


def test_contract_line_of_a_full_run_is_compact_and_complete(tmp_path, monkeypatch):
    bench = _bench()
    full = json.loads(open(os.path.join(ROOT, "profiles", "r04ar_bench.json")).read().strip().splitlines()[-1])
    assert len(json.dumps(full)) > 16000
    monkeypatch.setenv("MIND_BENCH_EXTRAS", str(tmp_path / "x.json"))
    text = bench.contract_line(full, None)
    assert "\n" not in text and len(text) < bench.LINE_LIMIT == 4096
    d = json.loads(text)
    for k in CONTRACT:
        assert k in d, k
    assert set(d["roofline"]) >= {"kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms"}
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample", "cpu", "value_1_thread"}
    assert abs(d["value"] - full["value"]) < 1e-4 * full["value"] and d["steps"] == full["steps"] and d["warmup"] == full["warmup"]
    assert abs(d["ms_per_step"] - full["ms_per_step"]) < 1e-4 * full["ms_per_step"]
    full_b = dict(full, bf16x3={"value": 1500.0, "ms_per_step": 3.3, "hbm_frac": 0.1, "note": "x" * 300},
                  tree_f32=dict(full["tree"], arith="f32"))
    d2 = json.loads(bench.contract_line(full_b, None))
    assert d2["bf16x3"] == {"value": 1500.0, "ms_per_step": 3.3, "hbm_frac": 0.1} and d2["tree_f32"]["ms_per_plan"] > 0 and "mfma_frac" in d2["tree_f32"]["k_pair"]
    json.dump(full, open(str(tmp_path / "x.json"), "w"), indent=1)
    assert d["k_ilqr"]["kernel_ms_per_launch"] > 0 and d["tree"]["ms_per_plan"] > 0 and d["tree"]["k_pair"]["hbm_frac"] > 0
    assert d["extras_file"] == str(tmp_path / "x.json") and json.load(open(d["extras_file"])) == full


def test_contract_line_sheds_optional_blocks_before_it_exceeds_the_limit(tmp_path, monkeypatch):
    bench = _bench()
    full = json.loads(open(os.path.join(ROOT, "profiles", "r04ar_bench.json")).read().strip().splitlines()[-1])
    full["config"]["workload"] = full["config"]["workload"] + " x" * 600          # an over-long description must not cost the contract keys
    monkeypatch.setenv("MIND_BENCH_EXTRAS", str(tmp_path / "x.json"))
    text = bench.contract_line(full, None)
    d = json.loads(text)
    assert len(text) <= bench.LINE_LIMIT and all(k in d for k in CONTRACT) and "stress_deep" not in d
