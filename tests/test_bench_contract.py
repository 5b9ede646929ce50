"""CPU checks of bench.py's reporting helpers and of the committed bench line (profiles/r06_bench.json): the JSON contract of the driver (metric / value / unit / n_gpus / steps /
warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload) plus the `roofline` and `cpu_baseline` objects, and the arithmetic behind the
roofline numbers (algorithmic bytes per solve; executed work from the committed counter pass)."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_algorithmic_bytes_follow_survey_8d():
    b = _bench()
    assert b.algorithmic_bytes_per_solve(20) == 1656 and b.algorithmic_bytes_per_solve(50) == 4056          # configs 1 and 2 / 4
    assert b.algorithmic_bytes_per_solve(80, 8, 160) == 7736 and b.algorithmic_bytes_per_solve(120, 4) == 4828      # config 3 (16 quads), config 5 in fp32


def test_executed_work_from_the_committed_counter_pass():
    b = _bench()
    flops = 1024 * 43.0 * 914.0 * 49
    ex = b.executed_work("carlike_n50_B1024_c4", 4.42, flops)
    assert ex is not None and 0.10 < ex["issue_slot_frac"] < 0.25 and ex["issue_slot_frac"] == ex["valu_issue_slot_frac"]
    assert 8.0 < ex["lane_slots_per_flop"] < 20.0 and ex["source"].startswith("profiles/r06_")
    assert 0.3 < ex["valu_busy_frac_of_wave_cycles"] < 0.7 and 0.1 < ex["wait_any_frac_of_wave_cycles"] < 0.6
    assert b.executed_work("no_such_workload", 1.0) is None
    t = b.measured_traffic("carlike_n50_B1024_c4")
    assert t is not None and 4.0e6 < t < 1.0e7          # 2 x FETCH_SIZE + WRITE_SIZE of the headline launch: ~1.5 x the 4.15 MB of algorithmic bytes


def test_committed_bench_line_keeps_the_contract():
    line = json.loads(open(os.path.join(ROOT, "profiles", "r06_bench.json")).read().strip().splitlines()[-1])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert line["metric"] in base["metric"] and line["unit"] == "solves/s" and line["n_gpus"] == 1 and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert line["vs_baseline"] is None and line["dtype"] == "f64" and line["data"] == "synthetic" and line["steps"] == 20 and line["warmup"] == 3
    assert "configs[1]" in line["config"]["workload"] and "model" not in line["config"]
    assert abs(line["value"] - 1024 * line["solver"]["converged_frac"] * 1e3 / line["ms_per_step"]) < 1e-6 * line["value"]
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["traffic"] > r["algorithmic_bytes_per_launch"]
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-9 and r["algorithmic_bytes_per_launch"] == 4056 * 1024
    assert r["waves_per_simd"] == 1.0 and 0.1 < r["issue_slot_frac"] < 0.25 and 8.0 < r["lane_slots_per_flop"] < 20.0
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "solves/s" and c["cores"] >= 1 and c["value"] > 0 and "instances" in c["sample"]
    assert line["solver"]["answers_equal_to_the_reference_path_alone"] == 1.0
    w2 = line["legs"]["reference_grid_n20_B32768"]
    assert w2["same_answers"] and w2["two_waves_per_simd"]["workgroups_per_cu"] == 8 and w2["two_waves_per_simd"]["value"] > 1.3 * w2["one_wave_per_simd"]["value"]
