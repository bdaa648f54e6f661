"""bench.py's stdout contract: ONE line, < 6 KB, strict JSON, carrying value / ms_per_step / roofline /
cpu_baseline -- the driver kept `parsed: null` for round 5's 22.9 KB line.  The formatter is run here on canned records (the
builder-run round-5 record committed under profiles/, a record padded with prose and non-finite numbers, a --gpus 8 record, the
--synthetic record): no GPU."""
import copy
import io
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIMIT = 6000


def _bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_line_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _strict(line):
    def fail(name):
        raise AssertionError("non-finite constant in the line: " + name)
    assert "\n" not in line and len(line.encode()) < LIMIT, len(line.encode())
    return json.loads(line, parse_constant=fail)


def _canned():
    return json.load(open(os.path.join(ROOT, "profiles", "r05", "bench_r05.json")))


def test_the_round_5_record_becomes_a_line_the_driver_can_read():
    B = _bench()
    assert B.COMPACT_LINE_MAX <= LIMIT
    full = _canned()
    assert len(json.dumps(full)) > 20000          # what the driver could not parse
    c = _strict(B.format_line(full))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in c, k
    assert c["value"] == full["value"] and c["ms_per_step"] == full["ms_per_step"] and c["schema"] == B.SCHEMA
    assert c["config"]["workload"].startswith("veach-ajar 1920x1080") and "model" not in c["config"]
    r = c["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us", "layout_bytes_per_launch",
              "frac_model_8d", "value_region_kernel", "value_region_frac", "s1_pg_sample_frac"):
        assert k in r, k
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and len(r["frac_basis"]) <= 120
    # recomputable: layout bytes per launch / the average launch / peak
    assert abs(r["layout_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9 / r["peak"] - r["frac"]) < 2e-3
    b = c["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in b, k
    assert b["kind"] == "port" and len(b["sample"]) <= 240
    assert all(isinstance(v, (int, float)) for v in c["kernels"].values())       # name -> ms per step, nothing else
    assert abs(sum(v for k, v in c["kernels"].items() if k != "k_wave_guide") - c["ms_per_step"]) < 0.05 * c["ms_per_step"]
    assert c["extra"]["source_hash"] == full["extra"]["source_hash"]
    # no prose anywhere but the few labelled strings
    def strings(x, path=""):
        if isinstance(x, dict):
            for k, v in x.items():
                yield from strings(v, path + "/" + k)
        elif isinstance(x, str):
            yield path, x
    assert max(len(s) for _, s in strings(c)) <= 300


def test_a_padded_record_with_non_finite_numbers_still_fits_and_parses():
    B = _bench()
    full = _canned()
    full["value_full_schedule"] = float("nan")
    full["roofline"]["traffic"] = float("inf")
    full["roofline"]["frac_counter_hi"] = float("-inf")
    full["config"]["workload"] = "w" * 5000
    full["cpu_baseline"]["sample"] = "s" * 5000
    full["extra"]["exchange"] = "x" * 5000
    full["roofline"]["note"] = "n" * 50000
    full["kernels"] = {"k_%03d" % i: {"ms_per_step": 0.5, "note": "z" * 999} for i in range(400)}   # would overflow: dropped, not truncated
    c = _strict(B.format_line(full))
    assert c["roofline"]["traffic"] is None and c["roofline"]["frac_counter_hi"] is None and c.get("value_full_schedule") is None
    assert c["value"] == full["value"] and "roofline" in c and "cpu_baseline" in c


def test_the_line_of_a_sharded_run_and_of_the_synthetic_workload():
    B = _bench()
    full = _canned()
    full.update({"n_gpus": 8, "ranks": 8})
    full["cpu_baseline"] = None                     # N > 1: no CPU leg
    full["config"].update({"pixels_per_rank_min": 259200, "pixels_per_rank_max": 259200, "steps_per_launch": 8, "passes_per_launch": 128})
    full["extra"].update({"rccl_ranks": 8, "exchange": "pg_allreduce (RCCL ncclAllReduce int64 issued by libpgsd.so)"})
    c = _strict(B.format_line(full))
    assert c["n_gpus"] == 8 and c["ranks"] == 8 and c["cpu_baseline"] is None
    assert c["extra"]["rccl_ranks"] == 8 and "RCCL" in c["extra"]["exchange"] and c["extra"]["exchange_bytes"] > 0
    assert c["config"]["pixels_per_rank_min"] == 259200 and c["config"]["steps_per_launch"] == 8
    synth = {"metric": "Msamples/s guided (SD-tree hot path only, synthetic pass)", "value": 1.0, "unit": "Msamples/s", "n_gpus": 1,
             "ranks": 1, "steps": 2, "warmup": 1, "ms_per_step": 3.0, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
             "dtype": "f32", "data": "synthetic", "config": {"workload": "C2-synthetic"},
             "roofline": {"bound": "hbm", "kernel": "k_guide_bounce", "achieved": 1.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.1, "traffic": None},
             "cpu_baseline": None, "kernels": {"k_guide_bounce": {"launches": 16, "avg_us": 100.0}}}
    c = _strict(B.format_line(synth))
    assert c["data"] == "synthetic" and c["kernels"]["k_guide_bounce"] == pytest.approx(0.8) and c["roofline"]["traffic"] is None


def test_emit_puts_the_full_record_in_a_file_and_one_line_on_stdout(tmp_path, monkeypatch, capsys):
    B = _bench()
    full = _canned()
    full["mse_vs_gt"] = float("nan")
    detail = str(tmp_path / "detail.json")
    B.emit(copy.deepcopy(full), detail)
    cap = capsys.readouterr()
    lines = cap.out.splitlines()
    assert len(lines) == 1 and cap.out.endswith("\n")
    _strict(lines[0])
    assert "full record" in cap.err and "{" not in cap.err            # stderr: a pointer to the file, written BEFORE the line
    d = json.load(open(detail), parse_constant=lambda n: pytest.fail(n))
    assert d["kernels_synthetic"] == full["kernels_synthetic"] and d["mse_vs_gt"] is None


def test_the_launcher_relays_exactly_that_one_line(tmp_path):
    """`python bench.py --gpus N` without a launcher: spawn_ranks relays rank 0's stdout; with the compact line as rank 0's only
    JSON the parent's stdout is that line and nothing else."""
    B = _bench()
    line = B.format_line(_canned())
    child = ("import os, sys\n"
             "if os.environ['RANK'] == '0':\n"
             "    sys.stderr.write('[bench] full record: somewhere\\n'); sys.stdout.write(sys.argv[1] + '\\n')\n")
    relay = io.StringIO()
    assert B.spawn_ranks([sys.executable, "-c", child, line], 2, relay=relay) == 0
    assert relay.getvalue() == line + "\n"
    _strict(relay.getvalue().rstrip("\n"))
