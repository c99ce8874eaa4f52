"""CPU: the parts of bench.py that do not need a GPU (argument contract, launcher, synthetic problem set, word budgets,
CPU-baseline helper)."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_default_arguments_follow_the_driver_contract(monkeypatch):
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = b.parse()
    assert a.gpus == 1 and a.steps >= 1 and a.warmup >= 0
    assert a.algo == "irrt" and a.dim == 2 and a.iters == 50000 and a.trees >= 4096      # BASELINE.json configs[1], >= one tree per wave slot
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "2"])
    a = b.parse()
    assert (a.gpus, a.steps, a.warmup) == (8, 3, 2)
    assert a.world == "b30r16" and b.config_key(a) == "irrt_2d_b30r16_%dx50000" % a.trees      # SURVEY 8(d)'s primary world


def test_problem_set_is_disjoint_across_ranks_and_sized_like_the_survey(monkeypatch):
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--trees", "6", "--world", "b30"])
    a = b.parse()
    p0, p1 = b.make_problems(a, 0), b.make_problems(a, 1)
    assert [p["pid"] for p in p0] == list(range(6)) and [p["pid"] for p in p1] == list(range(6, 12))
    for p in p0:
        assert p["clearance"] == 3 and len(p["env_dict"]["circle_obstacles"]) == 30 and not p["env_dict"]["rectangle_obstacles"]
        assert tuple(p["env_dict"]["env_dims"]) == (224, 224) and p["search_radius"] > 0
        assert all(8 <= c[2] <= 12 for c in p["env_dict"]["circle_obstacles"])
    n_np, n_py = b.word_budgets(a)
    assert n_np >= 2 * a.iters and n_py >= 4 * a.iters                                   # >= one SampleFree / unit-disk draw per iteration


def test_primary_world_of_the_survey_has_large_circles_and_connected_start_goal(monkeypatch):
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--trees", "2"])
    a = b.parse()
    for p in b.make_problems(a, 0):
        assert len(p["env_dict"]["circle_obstacles"]) == 30
        assert all(16 <= c[2] <= 24 for c in p["env_dict"]["circle_obstacles"])


@pytest.mark.parametrize("n", [2, 8])
def test_gpus_flag_spawns_that_many_ranks_and_prints_one_line(n):
    """`python bench.py --gpus N` with no launcher around it must become N ranks (gloo here: no GPU) and print ONE
    JSON line with n_gpus = N; the work of all ranks is in it (N = 8: the node the driver's scaling run uses)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--dry-run", "--steps", "3", "--trees", "10",
                        "--iters", "100"], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["dry_run"] is True and d["work_all_ranks"] == n * 10 * 100 * 3


def test_launcher_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                       cwd=ROOT, env=env, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_cpu_baseline_process_runs_the_oracle_on_a_batch_problem():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_bench.py"), "--iters", "1500", "--pid", "3"],
                       capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["iters"] == 1500 and d["seconds"] > 0 and 500 < d["n"] <= 1501


def test_traffic_entry_is_used_only_for_the_exact_configuration(tmp_path, monkeypatch):
    b = _bench()
    tab = {"formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024", "entries": {"irrt_2d_b30r16_4096x50000": {"traffic_bytes": 5e13, "collected": "2026-09-28",
                                                                                                    "kernel": "slim::k_run_sample<2>"}}}
    f = tmp_path / "t.json"
    f.write_text(json.dumps(tab))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--trees", "4096", "--traffic-file", str(f)])   # the entry's configuration
    t, src = b.measured_traffic(b.parse())
    assert t == 5e13 and isinstance(src, str) and "2026-09-28" in src and "FETCH_SIZE" in src and "irrt_2d_b30r16_4096x50000" in src
    monkeypatch.setattr(sys, "argv", ["bench.py", "--trees", "2048", "--traffic-file", str(f)])
    assert b.measured_traffic(b.parse()) == (None, None)


def test_strong_scaling_line_has_a_traffic_key_of_its_own(monkeypatch):
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--scaling", "strong", "--problems", "1000"])
    a = b.parse()
    assert b.config_key(a) == "irrt_2d_b30r16_set1000x50000"      # (not the default line's key: 1000 problems are another launch)


def test_another_refresh_policy_has_a_traffic_key_of_its_own(monkeypatch):
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--algo", "nirrt", "--dim", "3", "--trees", "2048"])
    assert b.config_key(b.parse()) == "nirrt_3d_ref3d_2048x50000"
    monkeypatch.setattr(sys, "argv", ["bench.py", "--algo", "nirrt", "--dim", "3", "--trees", "2048", "--pc-update-cost-ratio", "1.0"])
    assert b.config_key(b.parse()) == "nirrt_3d_ref3d_2048x50000_ratio1"      # (demo_planning_3d.py:21)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--pc-update-cost-ratio", "1.0"])
    assert b.config_key(b.parse()) == "irrt_2d_b30r16_8192x50000"             # (an unguided line has no refresh policy)


def test_every_bench_line_has_its_traffic_entry_in_the_committed_profile(monkeypatch):
    """the default line and every secondary line look their HBM traffic up in the round's traffic table (bench.py --traffic-file,
    default profiles/r06_traffic.json) by configuration key: a line added to bench.SECONDARY without its FETCH / WRITE passes would
    silently report traffic = null"""
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    with open(b.parse().traffic_file) as fh:
        entries = json.load(fh)["entries"]
    lines = [("default", [])] + list(b.SECONDARY)
    for label, extra in lines:
        monkeypatch.setattr(sys, "argv", ["bench.py"] + list(extra))
        a = b.parse()
        key = b.config_key(a)
        assert key in entries, (label, key)
        t, src = b.measured_traffic(a)
        assert t == entries[key]["traffic_bytes"] and t > 1e12, (label, key)


def test_scale_script_dry_run_collects_and_checks_every_line(tmp_path):
    """scripts/scale_1248.sh --dry-run: the launcher / process-group / timing protocol of every N (gloo ranks on the CPU) for the weak
    and the strong line, then scripts/scale_check.py over what it collected; a doctored line (wrong rank count) fails the check"""
    env = dict(os.environ, SCALE_NS="1 2 8", SCALE_STEPS="1", SCALE_WARMUP="0")
    r = subprocess.run([os.path.join(ROOT, "scripts", "scale_1248.sh"), "--dry-run", "--trees", "10", "--iters", "100"], capture_output=True,
                       text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    for mode in ("weak", "strong"):
        for n in (1, 2, 8):
            assert "%-6s N=%d  dry run: %d ranks" % (mode, n, n) in r.stdout, r.stdout
    out = os.path.join(ROOT, "gpurun_out", "scale")
    d = json.loads(open(os.path.join(out, "weak_2.json")).read().strip().splitlines()[-1])
    d["n_gpus"] = 3
    bad = tmp_path / "scale"
    bad.mkdir()
    (bad / "weak_2.json").write_text(json.dumps(d))
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "scale_check.py"), str(bad)], capture_output=True, text=True)
    assert r2.returncode == 1 and "3 ranks answered" in r2.stdout
