"""CPU: the parts of bench.py that do not need a GPU (argument contract, synthetic problem set, word budgets)."""
import importlib.util
import os
import sys

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
    assert a.algo == "irrt" and a.dim == 2 and a.iters == 50000 and a.trees == 4096      # BASELINE.json configs[1], one tree per wave slot
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "2"])
    a = b.parse()
    assert (a.gpus, a.steps, a.warmup) == (8, 3, 2)


def test_problem_set_is_disjoint_across_ranks_and_sized_like_the_survey(monkeypatch):
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--trees", "6"])
    a = b.parse()
    p0, p1 = b.make_problems(a, 0), b.make_problems(a, 1)
    assert [p["pid"] for p in p0] == list(range(6)) and [p["pid"] for p in p1] == list(range(6, 12))
    for p in p0:
        assert p["clearance"] == 3 and len(p["env_dict"]["circle_obstacles"]) == 30 and not p["env_dict"]["rectangle_obstacles"]
        assert tuple(p["env_dict"]["env_dims"]) == (224, 224) and p["search_radius"] > 0
    n_np, n_py = b.word_budgets(a)
    assert n_np >= 2 * a.iters and n_py >= 4 * a.iters                                   # >= one SampleFree / unit-disk draw per iteration
