"""The sharded evaluation harness END TO END on one rank: `python -m nirrt_star_amd.eval_sharded --max_problems 16` - argument
handling, problem loading, the batched device loop, the summary JSON and the reference-format result pickle
(eval_planning_2d.py:99-136) - and the same problems through plan_batch directly must give the same records.
Run plainly there is no process group and nothing is gathered (one rank).  Run under `torchrun --nproc-per-node 1` the harness
builds an RCCL process group of one rank and the records DO travel through dist.all_reduce / dist.gather / gather_object on
`cuda` tensors - the only way to execute that branch on a one-GPU box (RCCL refuses two ranks on one device: "Duplicate GPU
detected"); same for bench.py's barrier / all_reduce timing protocol."""
import json
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("launcher", ["plain", "torchrun-1"])
def test_eval_sharded_main_end_to_end(tmp_path, launcher):
    out = tmp_path / "res.json"
    pk = tmp_path / "res.pickle"
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    tail = ["-m", "nirrt_star_amd.eval_sharded", "--problem", "random_2d", "-p", "irrt_star", "--max_problems", "16",
            "--iter_max", "3000", "--batch", "8", "--out", str(out), "--pickle_out", str(pk)]
    if launcher == "plain":
        cmd = [sys.executable] + tail
    else:      # an RCCL process group of one rank: init_process_group("nccl"), all_reduce, gather, gather_object, barrier on cuda
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", "29533"] + tail
    p = subprocess.run(cmd, cwd=str(tmp_path), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    summary = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert summary["problems"] == 16 and summary["world_size"] == 1 and summary["solved"] > 0
    d = json.load(open(out))
    recs = np.array(d["records"])
    assert recs.shape[0] == 16 and sorted(recs[:, 0].astype(int).tolist()) == list(range(16))
    # the reference's pickle: list of env configs, each with its per-iteration 'result' list (eval_planning_2d.py:125-136)
    with open(pk, "rb") as f:
        cfgs = pickle.load(f)
    assert len(cfgs) == 16 and all("result" in c and "env_dict" in c for c in cfgs)
    assert all(len(c["result"]) > 0 for c in cfgs)
    if launcher != "plain":
        return
    # the same 16 problems through the batch function in this process: identical records (seeded, deterministic)
    from types import SimpleNamespace as NS
    from nirrt_star_amd import eval_sharded as es, problems as P
    a = NS(problem="random_2d", planner="irrt_star", neural_net="none", iter_max=3000, iter_after_initial=3000, step_len=10, clearance=3,
           pc_n_points=2048, pc_over_sample_scale=5, pc_sample_rate=0.5, pc_update_cost_ratio=0.9, connect_max_trial_attempts=5,
           root_dir=".", segment=2000)
    cfgs2 = P.get_random_2d_env_configs()[:16]
    probs = [P.get_random_2d_problem_input(c) for c in cfgs2]
    r, _ = es.plan_batch(probs, list(range(16)), a, 0, None)
    r = np.array(r).reshape(-1, es.RECORD_LEN)
    order = np.argsort(recs[:, 0])
    assert np.array_equal(np.nan_to_num(recs[order], posinf=1e300), np.nan_to_num(r, posinf=1e300))


def test_bench_timing_protocol_under_a_one_rank_rccl_group(tmp_path):
    """bench.py launched by torchrun with ONE rank: process group "nccl", barrier + all_reduce(MAX / SUM) of the timing protocol
    on cuda tensors, one JSON line"""
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29534", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--trees", "96", "--iters", "3000", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline", "--no-ttfs", "--no-secondary"]
    p = subprocess.run(cmd, cwd=str(tmp_path), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0 and d["process_group"] == "nccl"
    assert d["config"]["trees_stopped_early"] == 0


def test_eval_sharded_resumes_from_an_existing_result_file(tmp_path):
    """The reference's resume (eval_planning_2d.py:99-110): an existing result pickle holds the first K problems; they are kept
    as they are and only problems K.. are planned.  A run over 12 problems, its pickle cut back to 5 entries, the same command
    again: 7 problems planned, the final file equal to the uninterrupted one (seeded per problem: deterministic)."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    pk = tmp_path / "results" / "evaluation" / "2d" / "random_2d-irrt_star-none-12.pickle"
    cmd = [sys.executable, "-m", "nirrt_star_amd.eval_sharded", "--problem", "random_2d", "-p", "irrt_star", "--num_problems", "12",
           "--iter_max", "3000", "--iter_after_initial", "500", "--out", str(tmp_path / "res.json")]       # --pickle_out auto: the reference's file name

    def run():
        p = subprocess.run(cmd, cwd=str(tmp_path), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        with open(pk, "rb") as f:
            return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1]), pickle.load(f)

    s1, full = run()
    assert s1["problems"] == 12 and s1["resumed_from"] == 0 and s1["planned"] == 12 and len(full) == 12
    with open(pk, "wb") as f:
        pickle.dump(full[:5], f)
    s2, again = run()
    assert s2["resumed_from"] == 5 and s2["planned"] == 7 and s2["problems"] == 12 and s2["solved"] == s1["solved"]
    assert len(again) == 12
    for a, b in zip(full, again):
        assert a["img_idx"] == b["img_idx"] and a["start_goal_idx"] == b["start_goal_idx"]
        assert np.array_equal(np.asarray(a["result"]), np.asarray(b["result"]))
    assert s2["mean_cost_at"] == s1["mean_cost_at"] and s2["median_first_solution_iter"] == s1["median_first_solution_iter"]
    # a complete file: nothing left to plan
    s3, third = run()
    assert s3["resumed_from"] == 12 and s3["planned"] == 0 and len(third) == 12
    # the same file name, another experiment (the reference's name holds problem / planner / network / count only): the fingerprint
    # beside the pickle does not match -> nothing is reused, every problem is planned again
    cmd[cmd.index("--iter_after_initial") + 1] = "400"
    s4, fourth = run()
    assert s4["resumed_from"] == 0 and s4["planned"] == 12 and len(fourth) == 12
    assert all(len(a["result"]) == len(b["result"]) - 100 for a, b in zip(fourth, full) if np.isfinite(np.asarray(b["result"])).any())
    # ... and a result file without a fingerprint (written by something else) is not trusted either
    os.remove(str(pk) + ".meta.json")
    s5, _ = run()
    assert s5["resumed_from"] == 0 and s5["planned"] == 12


def test_free_segment_predictor_agrees_with_the_device_collision_test():
    """eval_sharded.straight_segment_free (host, sampled; used only to deal the slow problems evenly across ranks) against the
    device's exact segment test on the first 200 problems of the 2D set and 100 of the 3D set: it may err on grazing segments,
    nothing else"""
    from nirrt_star_amd import _hip, eval_sharded as es, problems as P
    for dim, cfgs, get, clr in ((2, P.get_random_2d_env_configs()[:200], P.get_random_2d_problem_input, 3),
                                (3, P.get_random_3d_env_configs()[:100], P.get_random_3d_problem_input, 2)):
        agree = 0
        for i, c in enumerate(cfgs):
            if dim == 3:
                np.random.seed(i)
            pr = get(c)
            t = _hip.HipTree(dim, 8, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], clr, pr["env"])
            free_dev = not t.is_collision(pr["x_start"], pr["x_goal"])
            agree += int(free_dev == es.straight_segment_free(c["env_dict"], clr))
            t.close()
        assert agree >= 0.97 * len(cfgs), (dim, agree, len(cfgs))
