"""N>1 path on CPU: world_size-2 gloo processes shard a problem list round-robin, build result records
and gather them on rank 0 (the planning itself needs a GPU and is covered by the -m gpu tests)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_problems, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from nirrt_star_amd import eval_sharded as es
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = es.shard_indices(n_problems, rank, world)
    recs = []
    for pid in mine:
        trace = np.full(400, np.inf)
        first = 10 + pid
        trace[first:] = 300.0 - 0.01 * np.arange(400 - first) - pid      # deterministic fake cost curve
        recs.append(es.make_record(pid, trace, 100 + pid))
    out = es.gather_records(np.array(recs), world, rank, device="cpu")
    # variable-length result lists (the reference's pickle wire format) travel as objects
    res = es.gather_results([(pid, [float(pid)] * (1 + pid % 3)) for pid in mine], world, rank)
    if rank == 0:
        q.put((out, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_problems", [7, 8])
def test_round_robin_shard_and_gather(n_problems):
    from nirrt_star_amd import eval_sharded as es
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + n_problems
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_problems, q)) for r in range(world)]
    for p in procs:
        p.start()
    out, res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out.shape == (n_problems, es.RECORD_LEN)
    assert np.array_equal(out[:, 0], np.arange(n_problems))                 # every problem exactly once, sorted
    assert np.array_equal(out[:, 1], 11 + np.arange(n_problems))            # first-solution iteration (1-based)
    assert np.array_equal(out[:, 2], 100 + np.arange(n_problems))
    assert np.allclose(out[:, 4], 300.0 - np.arange(n_problems))            # cost at +0
    assert np.isfinite(out[:, 5]).all() and np.isinf(out[:, 6]).all()       # +250 inside the 400-long trace, +500 not
    assert res == [(pid, [float(pid)] * (1 + pid % 3)) for pid in range(n_problems)]


def test_reference_result_pickle_wire_format(tmp_path):
    """eval_planning_2d.py:100-136: list of env-config copies with the planner's list under 'result'"""
    import pickle
    from nirrt_star_amd import eval_sharded as es
    cfgs = [{"img_idx": i, "env_dict": {"env_dims": [224, 224]}} for i in range(3)]
    traces = [np.array([np.inf, 90.0, 80.0]), np.array([np.inf, np.inf]), np.array([70.0, 60.0, 50.0])]
    lists = es.result_lists("block", traces, [85.0, 10.0, 65.0])
    assert lists == [[np.inf, 90.0, 80.0], [np.inf, np.inf], [70.0, 60.0]]      # cut right after the first sub-threshold entry
    assert es.result_lists("random_2d", traces)[2] == [70.0, 60.0, 50.0]
    path = str(tmp_path / "results" / "evaluation" / "2d" / "block-irrt_star-none-3.pickle")
    es.write_reference_pickle(path, cfgs, list(zip(range(3), lists)))
    with open(path, "rb") as f:
        got = pickle.load(f)
    assert [g["img_idx"] for g in got] == [0, 1, 2] and got[2]["result"] == [70.0, 60.0]
    assert "result" not in cfgs[0]                                          # the configs themselves are not modified


def test_shard_indices_partition():
    from nirrt_star_amd import eval_sharded as es
    for n in (0, 1, 5, 1000):
        for w in (1, 2, 8):
            parts = [es.shard_indices(n, r, w) for r in range(w)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_heavy_problems_are_dealt_across_the_ranks_first():
    """with the free-segment predictor every rank gets the same number of heavy problems (+- 1) and the same
    number in total (+- 1); a resumed run shards only the problems the loaded file does not hold"""
    from nirrt_star_amd import eval_sharded as es
    rng = np.random.default_rng(0)
    for n, w, first in ((1000, 8, 0), (1000, 8, 123), (37, 4, 0), (5, 8, 0), (0, 2, 0)):
        heavy = list(rng.random(n) < 0.12)
        parts = [es.shard_indices(n, r, w, heavy, first=first) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(first, n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
        hv = [sum(heavy[i] for i in p) for p in parts]
        assert max(hv) - min(hv) <= 1
        assert all(p == sorted(p) for p in parts)


def test_free_segment_predictor_on_known_worlds():
    from nirrt_star_amd import eval_sharded as es
    ed = {"env_dims": (224, 224), "circle_obstacles": [[100, 100, 10]], "rectangle_obstacles": [[150, 20, 20, 30]],
          "start": [[20, 100]], "goal": [[200, 100]]}
    assert not es.straight_segment_free(ed, 3)                                     # straight through the circle
    assert es.straight_segment_free(ed, 3, x_start=(20, 120), x_goal=(200, 120))   # passes 20 above its centre (r + clearance = 13)
    assert not es.straight_segment_free(ed, 3, x_start=(20, 112), x_goal=(200, 112))
    assert not es.straight_segment_free(ed, 3, x_start=(140, 35), x_goal=(200, 35))   # through the rectangle
    assert es.straight_segment_free(ed, 3, x_start=(140, 60), x_goal=(200, 60))
    ed3 = {"env_dims": [50, 50, 50], "ball_obstacles": [[25, 25, 25, 5]], "box_obstacles": [[5, 5, 5, 10, 10, 10]],
           "start": [[2, 25, 25]], "goal": [[48, 25, 25]]}
    assert not es.straight_segment_free(ed3, 2)
    assert es.straight_segment_free(ed3, 2, x_start=(2, 40, 40), x_goal=(48, 40, 40))
    assert not es.straight_segment_free(ed3, 2, x_start=(2, 10, 10), x_goal=(48, 10, 10))


def _worker8(rank, world, port, n_problems, first, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from nirrt_star_amd import eval_sharded as es
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    heavy = [(i * 7) % 11 == 0 for i in range(n_problems)]
    mine = es.shard_indices(n_problems, rank, world, heavy, first=first)
    recs = [es.make_record(pid, np.full(30, 100.0 - pid), 50 + pid) for pid in mine]
    out = es.gather_records(np.array(recs).reshape(-1, es.RECORD_LEN), world, rank, device="cpu")
    res = es.gather_results([(pid, [float(pid)]) for pid in mine], world, rank)
    secs = es.gather_rank_seconds(0.5 + rank, len(mine), world, rank)
    if rank == 0:
        q.put((out, res, secs))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_problems,first", [(21, 0), (21, 18), (1000, 997)])
def test_world_8_gather_with_uneven_and_empty_shards(n_problems, first):
    """eight gloo ranks (the node the path is meant for): balanced shards, ranks with one problem fewer than others and ranks
    with NO problem at all (a resumed run with 3 problems left) all take part in the one gather"""
    from nirrt_star_amd import eval_sharded as es
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31000 + (os.getpid() % 2000) + n_problems % 97 + first % 13
    procs = [ctx.Process(target=_worker8, args=(r, world, port, n_problems, first, q)) for r in range(world)]
    for p in procs:
        p.start()
    out, res, secs = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ids = np.arange(first, n_problems)
    assert out.shape == (len(ids), es.RECORD_LEN) and np.array_equal(out[:, 0], ids)
    assert np.allclose(out[:, 4], 100.0 - ids)
    assert res == [(int(pid), [float(pid)]) for pid in ids]
    assert [round(s[0] - 0.5) for s in secs] == list(range(8)) and sum(s[1] for s in secs) == len(ids)
