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
