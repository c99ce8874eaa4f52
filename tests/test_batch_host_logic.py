"""Host logic of the batched planners without a GPU: who owns a problem's generators when (batch.ProblemStreams / hand_over /
fetch_np) and how run_scheduled re-orders a batch between two launches.  The library calls are replaced by a fake device that
keeps the generator states it is given and "draws" by advancing real numpy / CPython generators - the host side must end up
exactly where a process that drew everything itself ends up (the reference draws from np.random / random in ONE process:
rrt_base_2d.py:46-52, irrt_star_2d.py:121-151, point_cloud_mask_utils.py:35-73)."""
import random

import numpy as np
import pytest

from nirrt_star_amd import _hip, batch


class FakeTree:
    def __init__(self, name):
        self.name = name
        self.np_state = None
        self.py_state = None


class FakeDevice:
    """stands in for nirrt_set_generators / nirrt_get_generators / nirrt_run; counts the calls"""

    def __init__(self):
        self.set_calls = []
        self.get_calls = []
        self.launches = []

    def set_generators(self, trees, np_states=None, py_states=None):
        self.set_calls.append((len(trees), np_states is not None, py_states is not None))
        for k, t in enumerate(trees):
            if np_states is not None:
                t.np_state = (np.array(np_states[k][0], dtype=np.uint32), int(np_states[k][1]))
            if py_states is not None:
                t.py_state = (np.array(py_states[k][0], dtype=np.uint32), int(py_states[k][1]))

    def get_generators(self, trees, want_np=True, want_py=True):
        self.get_calls.append((len(trees), want_np, want_py))
        nk = np.stack([t.np_state[0] for t in trees]) if want_np else None
        npos = np.array([t.np_state[1] for t in trees]) if want_np else None
        pk = np.stack([t.py_state[0] for t in trees]) if want_py else None
        ppos = np.array([t.py_state[1] for t in trees]) if want_py else None
        return nk, npos, pk, ppos

    @staticmethod
    def draw_on_device(tree, n_np, n_py):
        """what the kernel does to a tree's generators: n doubles from each"""
        rs = np.random.RandomState(0)
        _hip.set_np_state(tree.np_state[0], tree.np_state[1], rs)
        rs.random_sample(n_np)
        tree.np_state = _hip.np_state(rs)
        pg = random.Random(0)
        _hip.set_py_state(tree.py_state[0], tree.py_state[1], pg)
        for _ in range(n_py):
            pg.random()
        tree.py_state = _hip.py_state(pg)


@pytest.fixture
def dev(monkeypatch):
    d = FakeDevice()
    monkeypatch.setattr(_hip, "set_generators", d.set_generators)
    monkeypatch.setattr(_hip, "get_generators", d.get_generators)
    return d


def test_hand_over_uploads_every_fresh_stream_in_one_call(dev):
    trees = [FakeTree(i) for i in range(5)]
    streams = [batch.ProblemStreams(100 + i) for i in range(5)]
    batch.hand_over(trees, streams)
    assert dev.set_calls == [(5, True, True)]
    for i, (t, s) in enumerate(zip(trees, streams)):
        assert s.bound_to(t)
        k, p = _hip.np_state(np.random.RandomState(100 + i))
        assert t.np_state[1] == p and np.array_equal(t.np_state[0], k)
        k, p = _hip.py_state(random.Random(100 + i))
        assert t.py_state[1] == p and np.array_equal(t.py_state[0], k)
    # nothing changed hands since: the next launch uploads nothing
    batch.hand_over(trees, streams, only_touched=True)
    assert len(dev.set_calls) == 1


def test_host_draws_continue_where_the_device_stopped_and_go_back(dev):
    """device draws, host draws (cloud candidates), device draws again = one process drawing everything in that order"""
    tree, s = FakeTree(0), batch.ProblemStreams(7)
    twin_np, twin_py = np.random.RandomState(7), random.Random(7)
    batch.hand_over([tree], [s])
    dev.draw_on_device(tree, 11, 5)
    twin_np.random_sample(11)
    [twin_py.random() for _ in range(5)]
    s.device_drew()
    # the host object is behind until someone asks for it: ONE fetch, numpy only
    a = s.rs.uniform(0, 1, 4)
    assert dev.get_calls == [(1, True, False)]
    assert np.array_equal(a, twin_np.uniform(0, 1, 4))
    assert s.touched()
    # a second look without device draws in between fetches nothing
    b = s.rs.random_sample(2)
    assert len(dev.get_calls) == 1 and np.array_equal(b, twin_np.random_sample(2))
    # before the next launch the touched state goes back to the tree - numpy only, python was never handed out
    batch.hand_over([tree], [s], only_touched=True)
    assert dev.set_calls[-1] == (1, True, False) and not s.touched()
    dev.draw_on_device(tree, 3, 2)
    twin_np.random_sample(3)
    [twin_py.random() for _ in range(2)]
    s.device_drew()
    assert s.rs.random_sample() == twin_np.random_sample()
    assert s.py.random() == twin_py.random()


def test_a_stream_that_moves_to_another_tree_continues_where_the_first_tree_stopped(dev):
    """bind() to a different tree first fetches what the previous owner drew - the problem must not replay outputs"""
    a, b, s = FakeTree("a"), FakeTree("b"), batch.ProblemStreams(21)
    twin_np, twin_py = np.random.RandomState(21), random.Random(21)
    batch.hand_over([a], [s])
    dev.draw_on_device(a, 9, 4)
    twin_np.random_sample(9)
    [twin_py.random() for _ in range(4)]
    s.device_drew()
    batch.hand_over([b], [s])                       # tree b takes the problem over
    assert s.bound_to(b) and not s.bound_to(a)
    k, p = _hip.np_state(twin_np)
    assert b.np_state[1] == p and np.array_equal(b.np_state[0], k)
    k, p = _hip.py_state(twin_py)
    assert b.py_state[1] == p and np.array_equal(b.py_state[0], k)


def test_closing_the_tree_brings_the_generators_home(dev):
    """run_batch's promise `streams[i].rs afterwards`: a tree that is closed hands the states back first (HipTree.close runs
    the release hooks); afterwards the stream is the host's again and asks no dead handle for anything"""
    t, s = FakeTree(0), batch.ProblemStreams(5)
    twin = np.random.RandomState(5)
    batch.hand_over([t], [s])
    dev.draw_on_device(t, 6, 0)
    twin.random_sample(6)
    s.device_drew(py_too=False)
    assert [h() for h in t._release_hooks] == [s.release]      # (weak references: the tree keeps no stream alive)
    _hip.HipTree.close(t_close := type("T", (), {"h": 1, "L": type("L", (), {"nirrt_destroy": staticmethod(lambda h: None)})(),
                                              "_release_hooks": t._release_hooks})())
    assert t_close.h is None and not s.bound_to(t)
    n_get = len(dev.get_calls)
    assert s.rs.random_sample() == twin.random_sample()
    assert len(dev.get_calls) == n_get             # nothing fetched after the release: the host object is current


def test_device_drew_with_one_stream_only_leaves_the_other_current(dev):
    tree, s = FakeTree(0), batch.ProblemStreams(3)
    batch.hand_over([tree], [s])
    dev.draw_on_device(tree, 8, 0)
    s.device_drew(py_too=False)       # a 3D tree never touches the python generator
    s.py.random()
    assert dev.get_calls == []        # ... so looking at it costs no round trip
    s.rs.random_sample()
    assert dev.get_calls == [(1, True, False)]


def test_fetch_np_batches_the_streams_that_are_behind(dev):
    trees = [FakeTree(i) for i in range(4)]
    streams = [batch.ProblemStreams(20 + i) for i in range(4)]
    batch.hand_over(trees, streams)
    for i in (0, 2, 3):
        dev.draw_on_device(trees[i], 5 + i, 1)
        streams[i].device_drew()
    batch.fetch_np(trees, streams, [0, 1, 2])
    assert dev.get_calls == [(2, True, False)]      # 1 is current, 3 was not asked for
    for i in (0, 2):
        twin = np.random.RandomState(20 + i)
        twin.random_sample(5 + i)
        assert streams[i]._rs.random_sample() == twin.random_sample()
    assert streams[3]._behind[0]


def test_unbound_stream_is_a_plain_pair_of_generators(dev):
    s = batch.ProblemStreams(5)
    assert s.rs.random_sample() == np.random.RandomState(5).random_sample()
    assert s.py.random() == random.Random(5).random()
    assert not s.touched() and dev.get_calls == [] and dev.set_calls == []
    a, b = batch.ProblemStreams(9), batch.ProblemStreams(9)
    assert int(a.fps_start(2048)) == int(b.fps_start(2048))


# ------------------------------------------------------------------------------------------------
# run_scheduled
# ------------------------------------------------------------------------------------------------
def _fake_run(log, secs_of, visits_of, stop_at=None):
    def run_sampling(trees, iters, flags=0, lanes_hint=None, **kw):
        names = [t.name for t in trees]
        log.append({"names": names, "iters": iters, "hint": None if lanes_hint is None else list(lanes_hint)})
        n = len(trees)
        stats = np.zeros((n, _hip.N_STATS), dtype=np.int64)
        status = np.zeros(n, dtype=np.int32)
        done = np.full(n, iters, dtype=np.int64)
        for j, b in enumerate(names):
            stats[j, _hip.ST_ITERS] = iters
            stats[j, 0] = int(visits_of[b] * iters)
            stats[j, _hip.ST_BUSY] = int(secs_of[b] * 1e8)
            if stop_at is not None and b in stop_at and len(log) - 1 == stop_at[b]:
                status[j] = _hip.E_CAPACITY
                done[j] = iters // 2
        return {"kernel_ms": 1000.0 * max(secs_of[b] for b in names), "stats": stats, "alg_elems": done * 10, "iters_done": done,
                "status": status, "np_used": done * 2, "py_used": done * 4}
    return run_sampling


def test_run_scheduled_dispatches_longest_first_and_widens_heavy_trees(monkeypatch):
    trees = [FakeTree(i) for i in range(6)]
    secs = {0: 1.0, 1: 5.0, 2: 2.0, 3: 9.0, 4: 0.5, 5: 3.0}
    visits = {0: 900, 1: 7000, 2: 2500, 3: 14000, 4: 100, 5: 1999}
    log = []
    monkeypatch.setattr(_hip, "run_sampling", _fake_run(log, secs, visits))
    tot = batch.run_scheduled(trees, [100, 100, 50], 0, wide_visits=6000, narrow_visits=2000)
    assert [l["iters"] for l in log] == [100, 100, 50]
    assert log[0]["names"] == [0, 1, 2, 3, 4, 5] and log[0]["hint"] is None
    # second and third launch: by the device time of the launch before, heavy visits on wider workgroups
    for l in log[1:]:
        assert l["names"] == [3, 1, 5, 2, 0, 4]
        assert l["hint"] == [256, 256, 0, 128, 0, 0]
    assert tot["wide"] == 2 and tot["narrow"] == 1
    # sums land on the trees they belong to, whatever the dispatch order was
    assert np.array_equal(tot["iters_done"], np.full(6, 250))
    assert np.allclose(tot["seconds"], [3 * secs[b] for b in range(6)])
    assert tot["kernel_ms"] == pytest.approx(3 * 9000.0)
    assert tot["words"] == 6 * 250 * 6
    assert np.array_equal(tot["stats"][:, _hip.ST_ITERS], np.full(6, 250))


def test_run_scheduled_without_reorder_keeps_the_order_but_still_widens(monkeypatch):
    trees = [FakeTree(i) for i in range(3)]
    log = []
    monkeypatch.setattr(_hip, "run_sampling", _fake_run(log, {0: 1.0, 1: 3.0, 2: 2.0}, {0: 10, 1: 10, 2: 9000}))
    batch.run_scheduled(trees, [10, 10], 0, wide_visits=4000, reorder=False)
    assert log[1]["names"] == [0, 1, 2] and log[1]["hint"] == [0, 0, 256]


def test_run_scheduled_drops_a_stopped_tree_and_its_hint_follows_the_others(monkeypatch):
    trees = [FakeTree(i) for i in range(4)]
    secs = {0: 4.0, 1: 3.0, 2: 2.0, 3: 1.0}
    visits = {0: 5000, 1: 10, 2: 5000, 3: 10}
    log = []
    monkeypatch.setattr(_hip, "run_sampling", _fake_run(log, secs, visits, stop_at={0: 1}))   # tree 0 fills up in the second launch
    tot = batch.run_scheduled(trees, [10, 10, 10], 0, wide_visits=4000)
    assert log[1]["names"] == [0, 1, 2, 3] and log[1]["hint"] == [256, 0, 256, 0]
    assert log[2]["names"] == [1, 2, 3] and log[2]["hint"] == [0, 256, 0]
    assert tot["status"][0] == _hip.E_CAPACITY and tot["iters_done"][0] == 15
    assert list(tot["iters_done"][1:]) == [30, 30, 30]


def test_run_scheduled_first_launch_takes_the_callers_order_and_hints(monkeypatch):
    trees = [FakeTree(i) for i in range(3)]
    log = []
    monkeypatch.setattr(_hip, "run_sampling", _fake_run(log, {0: 1.0, 1: 1.0, 2: 1.0}, {0: 1, 1: 1, 2: 1}))
    tot = batch.run_scheduled(trees, [5], 0, order=[2, 0, 1], hint=np.array([256, 0, 128], dtype=np.int32))
    assert log[0]["names"] == [2, 0, 1] and log[0]["hint"] == [256, 0, 128]
    assert tot["wide"] == 1 and tot["narrow"] == 1


# ------------------------------------------------------------------------------------------------
# run_batch: launches of at most `window` iterations, per-tree budgets, cloud refreshes, failures
# ------------------------------------------------------------------------------------------------
class PlanTree(FakeTree):
    device_id = 0


class FakeGuidance:
    def __init__(self):
        self.calls = []
        self.seconds = {"candidates": 0.0}

    def refresh(self, due, problems, trees, streams, c_best, frames):
        self.calls.append((list(due), [float(c_best[i]) for i in due]))
        return {i: (np.zeros((0, 2)), np.zeros(0, dtype=np.int64)) for i in due}


def _planner(log, script):
    """script[name] = list of (iterations the tree really runs, status, best cost after them) per launch the tree takes part in"""
    step = {}

    def run_sampling(trees, iters, flags=0, want_trace=False, iters_each=None, **kw):
        n = len(trees)
        log.append({"names": [t.name for t in trees], "iters": iters, "each": list(iters_each), "flags": flags})
        done = np.zeros(n, dtype=np.int64)
        status = np.zeros(n, dtype=np.int32)
        stats = np.zeros((n, _hip.N_STATS), dtype=np.int64)
        trace = np.full((n, iters), np.inf)
        for j, t in enumerate(trees):
            k = step.get(t.name, 0)
            step[t.name] = k + 1
            want, st, cb = script[t.name][k]
            d = min(want, int(iters_each[j]))
            done[j], status[j] = d, st
            trace[j, :d] = cb
            stats[j, _hip.ST_ITERS] = d
            stats[j, _hip.ST_T0] = 1000 * len(log)
            stats[j, _hip.ST_T1] = 1000 * len(log) + 7
            stats[j, _hip.ST_CBEST] = np.float64(cb).view(np.int64)
            stats[j, 17] = np.float64(cb).view(np.int64)
        return {"kernel_ms": 2.0, "iters_done": done, "cost_trace": trace, "stats": stats, "status": status}
    return run_sampling


def test_run_batch_windows_budgets_and_sums(dev, monkeypatch):
    trees = [PlanTree(i) for i in range(3)]
    streams = [batch.ProblemStreams(i) for i in range(3)]
    log = []
    big = 10 ** 9
    script = {i: [(big, 0, 50.0 - i)] * 3 for i in range(3)}
    monkeypatch.setattr(_hip, "run_sampling", _planner(log, script))
    r = batch.run_batch(trees, streams, 2500, _hip.F_IRRT, 2, window=1000)
    assert [l["iters"] for l in log] == [1000, 1000, 500]
    assert [l["each"] for l in log] == [[1000] * 3, [1000] * 3, [500] * 3]
    assert all(l["flags"] == _hip.F_IRRT for l in log)            # no guidance, no stop-at-first
    assert r["launches"] == 3 and r["kernel_ms"] == pytest.approx(6.0)
    assert list(r["iters_done"]) == [2500] * 3 and not r["failed"]
    for i in range(3):
        assert r["traces"][i].shape == (2500,) and np.all(r["traces"][i] == 50.0 - i)
        assert r["stats"][i, _hip.ST_ITERS] == 2500
        assert r["stats"][i, _hip.ST_T0] == 1000 and r["stats"][i, _hip.ST_T1] == 3007     # first start, last end
        assert r["stats"][i, _hip.ST_CBEST] == np.float64(50.0 - i).view(np.int64)        # an absolute value, not a sum
    # every problem's generators went to its tree once, in one call
    assert dev.set_calls == [(3, True, True)]


def test_run_batch_stop_first_retires_solved_trees(dev, monkeypatch):
    trees = [PlanTree(i) for i in range(2)]
    streams = [batch.ProblemStreams(i) for i in range(2)]
    log = []
    script = {0: [(300, 0, 42.0)], 1: [(10 ** 9, 0, np.inf), (120, 0, 77.0)]}
    monkeypatch.setattr(_hip, "run_sampling", _planner(log, script))
    r = batch.run_batch(trees, streams, 5000, _hip.F_IRRT, 2, stop_first=True, window=1000)
    assert [l["names"] for l in log] == [[0, 1], [1]]
    assert all(l["flags"] == (_hip.F_IRRT | _hip.F_STOP_FIRST) for l in log)
    assert list(r["iters_done"]) == [300, 1120]
    assert r["traces"][0][-1] == 42.0 and r["traces"][1][-1] == 77.0 and np.isinf(r["traces"][1][999])


def test_run_batch_refreshes_due_clouds_and_reports_failures(dev, monkeypatch):
    trees = [PlanTree(i) for i in range(4)]
    streams = [batch.ProblemStreams(i) for i in range(4)]
    log = []
    big = 10 ** 9
    script = {0: [(400, _hip.E_CLOUD, 90.0), (big, 0, 80.0), (big, 0, 80.0)],   # cloud due after 400 iterations, then runs to the end
              1: [(big, 0, 70.0), (big, 0, 70.0)],
              2: [(250, _hip.E_CAPACITY, np.inf)],
              3: [(10, _hip.E_ARG, np.inf)]}
    monkeypatch.setattr(_hip, "run_sampling", _planner(log, script))
    g = FakeGuidance()
    r = batch.run_batch(trees, streams, 1500, _hip.F_IRRT, 2, problems=[{}] * 4, guidance=g, frames=[None] * 4, window=65536)
    # guided runs take short launches (a tree whose cloud is due idles until its launch ends)
    assert log[0]["iters"] == 1024 and log[0]["flags"] == (_hip.F_IRRT | _hip.F_PNG)
    # init_pc for everybody (c_best = inf), then tree 0 alone with the cost the kernel stopped at
    assert g.calls[0] == ([0, 1, 2, 3], [np.inf] * 4)
    assert g.calls[1] == ([0], [90.0])
    assert len(g.calls) == 2
    assert log[1]["names"] == [0, 1] and log[1]["each"] == [1024, 476]
    assert log[2]["names"] == [0] and log[2]["each"] == [76]
    assert list(r["iters_done"]) == [1500, 1500, 250, 10]
    assert set(r["failed"]) == {2, 3} and "capacity" in r["failed"][2] and "randint" in r["failed"][3]
    assert sorted(r["clouds"]) == [0, 1, 2, 3]


def test_overlapped_refresh_due_trees_sit_one_launch_out(dev, monkeypatch):
    """Round 6 (NIRRT_BATCH_OVERLAP=1, experimental): with one group the trees whose cloud is due are refreshed WHILE the others run
    their next launch and join the launch after; every tree still runs exactly its budget with exactly its own refreshes (launch
    boundaries change no result)."""
    trees = [PlanTree(i) for i in range(3)]
    streams = [batch.ProblemStreams(i) for i in range(3)]
    log = []
    big = 10 ** 9
    script = {0: [(400, _hip.E_CLOUD, 90.0), (300, _hip.E_CLOUD, 85.0)] + [(big, 0, 80.0)] * 8,
              1: [(big, 0, 70.0)] * 8,
              2: [(1024, _hip.E_PARK, np.inf), (100, _hip.E_CLOUD, 60.0)] + [(big, 0, 55.0)] * 8}
    monkeypatch.setattr(_hip, "run_sampling", _planner(log, script))
    monkeypatch.setenv("NIRRT_BATCH_OVERLAP", "1")      # (off by default)
    g = FakeGuidance()
    r = batch.run_batch(trees, streams, 3000, _hip.F_IRRT, 2, problems=[{}] * 3, guidance=g, frames=[None] * 3, window=65536)
    assert g.calls[0] == ([0, 1, 2], [np.inf] * 3)                 # init_pc for everybody before the first launch
    # tree 0 (due after launch 1) sits launch 2 out, tree 2 (due after launch 2) launch 3, tree 0 again launch 4
    assert [l["names"] for l in log] == [[0, 1, 2], [1, 2], [1, 0], [2], [2, 0], [0], [0]]
    assert [l["each"] for l in log] == [[1024] * 3, [1024, 1024], [952, 1024], [1024], [852, 1024], [1024], [252]]
    assert g.calls[1:] == [([0], [90.0]), ([2], [60.0]), ([0], [85.0])]
    assert list(r["iters_done"]) == [3000] * 3 and not r["failed"]
    assert r["launches"] == len(log)
    assert r["host_seconds"]["refresh_hidden"] <= r["host_seconds"]["refresh"]
    # the budgets each launch hands out follow the trees, whatever launch they are in
    for l in log:
        assert len(l["each"]) == len(l["names"]) and all(0 < e <= 1024 for e in l["each"])


def test_fps_starts_twin_of_the_torch_generator():
    """ProblemStreams.fps_starts draws the FPS start indices of a forward from a numpy MT19937 seeded like torch's CPU generator:
    draw by draw what `torch.randint(0, N, (1,), generator=torch.Generator().manual_seed(seed))` returns (pointnet2_utils.py:77)"""
    import torch
    for seed in (0, 1, 7, 4242, 2 ** 31 + 5, 2 ** 32 - 1):
        s = batch.ProblemStreams(seed)
        g = torch.Generator().manual_seed(seed)
        for rep in range(200):      # (800 outputs: across a twist of the 624-word state)
            sizes = (2048 if rep % 3 else 1777, 1024, 256, 64)
            want = [int(torch.randint(0, n, (1,), generator=g, dtype=torch.long)) for n in sizes]
            assert list(s.fps_starts(sizes)) == want, (seed, rep)
        assert int(s.fps_start(2048)) == int(torch.randint(0, 2048, (1,), generator=g, dtype=torch.long))
    with pytest.raises(ValueError):
        batch.ProblemStreams(2 ** 32).fps_starts((64,))


def test_guided_windows_are_paced_by_each_trees_own_speed(dev, monkeypatch):
    """Paced windows (round 5): in a guided batch a tree whose iterations took longer than 1.25 x the median in its last launch
    gets proportionally fewer iterations in the next one (never fewer than an eighth of the window), the others keep the full
    window; every tree still runs exactly its budget, and NIRRT_BATCH_PACE=0 restores equal windows."""
    ticks_per_iter = {0: 100, 1: 110, 2: 90, 3: 400, 4: 5000}       # trees 3 and 4 are slow (4 x / 50 x the typical tree)

    def fake(log):
        def run_sampling(trees, iters, flags=0, want_trace=False, iters_each=None, **kw):
            n = len(trees)
            log.append({"names": [t.name for t in trees], "each": [int(v) for v in iters_each]})
            done = np.array([int(v) for v in iters_each], dtype=np.int64)
            stats = np.zeros((n, _hip.N_STATS), dtype=np.int64)
            for j, t in enumerate(trees):
                stats[j, _hip.ST_ITERS] = done[j]
                stats[j, _hip.ST_BUSY] = ticks_per_iter[t.name] * done[j]
                stats[j, 17] = np.float64(60.0).view(np.int64)
                stats[j, _hip.ST_CBEST] = stats[j, 17]
            return {"kernel_ms": 1.0, "iters_done": done, "cost_trace": np.full((n, iters), 60.0), "stats": stats,
                    "status": np.zeros(n, dtype=np.int32)}
        return run_sampling

    for paced in (True, False):
        monkeypatch.setenv("NIRRT_BATCH_PACE", "1" if paced else "0")
        trees = [PlanTree(i) for i in range(5)]
        streams = [batch.ProblemStreams(i) for i in range(5)]
        log = []
        monkeypatch.setattr(_hip, "run_sampling", fake(log))
        r = batch.run_batch(trees, streams, 3000, _hip.F_IRRT, 2, problems=[{}] * 5, guidance=FakeGuidance(), frames=[None] * 5, window=1024)
        assert list(r["iters_done"]) == [3000] * 5 and not r["failed"]
        assert log[0]["each"] == [1024] * 5                       # nothing is known before the first launch
        if not paced:
            assert all(e in (1024, 3000 - 2048) for l in log for e in l["each"])
            continue
        second = dict(zip(log[1]["names"], log[1]["each"]))
        assert second[0] == second[1] == second[2] == 1024         # within 1.25 x the median (110 ticks)
        assert second[3] == int(1024 * (1.25 * 110) / 400)         # proportionally fewer
        assert second[4] == 1024 // 8                              # ... but never fewer than an eighth of the window
        assert len(log) > 3                                        # the slow trees catch up over more launches
        assert log[-1]["names"] == [4]
