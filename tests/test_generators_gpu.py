"""GPU: the trees' own MT19937 generators (nirrt_set_generators / nirrt_get_generators / nirrt_generator_words and
nirrt_run with np_words == NULL).  The reference draws from numpy's legacy RandomState and CPython's random inside its loop
(rrt_base_2d.py:46-52, irrt_star_2d.py:121-151, irrt_star_3d.py:117-158); here the recurrence runs in the tree's wave.
Checked against the generators themselves: raw outputs, get_state() / getstate() after the fact, and - through whole
planning loops - identical trees, word counts and final states as the path that feeds host-produced words."""
import random

import numpy as np
import pytest

from conftest import load_golden
from test_hip_parity import make_hip_tree

pytestmark = pytest.mark.gpu


def _np_words(rs, n):
    return rs.randint(0, 1 << 32, size=int(n), dtype=np.uint32)


def _py_words(rnd, n):
    v = rnd.getrandbits(32 * int(n))
    return np.array([(v >> (32 * i)) & 0xFFFFFFFF for i in range(int(n))], dtype=np.uint32)


def test_device_twists_equal_numpy_and_cpython_generators():
    """twelve and more blocks per stream, resumed mid-block, from seeded and mid-block initial states; outputs, the
    library's host generator, and the final get_state() / getstate() all agree"""
    from nirrt_star_amd import _hip
    g = load_golden("run_rrt2d_500")
    trees = [make_hip_tree(g) for _ in range(5)]
    seeds = [0, 1, 1234, 2 ** 31 + 7, 99]
    rss = [np.random.RandomState(s) for s in seeds]
    pys = [random.Random(s) for s in seeds]
    # generators 2.. start in the middle of a block (also position 0 of a fresh block via exactly 624 outputs)
    pre = [0, 0, 17, 624, 1000]
    for rs, rnd, k in zip(rss, pys, pre):
        if k:
            _np_words(rs, k)
            _py_words(rnd, k)
    _hip.set_generators(trees, [_hip.np_state(rs) for rs in rss], [_hip.py_state(r) for r in pys])
    lib = [[_hip.np_state(rs), _hip.py_state(r)] for rs, r in zip(rss, pys)]     # the library's host generator, same states
    for n in [5, 1, 700, 63, 64, 65, 624 * 10 + 3, 623, 1, 1248, 7]:
        for which in (0, 1):
            out = _hip.generator_words(trees, which, n)
            for i in range(len(trees)):
                ref = _np_words(rss[i], n) if which == 0 else _py_words(pys[i], n)
                assert np.array_equal(out[i], ref), (n, which, i)
                w, k2, p2 = _hip.mt19937_outputs(lib[i][which][0], lib[i][which][1], n)
                lib[i][which] = (k2, p2)
                assert np.array_equal(w, ref)
        nk, npos, pk, ppos = _hip.get_generators(trees)
        for i in range(len(trees)):
            k, p = _hip.np_state(rss[i])
            assert p == npos[i] and np.array_equal(k, nk[i]), ("numpy state", n, i)
            k, p = _hip.py_state(pys[i])
            assert p == ppos[i] and np.array_equal(k, pk[i]), ("python state", n, i)
    # a state handed over and read back untouched is the state itself (positions 0, mid-block and 624)
    states = [((np.arange(624, dtype=np.uint64) * 2654435761 + p) % (1 << 32), p) for p in (0, 1, 300, 623, 624)]
    states = [(np.asarray(k, dtype=np.uint32), p) for k, p in states]
    _hip.set_generators(trees, states, states)
    nk, npos, pk, ppos = _hip.get_generators(trees)
    for i, (k, p) in enumerate(states):
        assert npos[i] == p and ppos[i] == p and np.array_equal(nk[i], k) and np.array_equal(pk[i], k)
    # ... and only one stream can be set / read
    _hip.set_generators(trees, None, [_hip.py_state(r) for r in pys])
    nk2, npos2, pk2, ppos2 = _hip.get_generators(trees)
    assert np.array_equal(nk2, nk) and np.array_equal(npos2, npos)
    assert all(np.array_equal(pk2[i], _hip.py_state(pys[i])[0]) for i in range(len(trees)))
    for t in trees:
        t.close()


def _host_fed(g, flags, iters, np_budget, py_budget):
    from nirrt_star_amd import _hip, sampling
    dim = int(g["dim"])
    t = make_hip_tree(g, iter_max=iters)
    seed = int(g["seed"])
    np.random.seed(seed)
    random.seed(seed)
    if flags & _hip.F_IRRT:
        t.set_informed(*sampling.informed_frame(g["x_start"], g["x_goal"]))
    npw = sampling.peek_np_words(np_budget)
    pyw = sampling.peek_py_words(py_budget) if dim == 2 and flags & _hip.F_IRRT else None
    res = _hip.run_sampling([t], iters, [npw], [pyw] if pyw is not None else None, flags=flags, want_trace=True)
    sampling.advance_np_words(int(res["np_used"][0]))
    sampling.advance_py_words(int(res["py_used"][0]))
    return t, res


def _own_generators(g, flags, iters, pieces, variant_env=None):
    """the same run drawing from the tree's own generators, in `pieces` launches (stop / resume across launch boundaries)"""
    from nirrt_star_amd import _hip, sampling
    t = make_hip_tree(g, iter_max=iters)
    seed = int(g["seed"])
    rs, rnd = np.random.RandomState(seed), random.Random(seed)
    if flags & _hip.F_IRRT:
        t.set_informed(*sampling.informed_frame(g["x_start"], g["x_goal"]))
    _hip.set_generators([t], [_hip.np_state(rs)], [_hip.py_state(rnd)])
    done, used_np, used_py, traces = 0, 0, 0, []
    for k, n in enumerate(pieces):
        res = _hip.run_sampling([t], n, None, None, flags=flags, want_trace=True)
        assert res["status"][0] == 0
        d = int(res["iters_done"][0])
        assert d == n
        done += d
        used_np += int(res["np_used"][0])
        used_py += int(res["py_used"][0])
        traces.append(res["cost_trace"][0, :d])
    return t, done, used_np, used_py, np.concatenate(traces)


@pytest.mark.parametrize("name,flags_irrt,iters,pieces", [
    ("run_rrt2d_3000", 0, 3000, [3000]),
    ("run_irrt2d_3000", 1, 3000, [1, 999, 2000]),
    ("run_irrt2d_3000", 1, 9000, [9000]),
    ("run_rrt3d_3000", 0, 3000, [1500, 1500]),
    ("run_irrt3d_3000", 1, 3000, [3000]),
])
def test_loop_with_own_generators_equals_host_fed_words(name, flags_irrt, iters, pieces):
    """identical tree, best-cost trace, word counts and FINAL GENERATOR STATES whether the words come from the host or from
    the tree's own generators (early draws undone across block boundaries included: ~100 blocks per stream in the 9000-
    iteration run)"""
    from nirrt_star_amd import _hip
    g = load_golden(name)
    flags = _hip.F_IRRT if flags_irrt else 0
    dim = int(g["dim"])
    t0, r0 = _host_fed(g, flags, iters, iters * (300 if dim == 3 and flags_irrt else 40), iters * 40)
    assert r0["status"][0] == 0 and r0["iters_done"][0] == iters
    t1, done, used_np, used_py, trace = _own_generators(g, flags, iters, pieces)
    assert done == iters
    v0, p0 = t0.download()
    v1, p1 = t1.download()
    assert np.array_equal(p0, p1) and np.array_equal(v0, v1)
    assert np.array_equal(t0.solutions, t1.solutions)
    assert np.array_equal(trace, r0["cost_trace"][0, :iters])
    assert used_np == int(r0["np_used"][0]) and used_py == int(r0["py_used"][0])
    if iters == int(g["iter_max"]):      # ... which is the reference's own run
        assert np.array_equal(p1, g["parents"])
    nk, npos, pk, ppos = _hip.get_generators([t1])
    k, p = _hip.np_state()               # the process-global generators, advanced by the host-fed path
    assert p == npos[0] and np.array_equal(k, nk[0])
    k, p = _hip.py_state()
    assert p == ppos[0] and np.array_equal(k, pk[0])
    t0.close()
    t1.close()


def test_batch_of_trees_draws_from_its_own_generators():
    """many trees in one launch (the slim instantiation), every one seeded differently: each equals the same problem run
    alone, and the generator states come back per tree"""
    from nirrt_star_amd import _hip, sampling
    g = load_golden("run_irrt2d_800")
    iters, B = 800, 70
    trees = [make_hip_tree(g, iter_max=iters) for _ in range(B)]
    frame = sampling.informed_frame(g["x_start"], g["x_goal"])
    for t in trees:
        t.set_informed(*frame)
    seeds = [int(g["seed"])] + [5000 + i for i in range(B - 1)]
    _hip.set_generators(trees, [_hip.np_state(np.random.RandomState(s)) for s in seeds], [_hip.py_state(random.Random(s)) for s in seeds])
    res = _hip.run_sampling(trees, iters, None, None, flags=_hip.F_IRRT)
    assert (res["status"] == 0).all() and (res["iters_done"] == iters).all()
    v, p = trees[0].download()
    assert np.array_equal(p, g["parents"])     # tree 0 is the reference's run
    nk, npos, pk, ppos = _hip.get_generators(trees)
    for i in (0, 1, 33, B - 1):
        rs, rnd = np.random.RandomState(seeds[i]), random.Random(seeds[i])
        _np_words(rs, int(res["np_used"][i]))
        if res["py_used"][i]:
            _py_words(rnd, int(res["py_used"][i]))
        k, pos = _hip.np_state(rs)
        assert pos == npos[i] and np.array_equal(k, nk[i])
        k, pos = _hip.py_state(rnd)
        assert pos == ppos[i] and np.array_equal(k, pk[i])
    for t in trees:
        t.close()
