"""ProblemStreams (nirrt_star_amd/batch.py): the kept look-ahead of generator outputs equals a fresh peek at every
position, also after somebody else drew from the numpy generator (cloud candidates, pointcloud.py)."""
import numpy as np
import pytest

from nirrt_star_amd.batch import ProblemStreams


def test_lookahead_windows_equal_fresh_peeks_after_foreign_draws():
    a, b = ProblemStreams(7), ProblemStreams(7)
    for step in range(6):
        n = 5000 - step * 500
        assert np.array_equal(a.window_np(n), b.peek_np(n))
        assert np.array_equal(a.window_py(n), b.peek_py(n))
        for s in (a, b):
            s.advance_np(1234 + step)
            s.advance_py(77)
            s.rs.random_sample(333 + step)          # foreign draws: doubles, masked-rejection integers, a 2-D uniform block
            s.rs.randint(0, 17, size=5)
            s.rs.uniform(0, 3, size=(10, 2))
    assert a._npc["off"] > 0 and a._pyc["off"] > 0   # served from the look-ahead, not regenerated
    big = len(a._npc["host"]) + 10
    assert np.array_equal(a.window_np(big), b.peek_np(big))      # beyond the look-ahead: regenerated from the true position
    assert a.rs.random_sample() == b.rs.random_sample()          # windows never consume


def test_generators_end_at_the_consumed_position():
    a = ProblemStreams(3)
    ref = np.random.RandomState(3)
    a.window_np(1000)
    a.advance_np(10)
    ref.randint(0, 1 << 32, size=10, dtype=np.uint32)
    assert a.rs.random_sample() == ref.random_sample()


def test_advancing_jumps_through_the_lookahead_like_generate_and_discard():
    for seed in (1, 99):
        a, b = ProblemStreams(seed), ProblemStreams(seed)
        a.rs.normal(); b.rs.normal()                  # a cached gaussian has to survive the jumps
        a.py.gauss(0, 1); b.py.gauss(0, 1)
        a.window_np(300000)
        a.window_py(300000)
        for k in (5, 700, 12345, 100000):             # 5: fewer than 624 known outputs before the new position -> generated
            a.advance_np(k)
            b.rs.randint(0, 1 << 32, size=k, dtype=np.uint32)
            a.advance_py(k)
            b.py.getrandbits(32 * k)
            assert np.array_equal(a.window_np(1000), b.peek_np(1000))
            assert np.array_equal(a.window_py(1000), b.peek_py(1000))
            assert a.rs.get_state()[2] in (b.rs.get_state()[2], 624)
        assert a.rs.random_sample() == b.rs.random_sample() and a.py.random() == b.py.random()
        assert a.rs.normal() == b.rs.normal() and a.py.gauss(0, 1) == b.py.gauss(0, 1)


def test_native_mt19937_fill_equals_numpy_and_cpython():
    """nirrt_mt19937_fill (host code of the library): the raw outputs, and the state it leaves behind, are numpy's RandomState's
    and CPython's random.Random's - from any position inside a block, across block boundaries, for n = 0"""
    import random
    from nirrt_star_amd import _hip
    for seed, skip, n in ((1, 0, 5), (2, 3, 624), (3, 611, 2000), (4, 624 * 2 + 7, 1), (5, 10, 0)):
        rs = np.random.RandomState(seed)
        rs.randint(0, 1 << 32, size=skip, dtype=np.uint32)
        st = rs.get_state(legacy=True)
        want = rs.randint(0, 1 << 32, size=n, dtype=np.uint32)
        got, key, pos = _hip.mt19937_outputs(st[1], st[2], n)
        after = rs.get_state(legacy=True)
        assert np.array_equal(got, want)
        if n:
            assert pos == after[2] and np.array_equal(key, after[1])
        r = random.Random(seed)
        if skip:
            r.getrandbits(32 * skip)
        ps = r.getstate()[1]
        v = r.getrandbits(32 * n) if n else 0
        got, _, _ = _hip.mt19937_outputs(np.array(ps[:624], dtype=np.uint32), ps[624], n)
        assert [int(x) for x in got] == [(v >> (32 * i)) & 0xFFFFFFFF for i in range(n)]
    with pytest.raises(ValueError):
        _hip.mt19937_outputs(np.zeros(10, dtype=np.uint32), 0, 4)
