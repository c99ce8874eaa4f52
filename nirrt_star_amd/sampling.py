"""Host side of the RNG contract (SURVEY.md §5 "RNG", Appendix B).

The reference draws from three process-global generators; two matter for the tree loop:
  * numpy's legacy global RandomState  (SampleFree, 3D SampleUnitBall, SamplePointCloud, ...)
  * CPython's `random` module           (2D SampleUnitBall)
Both are MT19937 and build doubles as (a>>5, b>>6) -> (a*2^26+b)/2^53 from two 32-bit outputs.

The device-resident loop consumes those generators data-dependently (rejection loops), so the
host hands it the upcoming raw 32-bit outputs (`peek_*`) and afterwards advances the real generators
by exactly the number of words the kernel used (`advance_*`).  Net effect on the process-global
state = what the reference's own Python loop would have left behind.
"""
import random

import numpy as np


_SCRATCH = np.random.RandomState(0)   # carrier of a copied python-generator state (peek_py_words)


def peek_np_words(n):
    """next n raw MT19937 outputs of numpy's global RandomState, without consuming them"""
    st = np.random.get_state()
    w = np.random.randint(0, 1 << 32, size=int(n), dtype=np.uint32)  # full range: one output per value
    np.random.set_state(st)
    return w


def advance_np_words(n):
    if n:
        np.random.randint(0, 1 << 32, size=int(n), dtype=np.uint32)


def peek_py_words(n):
    """next n raw MT19937 outputs of python's `random`, without consuming them"""
    n = int(n)
    if n == 0:
        return np.zeros(0, dtype=np.uint32)
    # getrandbits(32 n) would be n consecutive outputs, least-significant word first; the same MT19937 state inside a numpy
    # RandomState yields them 3-4x faster than building (and splitting) a multi-megabit Python integer
    st = random.getstate()[1]
    _SCRATCH.set_state(("MT19937", np.array(st[:624], dtype=np.uint32), int(st[624])))
    return _SCRATCH.randint(0, 1 << 32, size=n, dtype=np.uint32)


def advance_py_words(n):
    if n:
        random.getrandbits(32 * int(n))


def words_to_doubles(w):
    """numpy random_sample() / python random.random() construction"""
    w = np.asarray(w, dtype=np.uint64)
    a, b = w[0::2] >> np.uint64(5), w[1::2] >> np.uint64(6)
    return (a.astype(np.float64) * 67108864.0 + b.astype(np.float64)) / 9007199254740992.0


def informed_frame(x_start, x_goal):
    """IRRTStar.init (irrt_star_2d.py:35-40, :153-161 / irrt_star_3d.py:32-36, :160-172):
    (c_min, x_center (dim,), C (3,3)) with the reference's own numpy/math calls."""
    import math
    xs = np.asarray(x_start, dtype=np.float64)
    xg = np.asarray(x_goal, dtype=np.float64)
    dim = len(xs)
    if dim == 2:
        dx, dy = xg - xs
        c_min = math.hypot(dx, dy)
        a1 = np.zeros((3, 1))
        a1[:2, 0] = (xg - xs) / c_min
        e1 = np.array([[1.0], [0.0], [0.0]])
        M = a1 @ e1.T
        U, _, V_T = np.linalg.svd(M, True, True)
        C = U @ np.diag([1.0, 1.0, np.linalg.det(U) * np.linalg.det(V_T.T)]) @ V_T
    else:
        dx, dy, dz = xg - xs
        c_min = math.hypot(dx, dy, dz)
        a1 = (xg - xs) / c_min
        M = np.outer(a1, [1, 0, 0])
        U, S, V = np.linalg.svd(M)
        C = U @ np.diag([1, 1, np.linalg.det(U) * np.linalg.det(V)]) @ V.T
    return c_min, (xs + xg) / 2.0, C
