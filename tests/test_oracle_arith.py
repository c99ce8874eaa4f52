"""Pin the oracle's distance primitives against the Python/numpy functions the reference calls
(SURVEY.md Appendix A).  These run on every box, so they also check the platform-dependent rows
(BLAS ddot FMA chain) on the GPU box's host CPU."""
import ctypes as C
import math

import numpy as np


def _rand(n, dim, seed, scale=230.0):
    rng = np.random.default_rng(seed)
    a = rng.uniform(-scale, scale, size=(n, dim))
    # mix in short edges (typical tree edges are <= step_len) and tiny/zero components
    a[: n // 3] *= 0.05
    a[n // 3: n // 3 + 50, 0] = 0.0
    a[n // 3 + 50: n // 3 + 100] *= 1e-9
    return a


def test_hypot_py_matches_math_hypot(oracle):
    L = oracle.lib()
    for dim in (2, 3):
        a = _rand(200000, dim, 1 + dim)
        bad = 0
        for row in a:
            got = L.orc_hypot_py(dim, row.ctypes.data_as(C.POINTER(C.c_double)))
            exp = math.hypot(*row)
            bad += got != exp
        assert bad == 0


def test_hypot_np_and_restatement_match_numpy(oracle):
    L = oracle.lib()
    a = _rand(200000, 2, 5)
    exp = np.hypot(a[:, 0], a[:, 1])
    for x, y, e in zip(a[:, 0], a[:, 1], exp):
        assert L.orc_hypot_np(x, y) == e
        assert L.orc_hypot_glibc_restated(x, y) == e
    # scaling branches of the restatement
    for x, y in [(1e300, 1e300), (1e-300, 1e-300), (1e200, 1e-200), (0.0, 0.0), (3.0, 0.0), (0.0, -4.0),
                 (1e-310, 1e-310), (2e-160, 1e-160), (5e153, 5e153), (1e154, 3e153)]:
        assert L.orc_hypot_glibc_restated(x, y) == float(np.hypot(x, y)), (x, y)


def test_norm_axis_matches_numpy():
    # np.linalg.norm(A, axis=1) == sqrt of left-to-right unfused sum of squares (nearest 3D, Near 3D)
    for dim in (2, 3):
        a = _rand(200000, dim, 9 + dim)
        exp = np.linalg.norm(a, axis=1)
        s = a[:, 0] * a[:, 0] + a[:, 1] * a[:, 1]
        if dim == 3:
            s = s + a[:, 2] * a[:, 2]
        assert np.array_equal(np.sqrt(s), exp)


def test_blas_1d_norm_and_dot_fma_chain(oracle):
    """np.linalg.norm(v) / np.dot on 2-/3-vectors == forward FMA chain (platform dependent row)."""
    L = oracle.lib()
    dp = C.POINTER(C.c_double)
    for dim in (2, 3):
        a = _rand(20000, dim, 20 + dim)
        b = _rand(20000, dim, 30 + dim)
        bad_n = bad_d = 0
        for u, w in zip(a, b):
            bad_n += L.orc_norm_1d(dim, u.ctypes.data_as(dp)) != float(np.linalg.norm(u))
            bad_d += L.orc_dot_blas(dim, u.ctypes.data_as(dp), w.ctypes.data_as(dp)) != float(np.dot(u, w))
        assert bad_n == 0 and bad_d == 0
