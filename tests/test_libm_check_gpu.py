"""GPU: the load-time check of the restated libm routines (csrc/glibc235_libm.inc) against THIS host's libm, and what happens when
they differ.  The reference steers with math.atan2 / math.cos / math.sin (rrt_star_2d.py:67-78) - whatever libm its host has; the
device evaluates glibc 2.35's x86-64 FMA variants.  nirrt_libm_probe evaluates them on the device for arguments the caller chooses,
_hip.libm_check compares with the host bit for bit, and on a mismatch the planner classes default to mode="exact" (steer on the
host between two launches) - forced here by flipping table words for the duration of a probe (NIRRT_LIBM_PROBE_FLIP)."""
import math
import random
import warnings

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _host_is_the_restated_libm():
    import platform
    if platform.machine() != "x86_64" or platform.libc_ver() != ("glibc", "2.35"):
        return False
    with open("/proc/cpuinfo") as f:
        flags = next((l for l in f if l.startswith("flags")), "")
    return " fma " in flags + " " and " avx2 " in flags + " "


def test_probe_equals_the_hosts_libm_and_reports_unsupported_arguments_as_nan():
    import ctypes as C
    from nirrt_star_amd import _hip
    if not _host_is_the_restated_libm():
        pytest.skip("another libm / CPU: the check is expected to fail here, see the next test for what follows")
    assert _hip.libm_check(force=True) is True
    # the translation does not cover the huge-argument reduction: NaN, which steer turns into NIRRT_E_LIBM instead of a vertex
    L = _hip.load()
    a = np.array([0.5, 1e9, np.inf, -3.0], dtype=np.float64)
    out = np.zeros(4)
    dp = C.POINTER(C.c_double)
    assert L.nirrt_libm_probe(1, 4, a.ctypes.data_as(dp), a.ctypes.data_as(dp), out.ctypes.data_as(dp), 0) == 0
    assert out[0] == math.sin(0.5) and out[3] == math.sin(-3.0) and np.isnan(out[1]) and np.isnan(out[2])


def test_a_libm_mismatch_warns_once_and_makes_exact_the_default_mode(monkeypatch):
    from nirrt_star_amd import _hip, planners
    from nirrt_star_amd.env import Env, Env3D
    g = load_golden("run_rrt2d_500")
    monkeypatch.setenv("NIRRT_LIBM_PROBE_FLIP", "1")
    try:
        with pytest.warns(RuntimeWarning, match="libm differs"):
            assert _hip.libm_check(force=True) is False
        with warnings.catch_warnings():
            warnings.simplefilter("error")           # the cached verdict does not warn again
            p = planners.RRTStar2D(tuple(g["x_start"]), tuple(g["x_goal"]), 10, float(g["search_radius"]), int(g["iter_max"]), Env(g["env"]),
                                   int(g["clearance"]))
            g3 = load_golden("run_rrt3d_500")
            p3 = planners.RRTStar3D(tuple(g3["x_start"]), tuple(g3["x_goal"]), 10, float(g3["search_radius"]), 10, Env3D(g3["env"]), 2)
        assert p.mode == "exact" and p3.mode == "resident"      # (3D RRT* needs IEEE operations only)
        # ... and that mode reproduces the reference on this host whatever the device tables hold: steer with the host's libm
        np.random.seed(int(g["seed"]))
        random.seed(int(g["seed"]))
        p.planning()
        n = p.num_vertices
        assert n == int(g["n"]) and np.array_equal(p.vertex_parents[:n], g["parents"]) and np.array_equal(p.vertices[:n], g["vertices"])
        # an explicit mode is honoured
        q = planners.RRTStar2D(tuple(g["x_start"]), tuple(g["x_goal"]), 10, float(g["search_radius"]), 10, Env(g["env"]), int(g["clearance"]), mode="resident")
        assert q.mode == "resident"
    finally:
        monkeypatch.delenv("NIRRT_LIBM_PROBE_FLIP", raising=False)
        _hip._libm_ok = None      # the next planner of this process probes again (without the flip)
    if _host_is_the_restated_libm():
        assert _hip.libm_check(force=True) is True
