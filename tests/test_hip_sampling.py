"""GPU: device-resident loop with IN-KERNEL sampling vs the seeded golden runs (L2 parity):
same seeds -> same node_rand stream -> same tree, and the host generators end up where the
reference's own loop would have left them."""
import random

import numpy as np
import pytest

from conftest import load_golden
from test_hip_parity import make_hip_tree

pytestmark = pytest.mark.gpu


def _run(g, flags, iters=None, np_budget=None, py_budget=None):
    from nirrt_star_amd import _hip, sampling
    dim = int(g["dim"])
    iters = int(iters if iters is not None else g["iter_max"])
    t = make_hip_tree(g)
    seed = int(g["seed"])
    np.random.seed(seed)
    random.seed(seed)
    if flags & _hip.F_IRRT:
        c_min, xc, C = sampling.informed_frame(g["x_start"], g["x_goal"])
        t.set_informed(c_min, xc, C)
    npw = sampling.peek_np_words(np_budget or iters * dim * 2 * 4)
    pyw = sampling.peek_py_words(py_budget or iters * 2 * 2 * 3) if dim == 2 and flags & _hip.F_IRRT else None
    res = _hip.run_sampling([t], iters, [npw], [pyw] if pyw is not None else None, flags=flags, want_trace=True)
    return t, res, npw, pyw


@pytest.mark.parametrize("name", ["run_rrt2d_500", "run_rrt2d_3000", "run_rrt3d_3000"])
def test_rrt_in_kernel_sample_free(name):
    from nirrt_star_amd import _hip
    g = load_golden(name)
    dim = int(g["dim"])
    t, res, npw, _ = _run(g, 0)
    assert res["iters_done"][0] == int(g["iter_max"]) and res["status"][0] == 0
    v, p = t.download()
    assert len(v) == int(g["n"]) and np.array_equal(p, g["parents"])
    assert np.array_equal(v, g["vertices"])
    # words consumed = 2*dim per SampleFree attempt; the last accepted attempt produced the last sample
    used = int(res["np_used"][0])
    assert used % (2 * dim) == 0
    from nirrt_star_amd import sampling
    d = sampling.words_to_doubles(npw[:used]).reshape(-1, dim)
    lo, hi = (3.0, 221.0) if dim == 2 else (2.0, 48.0)
    assert np.array_equal(lo + (hi - lo) * d[-1], g["samples"][-1])
    t.close()


@pytest.mark.parametrize("name", ["run_irrt2d_800", "run_irrt2d_3000", "run_irrt2d_free_5000"])
def test_irrt2d_in_kernel_informed_sampling(name):
    from nirrt_star_amd import _hip
    g = load_golden(name)
    t, res, npw, pyw = _run(g, _hip.F_IRRT)
    assert res["iters_done"][0] == int(g["iter_max"]) and res["status"][0] == 0
    v, p = t.download()
    assert len(v) == int(g["n"]) and np.array_equal(p, g["parents"])
    assert np.array_equal(v, g["vertices"])        # (round 5: bit-equal - the steer is the reference's libm, restated)
    assert np.array_equal(t.solutions, g["path_solutions"])
    if len(g["path_solutions"]):
        c, x = t.best_solution()
        assert abs(c - float(g["path_len"])) <= 1e-5
        assert int(res["py_used"][0]) > 0
    t.close()


def test_irrt3d_in_kernel_sampling_bit_equal():
    """3D informed sampling goes through np.sin / np.cos - libm's for float64 in numpy 2.2 - which the device restates since round 5
    (csrc/glibc235_libm.inc): the reference's tree bit for bit (rounds 1-4: the device's own sin / cos, vertices <= 1e-9)."""
    from nirrt_star_amd import _hip
    g = load_golden("run_irrt3d_3000")
    t, res, npw, _ = _run(g, _hip.F_IRRT, np_budget=3000 * 6 * 60)   # informed rejection loops are long in 3D
    assert res["iters_done"][0] == int(g["iter_max"]) and res["status"][0] == 0
    v, p = t.download()
    assert len(v) == int(g["n"])
    assert np.array_equal(p, g["parents"])
    assert np.array_equal(v, g["vertices"])
    assert np.array_equal(t.solutions, g["path_solutions"])
    t.close()


def test_stream_exhaustion_stops_cleanly_and_resumes():
    from nirrt_star_amd import _hip, sampling
    g = load_golden("run_rrt2d_500")
    iters = int(g["iter_max"])
    t, res, npw, _ = _run(g, 0, np_budget=600)   # far too few words
    done = int(res["iters_done"][0])
    assert 0 < done < iters and res["status"][0] == _hip.E_STREAM
    used = int(res["np_used"][0])
    assert used <= 600 and used % 4 == 0
    # resume with the rest of the stream: same final tree as the uninterrupted run
    np.random.seed(int(g["seed"]))
    sampling.advance_np_words(used)
    rest = sampling.peek_np_words(iters * 16)
    res2 = _hip.run_sampling([t], iters - done, [rest], None, flags=0)
    assert res2["iters_done"][0] == iters - done and res2["status"][0] == 0
    v, p = t.download()
    assert len(v) == int(g["n"]) and np.array_equal(p, g["parents"])
    t.close()
