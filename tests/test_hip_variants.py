"""GPU: every kernel instantiation under the parity suite.

The library compiles each kernel three times (csrc/nirrt_hip.hip): `slim` (64 threads, ONE wave per tree - what the
default bench of 4096 problems per GPU runs), `narrow` (128 threads) and `wide` (256 threads); nirrt_run normally picks
by batch size.  NIRRT_FORCE_VARIANT pins every launch (primitives, step kernel, persistent loops) to one of them, so
the fixture / oracle comparisons of test_hip_parity / test_hip_sampling / test_hip_grid run whole under each.  On top:
one launch of more than 2048 trees (the natural dispatch to `slim`) with trees compared against the oracle, and the
suite once more against a build with tiny compile-time limits (parent chains longer than the LDS chain cache, Near sets
larger than the LDS stash), loaded through NIRRT_HIP_SO in a child process.

Reference functions matched: rrt_star_2d.py:37-99, irrt_star_2d.py:42-97 (and the 3D twins)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden, make_oracle_tree
import test_hip_grid as G
import test_hip_parity as P
import test_hip_sampling as S

pytestmark = pytest.mark.gpu

VARIANTS = ["slim", "narrow", "wide"]


@pytest.fixture(params=VARIANTS)
def forced_variant(request, monkeypatch):
    monkeypatch.setenv("NIRRT_FORCE_VARIANT", request.param)
    return request.param


@pytest.fixture
def forced_slim_grid(monkeypatch):
    """one wave per tree AND the grid index from 32 vertices on (rebuild every 8): cell rows, tail, widening, ordering"""
    monkeypatch.setenv("NIRRT_FORCE_VARIANT", "slim")
    monkeypatch.setenv("NIRRT_GRID_MIN", "32")
    monkeypatch.setenv("NIRRT_GRID_REBUILD", "8")


@pytest.mark.parametrize("name", P.RUNS)
def test_variant_step_replay(forced_variant, name):
    P.test_step_replay_device_steer(name)


@pytest.mark.parametrize("name", ["run_rrt2d_3000", "run_irrt2d_3000", "run_irrt3d_3000"])
def test_variant_host_steer_bit_exact(forced_variant, oracle, name):
    P.test_step_replay_host_steer_bit_exact(oracle, name)


@pytest.mark.parametrize("name", ["run_rrt2d_3000", "run_irrt3d_3000"])
def test_variant_primitives_on_frozen_tree(forced_variant, oracle, name):
    P.test_primitives_on_frozen_tree(oracle, name)


def test_variant_resident_replay(forced_variant, oracle):
    P.test_resident_loop_replay_many_trees()
    P.test_resident_loop_3d_with_goal_trace(oracle)


@pytest.mark.parametrize("name", ["run_rrt2d_500", "run_rrt2d_3000", "run_rrt3d_3000"])
def test_variant_in_kernel_sampling_rrt(forced_variant, name):
    S.test_rrt_in_kernel_sample_free(name)


@pytest.mark.parametrize("name", ["run_irrt2d_800", "run_irrt2d_3000"])
def test_variant_in_kernel_sampling_irrt(forced_variant, name):
    S.test_irrt2d_in_kernel_informed_sampling(name)


def test_variant_in_kernel_sampling_irrt3d_and_resume(forced_variant):
    S.test_irrt3d_in_kernel_sampling_bit_equal()
    S.test_stream_exhaustion_stops_cleanly_and_resumes()


@pytest.mark.parametrize("name", P.RUNS)
def test_slim_grid_step_replay(forced_slim_grid, name):
    P.test_step_replay_device_steer(name)


@pytest.mark.parametrize("name", ["run_rrt2d_3000", "run_rrt3d_3000"])
def test_slim_grid_sampling_rrt(forced_slim_grid, name):
    S.test_rrt_in_kernel_sample_free(name)


@pytest.mark.parametrize("name", ["run_irrt2d_800", "run_irrt2d_3000"])
def test_slim_grid_sampling_irrt(forced_slim_grid, name):
    S.test_irrt2d_in_kernel_informed_sampling(name)


@pytest.mark.parametrize("name", ["run_rrt2d_3000", "run_irrt3d_3000"])
def test_slim_grid_primitives_after_a_run(forced_slim_grid, oracle, name):
    G.test_grid_primitives_after_a_run(oracle, "default-cells", name)


@pytest.mark.parametrize("dim,irrt,iters", [(2, False, 12000), (2, True, 9000), (3, False, 12000), (2, True, 22000)])
def test_slim_default_index_against_the_oracle_at_mid_size(monkeypatch, oracle, dim, irrt, iters):
    monkeypatch.setenv("NIRRT_FORCE_VARIANT", "slim")
    G.test_default_index_against_the_oracle_at_mid_size(oracle, dim, irrt, iters)


@pytest.mark.parametrize("irrt", [False, True])
def test_more_than_2048_trees_in_one_launch_against_the_oracle(oracle, irrt):
    """The dispatch rule itself: 2112 different problems in one nirrt_run -> the one-wave-per-tree kernels.  Every tree has
    its own world / start / goal and its own generator words; six of them (first, last, some between) are re-run by the
    oracle on the same words: same vertex count, parents, solution list and words consumed."""
    from nirrt_star_amd import _hip, sampling, worlds
    assert "NIRRT_FORCE_VARIANT" not in os.environ and "NIRRT_SLIM_MIN_TREES" not in os.environ
    B, iters = 2112, 2500
    cache, probs, trees = {}, [], []
    for b in range(B):
        w = b % 16
        if w not in cache:
            cache[w] = worlds.random_world_2d(w, "b30")
        pr = worlds.problem_2d(cache[w], (b // 16) % 4)
        probs.append(pr)
        t = _hip.HipTree(2, iters, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], 3.0, pr["env"])
        t.set_informed(*sampling.informed_frame(pr["x_start"], pr["x_goal"]))
        trees.append(t)
    rs = np.random.RandomState(5)
    base_np = rs.randint(0, 1 << 32, size=iters * 8 + B + 4096, dtype=np.uint32)
    base_py = rs.randint(0, 1 << 32, size=iters * 16 + B + 4096, dtype=np.uint32)
    npw = [base_np[b:] for b in range(B)]            # a different stream per tree (shifted windows of one pool)
    pyw = [base_py[b:] for b in range(B)]
    flags = _hip.F_IRRT if irrt else 0
    res = _hip.run_sampling(trees, iters, npw, pyw if irrt else None, flags=flags)
    assert (res["iters_done"] == iters).all() and not res["status"].any()
    for b in (0, 1, 777, 1500, 2048, B - 1):
        pr = probs[b]
        o = oracle.OracleTree(2, iters, pr["x_start"], pr["x_goal"], 10.0, float(pr["search_radius"]), 3.0, pr["env_dict"])
        ro = o.run_sampling(iters, np.ascontiguousarray(npw[b]), np.ascontiguousarray(pyw[b]) if irrt else None, irrt=irrt,
                            frame=sampling.informed_frame(pr["x_start"], pr["x_goal"]))
        assert ro["iters_done"] == iters
        assert int(res["np_used"][b]) == ro["np_used"] and int(res["py_used"][b]) == ro["py_used"]
        v, p = trees[b].download()
        assert len(v) == o.n and np.array_equal(p, o.parents)
        assert np.array_equal(v, o.vertices)
        if irrt:
            assert np.array_equal(trees[b].solutions, o.solutions)
        o.close()
    for t in trees:
        t.close()


def test_lanes_hint_groups_run_concurrently_and_change_nothing():
    """nirrt_run_args.lanes_hint: trees of one call split into 256- / 128- / 64-lane groups launched side by side; every
    output lands at its tree's own index and equals the un-hinted run"""
    from nirrt_star_amd import _hip, sampling, worlds
    iters, B = 4000, 7
    outs = []
    for hint in (None, [0, 256, 128, 64, 0, 256, 128]):
        trees, npw, pyw = [], [], []
        for b in range(B):
            pr = worlds.problem_2d(worlds.random_world_2d(80 + b, "b30"), 0)
            t = _hip.HipTree(2, iters, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], 3.0, pr["env"])
            t.set_informed(*sampling.informed_frame(pr["x_start"], pr["x_goal"]))
            trees.append(t)
            rs = np.random.RandomState(900 + b)
            npw.append(rs.randint(0, 1 << 32, size=iters * 8 + 4096, dtype=np.uint32))
            pyw.append(rs.randint(0, 1 << 32, size=iters * 16 + 4096, dtype=np.uint32))
        each = [iters, iters - 7, iters, 1234, iters, iters, 77]
        r = _hip.run_sampling(trees, iters, npw, pyw, flags=_hip.F_IRRT, want_trace=True, iters_each=each, lanes_hint=hint)
        assert list(r["iters_done"]) == each and not r["status"].any()
        outs.append((r, [t.download() for t in trees], [list(t.solutions) for t in trees]))
        for t in trees:
            t.close()
    (ra, da, sa), (rb, db, sb) = outs
    assert np.array_equal(ra["np_used"], rb["np_used"]) and np.array_equal(ra["py_used"], rb["py_used"])
    for b in range(B):
        assert np.array_equal(da[b][1], db[b][1]) and np.array_equal(da[b][0], db[b][0]) and sa[b] == sb[b]
        n_it = int(ra["iters_done"][b])
        assert np.array_equal(ra["cost_trace"][b, :n_it], rb["cost_trace"][b, :n_it])
    assert np.array_equal(ra["stats"][:, 13], rb["stats"][:, 13])


def test_suite_against_the_small_limits_build():
    """libnirrt_hip_small.so = same sources with -DCHAIN_MAX=8 -DNEAR_STASH=8: parent chains longer than 8 edges take the
    global-walk branches of wg_recost_subtree / the rewire leaf path, all but 8 members of a Near set live in the HBM
    continuation of the stash, and rewire candidates beyond the 8 list slots are searched there every round.
    The fixture and oracle comparisons must not notice.  Run in a child interpreter because the library path is read
    once per process."""
    from nirrt_star_amd import build
    assert os.path.exists(build.SO_SMALL), "python -m nirrt_star_amd.build makes it"
    env = dict(os.environ, NIRRT_HIP_SO=build.SO_SMALL, NIRRT_FORCE_VARIANT="slim")
    for variant in ("slim", "wide"):
        env["NIRRT_FORCE_VARIANT"] = variant
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                            os.path.join(ROOT, "tests", "test_hip_parity.py"), os.path.join(ROOT, "tests", "test_hip_sampling.py"),
                            os.path.join(ROOT, "tests", "test_hip_grid.py") + "::test_default_index_against_the_oracle_at_mid_size"],
                           env=env, cwd=ROOT, capture_output=True, text=True)
        assert r.returncode == 0, "%s kernels, small-limits build:\n%s\n%s" % (variant, r.stdout[-3000:], r.stderr[-2000:])
