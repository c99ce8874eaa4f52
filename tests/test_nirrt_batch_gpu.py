"""GPU: NIRRT* as composed in BASELINE.json configs 3 and 4, and NIRRT* as a batched workload.

  * configs 3 / 4 (`nirrt_star -n pointnet2 -c bfs random_2d`, `nirrt_star -n pointnet2 random_3d`): the planner classes with
    the REAL PNGWrapper / PNGWrapper3D on 'cuda' + the HIP tree.  Fixtures config3_nirrtc2d_real / config4_nirrt3d_real hold
    runs of the reference's own planner + wrapper + PointNet++ (CPU) with every network call recorded.  L3: the wrapper
    here runs its forward on the GPU but hands the planner the fixture's path_pred - clouds and masks of every call must
    equal the reference's, and the tree must be the reference's tree.  L4: the scores of those very forwards vs the
    reference model's (<= 2e-3 incl. the fixture's float16 storage; labels agree on >= 99 % of the points).
    Reference: demo_planning_2d.py:53-59,75-90, nirrt_star_png_2d.py:56-174, nirrt_star_png_c_2d.py:52-87, 3D twins.
  * batched NIRRT*[-C] / NRRT* through nirrt_star_amd.batch (what eval_sharded --planner nirrt_star and bench.py --algo
    nirrt run): with a deterministic wrapper every tree of the batch equals the planner class run alone on the same
    seeds; with the real wrapper the due clouds are classified in ONE forward.
"""
import random
from types import SimpleNamespace as NS

import numpy as np
import pytest

from conftest import FakePNG, load_golden, synthetic_checkpoint_root

pytestmark = pytest.mark.gpu


def _planner(g, wrapper, mode="resident"):
    from nirrt_star_amd import planners
    dim = int(g["dim"])
    connect = str(g["algo"]) == "nirrt_c"
    common = [tuple(g["x_start"]), tuple(g["x_goal"]), 10, float(g["search_radius"]), int(g["iter_max"]), g["env"], wrapper]
    tail = [int(g["clearance"]), 2048, 5, 0.5, 0.9]
    if dim == 2:
        common.append(g["binary_mask"].astype(np.float64))
        cls = planners.NIRRTStarPNGC2D if connect else planners.NIRRTStarPNG2D
    else:
        cls = planners.NIRRTStarPNGC3D if connect else planners.NIRRTStarPNG3D
    return cls(*common, *tail, 5, mode=mode) if connect else cls(*common, *tail, mode=mode)


@pytest.mark.parametrize("name", ["config3_nirrtc2d_real", "config4_nirrt3d_real"])
def test_configs_3_and_4_real_wrapper_on_cuda(name):
    import torch
    from nirrt_star_amd import png_wrapper
    g = load_golden(name)
    dim = int(g["dim"])
    base = png_wrapper.PNGWrapper if dim == 2 else png_wrapper.PNGWrapper3D
    seen = []

    class Injecting(base):
        """real forward on the GPU (L4), the reference's labels to the planner (L3)"""

        def classify_batch(self, clouds, start_masks, goal_masks, fps_starts=None):
            pred, score = base.classify_batch(self, clouds, start_masks, goal_masks, fps_starts)
            out = []
            for c, sm, gm, p_, s_ in zip(clouds, start_masks, goal_masks, pred, score):
                i = len(seen)
                assert i < int(g["n_calls"]), "more network calls than the reference made"
                assert np.array_equal(np.asarray(c, dtype=np.float32), g["call%d_pc" % i]), "cloud of call %d" % i
                assert np.array_equal(np.asarray(sm) > 0, g["call%d_start" % i] > 0) and np.array_equal(np.asarray(gm) > 0, g["call%d_goal" % i] > 0)
                seen.append((p_, s_))
                out.append(g["call%d_pred" % i].astype(np.int64))
            return np.stack(out), score

    w = Injecting(root_dir=synthetic_checkpoint_root(dim), device="cuda")
    assert next(w.model.parameters()).is_cuda
    p = _planner(g, w)
    seed = int(g["seed"])
    np.random.seed(seed)
    random.seed(seed)
    torch.manual_seed(seed)
    p.planning()
    assert len(seen) == int(g["n_calls"])
    n = p.num_vertices
    assert n == int(g["n"]) and np.array_equal(p.vertex_parents[:n], g["parents"])
    assert np.array_equal(p.vertices[:n], g["vertices"])
    assert np.array_equal(np.array(p.path_solutions), g["path_solutions"])
    assert abs(p.get_path_len(p.path) - float(g["path_len"])) <= 1e-5
    # L4: this repo's forward (folded conv+BN GEMMs, HIP FPS / ball query / 3-NN) on the reference's inputs
    for i, (pred, score) in enumerate(seen):
        ref_score = g["call%d_score" % i].astype(np.float32)
        assert np.max(np.abs(score - ref_score)) <= 2e-3, "call %d" % i
        assert np.mean(pred == g["call%d_pred" % i]) >= 0.99, "call %d" % i


def _args(planner, dim, iter_max, after):
    return NS(problem="random_2d" if dim == 2 else "random_3d", planner=planner, iter_max=iter_max, iter_after_initial=after, step_len=10,
              clearance=3 if dim == 2 else 2, pc_n_points=2048, pc_over_sample_scale=5, pc_sample_rate=0.5, pc_update_cost_ratio=0.9,
              connect_max_trial_attempts=5, root_dir=".", segment=1000)


class DiagonalFake(FakePNG):
    """deterministic wrapper whose labels do not depend on the problem (a point is "path" iff it lies near the world's main
    diagonal), with the batched entry point of PNGWrapper on top of FakePNG's single-cloud one"""

    def __init__(self, dim):
        hi = 224.0 if dim == 2 else 50.0
        FakePNG.__init__(self, np.zeros(dim), np.full(dim, hi), 40.0 if dim == 2 else 12.0)
        self.forwards = 0

    def classify_batch(self, clouds, start_masks, goal_masks, fps_starts=None):
        self.forwards += 1
        res = [self.classify_path_points(c, s, g_) for c, s, g_ in zip(clouds, start_masks, goal_masks)]
        return np.stack([r[0] for r in res]), np.stack([r[1] for r in res])


@pytest.mark.parametrize("planner,dim", [("nirrt_star", 2), ("nirrt_star", 3), ("nrrt_star", 2)])
def test_batched_guided_planners_equal_the_planner_class(planner, dim):
    """planning_random through eval_sharded.plan_batch (3 problems per persistent launch, batched cloud refresh) ==
    planning_random of the planner class run alone with np.random.seed / random.seed (1000 + problem id)"""
    from nirrt_star_amd import eval_sharded as es, planners, worlds
    if dim == 2:
        probs = [worlds.problem_2d(worlds.random_world_2d(20 + i, "b30"), 0) for i in range(3)]
    else:
        probs = []
        for i in range(3):
            np.random.seed(30 + i)
            probs.append(worlds.problem_3d(worlds.random_world_3d(30 + i)))
    pids = [11, 12, 13]
    args = _args(planner, dim, 3000, 200)
    w = DiagonalFake(dim)
    recs, traces = es.plan_batch(probs, pids, args, 0, wrapper=w)
    assert w.forwards >= 1
    for pr, pid, rec, tr in zip(probs, pids, recs, traces):
        fake = DiagonalFake(dim)
        clr = 3 if dim == 2 else 2
        if planner == "nirrt_star":
            if dim == 2:
                p = planners.NIRRTStarPNG2D(pr["x_start"], pr["x_goal"], 10, pr["search_radius"], 3200, pr["env_dict"], fake, pr["binary_mask"], clr, 2048, 5, 0.5, 0.9)
            else:
                p = planners.NIRRTStarPNG3D(pr["x_start"], pr["x_goal"], 10, pr["search_radius"], 3200, pr["env_dict"], fake, clr, 2048, 5, 0.5, 0.9)
        else:
            p = planners.NRRTStarPNG2D(pr["x_start"], pr["x_goal"], 10, pr["search_radius"], 3200, pr["env_dict"], fake, pr["binary_mask"], clr, 2048, 5, 0.5)
        np.random.seed(1000 + pid)
        random.seed(1000 + pid)
        lst = np.array(p.planning_random(200))
        tr = np.asarray(tr)
        assert len(tr) == len(lst) and np.array_equal(np.isinf(tr), np.isinf(lst)), "problem %d" % pid
        fin = np.isfinite(lst)
        assert fin.any() and np.max(np.abs(tr[fin] - lst[fin])) <= 1e-9
        assert rec[2] == p.num_vertices


def test_batched_real_wrapper_classifies_all_due_clouds_in_one_forward():
    """nirrt_star -n pointnet2 over a batch with the real PNGWrapper on cuda: init_pc of 6 problems = ONE forward with B = 6;
    every problem gets solved and its best cost never rises"""
    from nirrt_star_amd import eval_sharded as es, png_wrapper, worlds
    probs = [worlds.problem_2d(worlds.random_world_2d(40 + i, "b30"), 0) for i in range(6)]
    w = png_wrapper.PNGWrapper(root_dir=synthetic_checkpoint_root(2), device="cuda")
    sizes = []
    inner = w.classify_batch

    def counting(clouds, sm, gm, fps_starts=None):
        sizes.append(len(clouds))
        return inner(clouds, sm, gm, fps_starts)

    w.classify_batch = counting
    inner_dev = w.classify_device

    def counting_dev(x, fps_starts=None):       # the batched refresh hands over input blocks assembled on the device
        sizes.append(x.shape[0])
        return inner_dev(x, fps_starts)

    w.classify_device = counting_dev
    recs, traces = es.plan_batch(probs, list(range(6)), _args("nirrt_star", 2, 4000, 300), 0, wrapper=w)
    assert sizes[0] == 6 and max(sizes) <= 6
    for rec, tr in zip(recs, traces):
        fin = np.isfinite(tr)
        assert rec[1] > 0 and fin.any() and len(tr) == rec[1] + 300
        assert np.all(np.diff(np.asarray(tr)[fin]) <= 1e-12)
    # B = 1 and B = 6 forwards of the same cloud agree (L4 tolerance): batch composition does not change a label beyond ties
    g = load_golden("config3_nirrtc2d_real")
    pc, sm, gm = g["call0_pc"], g["call0_start"].astype(np.float32), g["call0_goal"].astype(np.float32)
    import torch
    st = [torch.tensor([7] * 6), torch.tensor([5] * 6), torch.tensor([3] * 6), torch.tensor([1] * 6)]
    p6, s6 = inner([pc] * 6, [sm] * 6, [gm] * 6, fps_starts=st)
    p1, s1 = inner([pc], [sm], [gm], fps_starts=[t[:1] for t in st])
    assert np.max(np.abs(s6 - s1[0])) <= 1e-3 and np.mean(p6 == p1[0]) >= 0.99


@pytest.mark.gpu
@pytest.mark.parametrize("dim", [2, 3])
def test_device_assembled_network_input_changes_no_result(dim, monkeypatch):
    """the batched refresh with input blocks assembled on the device (k_net_input, predictions handed to the trees without a
    host round trip) against the same run with the numpy assembly (NIRRT_HOST_INPUT=1): identical cost traces and clouds"""
    from nirrt_star_amd import eval_sharded as es, png_wrapper, worlds
    mk = (lambda i: worlds.problem_2d(worlds.random_world_2d(60 + i, "b30"), 0)) if dim == 2 else (lambda i: worlds.problem_3d(worlds.random_world_3d(60 + i)))
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("NIRRT_HOST_INPUT", mode)
        if dim == 3:
            np.random.seed(5)
        probs = [mk(i) for i in range(5)]
        w = (png_wrapper.PNGWrapper if dim == 2 else png_wrapper.PNGWrapper3D)(root_dir=synthetic_checkpoint_root(dim), device="cuda")
        calls = []
        inner = w.classify_device
        w.classify_device = lambda x, fps_starts=None: (calls.append(x.shape[0]), inner(x, fps_starts))[1]
        out[mode] = (es.plan_batch(probs, list(range(5)), _args("nirrt_star", dim, 3000, 400), 0, wrapper=w), len(calls))
    (r0, t0), n0 = out["0"]
    (r1, t1), n1 = out["1"]
    assert n0 > 0 and n1 == 0
    for a, b in zip(t0, t1):
        assert np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)
    assert [r[1] for r in r0] == [r[1] for r in r1]


@pytest.mark.gpu
def test_launch_groups_change_no_result(monkeypatch):
    """a guided batch split into groups that are launched from worker threads (NIRRT_BATCH_GROUPS / _INFLIGHT, two launches on the
    device at once) against the default single group: trees are independent, so identical traces"""
    from nirrt_star_amd import eval_sharded as es, png_wrapper, worlds
    out = {}
    for mode, env in (("one", {}), ("split", {"NIRRT_BATCH_GROUPS": "3", "NIRRT_BATCH_INFLIGHT": "2", "NIRRT_BATCH_OVERLAP_MIN": "1",
                                               "NIRRT_BATCH_WINDOW": "700"})):
        for k in ("NIRRT_BATCH_GROUPS", "NIRRT_BATCH_INFLIGHT", "NIRRT_BATCH_OVERLAP_MIN", "NIRRT_BATCH_WINDOW"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        probs = [worlds.problem_2d(worlds.random_world_2d(80 + i, "b30"), 0) for i in range(7)]
        # (labels that do not depend on the batch: a real forward's GEMMs sum in a different order for a different batch size)
        out[mode] = es.plan_batch(probs, list(range(7)), _args("nirrt_star", 2, 3000, 400), 0, wrapper=DiagonalFake(2))
    for a, b in zip(out["one"][1], out["split"][1]):
        assert np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize("ratio", [0.9, 1.0])
def test_parked_launches_change_no_result(monkeypatch, ratio):
    """a guided batch whose launches end as soon as a share of the trees waits for a cloud refresh (nirrt_run_args.park_limit:
    the others come back with NIRRT_E_PARK, their early draw taken back, and resume) against launches that always run their
    window: identical traces - also at the 3D demo's pc_update_cost_ratio = 1.0 (demo_planning_3d.py:21), where every improvement
    of the best cost parks a tree"""
    from nirrt_star_amd import eval_sharded as es, worlds

    class EveryOtherPoint(DiagonalFake):
        """labels that depend on nothing but the point's place in the cloud - never an empty prediction (an empty one ends a
        planner like it ends the reference: np.random.randint(0, 0)), the same for every batch composition"""

        def classify_path_points(self, pc, start_mask, goal_mask):
            pred = (np.arange(len(pc)) % 2 == 0).astype(np.int64)
            return pred, pred.astype(np.float32)

    out = {}
    for mode, env in (("off", {"NIRRT_BATCH_PARK": "0"}), ("on", {"NIRRT_BATCH_PARK": "0.3", "NIRRT_BATCH_PARK_MIN": "1"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        probs = [worlds.problem_2d(worlds.random_world_2d(80 + i, "b30"), 0) for i in range(9)]
        a = _args("nirrt_star", 2, 3000, 600)
        a.pc_update_cost_ratio = ratio
        out[mode] = es.plan_batch(probs, list(range(9)), a, 0, wrapper=EveryOtherPoint(2))
        monkeypatch.delenv("NIRRT_BATCH_PARK_MIN", raising=False)
    for a, b in zip(out["off"][1], out["on"][1]):
        assert np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)
    assert [r[1] for r in out["off"][0]] == [r[1] for r in out["on"][0]]
