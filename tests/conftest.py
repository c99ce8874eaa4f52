import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "oracle_pointops: CPU test that runs the network / down-sampling through the oracle's point operators")


@pytest.fixture(autouse=True)
def _oracle_pointops_where_asked(request):
    """The package has only the HIP point operators and no hook for others.  A CPU test that needs a forward or a down-sampling
    on a host WITHOUT a GPU says so (`@pytest.mark.oracle_pointops`) and gets the oracle's torch / numpy operators patched into
    the module for ITS duration only; every other test - and every test on a GPU box - sees the product as it is, so a call
    that should not reach the point operators fails loudly instead of being served by the oracle."""
    if request.node.get_closest_marker("oracle_pointops") is None or _gpu_visible():
        yield
        return
    from nirrt_star_amd import pointops
    from oracle import pointops_ref
    with pointops_ref.patched(pointops):
        yield


def _gpu_visible():
    import torch
    return torch.cuda.is_available()


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out = {k: z[k] for k in z.files}
    for k in list(out):
        if k == "env" or k.endswith("_env") or (k.startswith("env") and out[k].ndim == 0):
            out[k] = json.loads(str(out[k]))
    return out


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


def make_oracle_tree(orc, g, iter_max=None):
    """OracleTree for a run_* / random_* fixture dict."""
    return orc.OracleTree(int(g["dim"]), int(iter_max if iter_max is not None else g["iter_max"]),
                          g["x_start"], g["x_goal"], float(g["step_len"]), float(g["search_radius"]),
                          float(g["clearance"]), g["env"])


class FakePNG:
    """Deterministic stand-in for the PointNet++ wrapper (same interface): a cloud point is a "path point"
    iff it lies within `width` of the start-goal segment.  Used to pin the NIRRT* control flow (cloud
    refresh rule, sampling mix, RNG consumption) against the reference independently of network weights."""

    def __init__(self, x_start, x_goal, width=25.0):
        self.a = np.asarray(x_start, dtype=np.float64)
        self.b = np.asarray(x_goal, dtype=np.float64)
        self.width = width
        self.calls = 0

    def classify_path_points(self, pc, start_mask, goal_mask):
        self.calls += 1
        p = np.asarray(pc, dtype=np.float64)
        ab = self.b - self.a
        t = np.clip(((p - self.a) @ ab) / (ab @ ab), 0, 1)
        d = np.linalg.norm(p - (self.a + t[:, None] * ab), axis=1)
        pred = (d < self.width).astype(np.int64)
        return pred, (1.0 / (1.0 + d)).astype(np.float32)

    def generate_connected_path_points(self, pc, x_start, x_goal, env_dict, neighbor_radius, max_trial_attempts,
                                       visualize=False, vis_folderpath="", token=""):
        pred, _ = self.classify_path_points(pc, None, None)
        return True, 1, pred.astype(np.float32)


def seed_grow_classifier(grow, calls=None):
    """the deterministic classifier of the neural-connect fixtures (tests/golden/make_golden.py, same function there): a point
    is "path" iff it lies within `grow` of a point of the start or goal mask - several rounds are needed to connect"""
    def classify(pc_, start_mask, goal_mask):
        pc_ = np.asarray(pc_)
        seeds = pc_[(np.asarray(start_mask) + np.asarray(goal_mask)) > 0]
        d = np.linalg.norm(pc_[:, None] - seeds[None], axis=2).min(axis=1) if len(seeds) else np.full(len(pc_), np.inf)
        if calls is not None:
            calls.append((np.asarray(start_mask, dtype=np.float32).copy(), np.asarray(goal_mask, dtype=np.float32).copy()))
        return (d < grow).astype(np.int64), (1.0 / (1.0 + d)).astype(np.float32)
    return classify


def stub_connect_wrapper(dim, grow, calls=None):
    """PNGWrapper / PNGWrapper3D without a network: generate_connected_path_points is the package's, the classifier the seed-growing one"""
    from nirrt_star_amd import png_wrapper
    w = object.__new__(png_wrapper.PNGWrapper if dim == 2 else png_wrapper.PNGWrapper3D)
    classify = seed_grow_classifier(grow, calls)
    w.classify_path_points = classify
    w.classify_batch = lambda clouds, sm, gm, fps_starts=None: (np.stack([classify(c, s, g_)[0] for c, s, g_ in zip(clouds, sm, gm)]), None)
    return w


_CK_ROOTS = {}


def synthetic_checkpoint_root(dim):
    """root_dir with the seeded synthetic PointNet++ checkpoint of `dim` in the reference's layout, written the way
    tests/golden/make_golden.py wrote the one the config3 / config4 fixtures were generated with (CPU calibration
    forwards, oracle point operators)"""
    import tempfile
    from nirrt_star_amd import png_wrapper
    if dim not in _CK_ROOTS:
        from nirrt_star_amd import pointops
        from oracle import pointops_ref
        root = tempfile.mkdtemp(prefix="nirrt_ck_")
        with pointops_ref.patched(pointops):   # the fixtures' weights were calibrated by CPU forwards: same here, explicitly
            png_wrapper.make_synthetic_checkpoint(png_wrapper.checkpoint_path(root, dim), seed=0, dim=dim, device="cpu")
        _CK_ROOTS[dim] = root
    return _CK_ROOTS[dim]
