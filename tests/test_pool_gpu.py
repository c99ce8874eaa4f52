"""GPU: the arena pool: trees of varying sizes created and destroyed in a loop do not grow device memory, a freed
block serves a smaller tree, and an emptied chunk starts over."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_trees_of_varying_sizes_do_not_grow_device_memory(monkeypatch):
    import torch
    from nirrt_star_amd import _hip, worlds
    monkeypatch.setenv("NIRRT_POOL_CHUNK_MB", "64")
    _hip.pool_trim()
    pr = worlds.problem_2d(worlds.random_world_2d(0, "b30"), 0)
    # (the runtime's own one-time allocations - code objects, the kernels' scratch ring, streams - happen before the baseline is taken)
    t = _hip.HipTree(2, 20000, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], 3, pr["env"])
    _hip.run_replay([t], np.random.default_rng(99).uniform(3, 221, size=(1, 300, 2)))
    t.close()
    _hip.pool_trim()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    low = free0
    for rep in range(12):
        trees = [_hip.HipTree(2, 20000 + 1500 * ((rep * 7 + k) % 9), pr["x_start"], pr["x_goal"], 10, pr["search_radius"], 3, pr["env"]) for k in range(6)]
        # the trees work (fresh arenas or reused blocks alike)
        rng = np.random.default_rng(rep)
        r = _hip.run_replay(trees[:2], rng.uniform(3, 221, size=(2, 300, 2)))
        assert (r["iters_done"] == 300).all()
        low = min(low, torch.cuda.mem_get_info()[0])
        for t in trees:
            t.close()
    _hip.pool_trim()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - low < 400 << 20          # never more than a handful of 64 MB chunks live at once (6 trees of <= 9 MB)
    assert abs(free0 - free1) < 200 << 20   # everything returned (up to what the runtime keeps for itself)
