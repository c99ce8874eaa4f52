"""Oracle geometry vs known answers produced by the reference's collision_check_utils{,_3d}.py."""
import numpy as np
import pytest

from conftest import load_golden


@pytest.mark.parametrize("name,dim", [("geom2d", 2), ("geom3d", 3)])
def test_geometry_known_answers(oracle, name, dim):
    g = load_golden(name)
    clr = float(g["clearance"])
    for wi in range(int(g["n_worlds"])):
        ed = g["w%d_env" % wi]
        t = oracle.OracleTree(dim, 10, ed["start"][0], ed["goal"][0], 10.0, 100.0, clr, ed)
        a, b = g["w%d_seg_a" % wi], g["w%d_seg_b" % wi]
        col = np.array([t.is_collision(a[i], b[i]) for i in range(len(a))], dtype=np.uint8)
        assert np.array_equal(col, g["w%d_collision" % wi]), "world %d" % wi
        pts = g["w%d_pts" % wi]
        inside = np.array([t.is_inside_obs(p) for p in pts], dtype=np.uint8)
        valid = np.array([t.is_valid(p) for p in pts], dtype=np.uint8)
        assert np.array_equal(inside, g["w%d_inside" % wi])
        assert np.array_equal(valid, g["w%d_valid" % wi])
        if dim == 2:
            inr = np.array([t.is_in_range(p) for p in pts], dtype=np.uint8)
            assert np.array_equal(inr, g["w%d_inrange" % wi])
        # fixture sanity: both outcomes are represented
        assert 0 < col.mean() < 1 and 0 < inside.mean() < 1
        t.close()
