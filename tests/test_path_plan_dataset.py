"""CPU: the training-set schema and Dataset vs the reference's PathPlanDataset on the same synthetic .npz
(tests/golden/dataset_ref.npz, written by make_golden.py from /root/reference)."""
import numpy as np

from conftest import load_golden


def test_dataset_items_match_the_reference(tmp_path):
    from nirrt_star_amd.path_plan_dataset import PathPlanDataset
    g = load_golden("dataset_ref")
    path = str(tmp_path / "test.npz")
    np.savez(path, **{k[3:]: g[k] for k in g if k.startswith("in_")})
    ds = PathPlanDataset(path)
    assert len(ds) == int(g["length"])
    assert np.array_equal(ds.labelweights, g["labelweights"])
    for i in (0, 3):
        raw, xyz, feat, lab, tok = ds[i]
        assert np.array_equal(raw, g["item%d_raw" % i]) and raw.dtype == np.float32 and raw.shape[1] == 3
        assert np.array_equal(xyz, g["item%d_xyz" % i]) and np.array_equal(feat, g["item%d_feat" % i])
        assert np.array_equal(lab, g["item%d_lab" % i]) and str(tok) == str(g["item%d_tok" % i])


def test_sample_schema_round_trip(tmp_path):
    from nirrt_star_amd import path_plan_dataset as ppd
    rng = np.random.default_rng(2)
    samples = []
    for i in range(3):
        pc = rng.uniform(0, 224, size=(128, 2))
        path = np.stack([np.linspace(20, 200, 30), np.linspace(30, 190, 30)], axis=1)
        samples.append(ppd.make_sample(pc, path[0], path[-1], path, 20, 20, 20, "train-%d_0" % i))
    s = samples[0]
    assert set(s) == set(ppd.KEYS) and s["pc"].dtype == np.float32 and s["astar"].shape == (128,)
    assert np.array_equal(s["free"], (1 - s["start"]) * (1 - s["goal"]))
    assert s["start"].sum() > 0 and s["astar"].sum() >= s["start"].sum()      # the path passes through the start disc
    out = str(tmp_path / "train.npz")
    ppd.save_dataset(out, samples)
    ds = ppd.PathPlanDataset(out)
    assert len(ds) == 3 and ds[1][2].shape == (128, 3) and str(ds[2][4]) == "train-2_0"
