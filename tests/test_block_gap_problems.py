"""CPU: block / gap evaluation problems vs the reference's own generator script and loader
(tests/golden/block_gap_seed7.json, written by make_golden.py from /root/reference)."""
import json
import os

import numpy as np

from nirrt_star_amd import problems

HERE = os.path.dirname(os.path.abspath(__file__))


def _golden():
    with open(os.path.join(HERE, "golden", "block_gap_seed7.json")) as f:
        return json.load(f)


def test_generator_reproduces_the_reference_script():
    g = _golden()
    state = np.random.get_state()
    cfg = problems.generate_block_gap_configs(seed=g["seed"])
    np.random.set_state(state)
    assert json.loads(json.dumps(cfg)) == g["configs"]          # same keys, ints and float64 values after the JSON trip


def test_problem_inputs_match_the_reference_loader():
    g = _golden()
    for key, exp in g["problems"].items():
        kind, i = key.split("_")
        c = g["configs"][kind][int(i)]
        pr = problems.get_block_problem_input(c) if kind == "block" else problems.get_gap_problem_input(c)
        assert list(pr["x_start"]) == exp["x_start"] and list(pr["x_goal"]) == exp["x_goal"]
        assert list(pr["env_dict"]["env_dims"]) == exp["env_dims"]
        assert [list(r) for r in pr["env_dict"]["rectangle_obstacles"]] == exp["rectangle_obstacles"]
        assert pr["env_dict"]["circle_obstacles"] == []
        assert float(pr["binary_mask"].sum()) == exp["free_pixels"]
        assert float(pr["search_radius"]) == exp["search_radius"]
        assert float(pr["best_path_len" if kind == "block" else "flank_path_len"]) == exp["threshold"]
        # the start / goal of every such problem are collision-free and the env object carries the rectangles
        assert len(pr["env"].obs_rectangle) == len(exp["rectangle_obstacles"])


def test_offline_config_set_is_fixed_and_leaves_the_generator_alone(tmp_path):
    np.random.seed(5)
    before = np.random.get_state()[1].copy()
    a = problems.get_block_env_configs(str(tmp_path))
    b = problems.get_gap_env_configs(str(tmp_path))
    assert np.array_equal(before, np.random.get_state()[1])
    assert len(a) == 500 and len(b) == 500
    assert a == problems.get_block_env_configs(str(tmp_path))


def test_random_2d_loader_uses_dataset_images_when_present(tmp_path):
    """envs.json + env_imgs/<i>.png as the reference's dataset lays them out: mask and gamma come from the image"""
    from PIL import Image
    from nirrt_star_amd import worlds
    ed = worlds.random_world_2d(3, "ref2d")
    d = tmp_path / "data" / "random_2d" / "test"
    (d / "env_imgs").mkdir(parents=True)
    with open(d / "envs.json", "w") as f:
        json.dump([ed], f)
    mask = worlds.rasterize_mask_2d(ed["env_dims"], ed["rectangle_obstacles"], ed["circle_obstacles"])
    mask[:5, :] = 0.0                                          # the image differs from the analytic raster on purpose
    img = np.repeat((mask * 255).astype(np.uint8)[:, :, None], 3, axis=2)
    Image.fromarray(img).save(d / "env_imgs" / "0.png")
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        cfgs = problems.get_random_2d_env_configs()
        assert len(cfgs) == len(ed["start"]) and cfgs[1]["start_goal_idx"] == 1 and cfgs[1]["img_idx"] == 0
        pr = problems.get_random_2d_problem_input(cfgs[1])
    finally:
        os.chdir(cwd)
    assert np.array_equal(pr["binary_mask"], mask)
    assert pr["search_radius"] == problems.compute_gamma_rrt_star(mask)
    assert pr["x_start"] == tuple(ed["start"][1]) and pr["x_goal"] == tuple(ed["goal"][1])
