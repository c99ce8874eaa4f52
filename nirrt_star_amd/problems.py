"""Problem loaders with the reference's names (datasets/planning_problem_utils_2d.py:30-47,145-172;
datasets_3d/planning_problem_utils_3d.py:47-97).

If the reference's dataset is present (data/random_{2d,3d}/test/envs.json, same JSON schema) it is
used - masks are rasterised analytically from the obstacle lists instead of read from the PNGs with
cv2 (un-vendored; pixel parity unpinned).  Otherwise the synthetic evaluation set of
SURVEY.md §8d is generated: 250 worlds x 4 start/goal pairs (2D) / 1000 worlds (3D).
"""
import json
import os
from copy import copy
from os.path import join

from . import worlds
from .env import Env3D

SYNTHETIC_WORLD_KIND_2D = os.environ.get("NIRRT_WORLD_2D", "ref2d")
N_SYNTHETIC_WORLDS_2D = 250
N_SYNTHETIC_WORLDS_3D = 1000


def get_random_2d_env_configs(root_dir='.'):
    path = join(root_dir, "data", "random_2d", "test", "envs.json")
    if os.path.exists(path):
        with open(path) as f:
            maps = json.load(f)
    else:
        maps = [worlds.random_world_2d(i, SYNTHETIC_WORLD_KIND_2D) for i in range(N_SYNTHETIC_WORLDS_2D)]
    out = []
    for map_idx, ed in enumerate(maps):
        for pair in range(len(ed['start'])):
            cfg = {'img_idx': map_idx, 'start_goal_idx': pair, 'env_dict': copy(ed)}
            cfg['env_dict']['start'] = [ed['start'][pair]]
            cfg['env_dict']['goal'] = [ed['goal'][pair]]
            out.append(cfg)
    return out


def read_env_image_mask(path):
    """get_binary_mask(cv2.imread(path)) (datasets/point_cloud_mask_utils.py:8-17) without cv2: 1.0 where the first
    colour channel of the PNG is non-zero.  cv2.imread returns BGR and the dataset images are black / white, so which
    channel is read makes no difference for them; PIL (a matplotlib dependency) decodes the file."""
    import numpy as np
    from PIL import Image
    img = np.asarray(Image.open(path).convert("RGB"))
    return (img[:, :, 2] != 0).astype(np.float64)          # channel 0 of cv2's BGR = blue


def get_random_2d_problem_input(random_2d_env_config, root_dir='.'):
    """planning_problem_utils_2d.py:145-162.  With the dataset's env_imgs/<img_idx>.png present the free-space mask (and
    with it gamma) comes from the image like in the reference; otherwise it is rasterised from the obstacle lists."""
    problem = worlds.problem_2d(random_2d_env_config['env_dict'], 0)
    png = join(root_dir, "data", "random_2d", "test", "env_imgs", "%s.png" % random_2d_env_config.get('img_idx'))
    if os.path.exists(png):
        problem['binary_mask'] = read_env_image_mask(png)
        problem['search_radius'] = compute_gamma_rrt_star(problem['binary_mask'], dim=2)
    return problem


# ------------------------------------------------------------------------------------------------
# block / gap problems (datasets/planning_problem_utils_2d.py:14-27, 49-142; generate_block_gap_env_2d.py)
# ------------------------------------------------------------------------------------------------
BLOCK_GAP_SYNTHETIC_SEED = 20241008


def generate_block_gap_configs(seed=None):
    """The reference's generator script (generate_block_gap_env_2d.py:8-48) as a function: 5 x 100 centre-block worlds
    (map side = 2..6 x d_goal) and 5 x 100 wall-with-gap worlds (gap height 7..3), block widths / gap positions from
    the global numpy generator in the script's order.  seed=None keeps the caller's generator state (the script itself
    is unseeded)."""
    import numpy as np
    if seed is not None:
        np.random.seed(seed)
    cfg = {'block': [], 'gap': []}
    num_envs, d_goal = 100, 60
    block_widths = np.random.randint(10, 50, num_envs)
    for ratio in (2, 3, 4, 5, 6):
        side = d_goal * ratio
        for bw in block_widths:
            rest = d_goal - bw
            best = bw + ((rest // 2) ** 2 + (bw // 2) ** 2) ** 0.5 + ((rest - rest // 2) ** 2 + (bw // 2) ** 2) ** 0.5
            cfg['block'].append({'w': int(bw), 'd_goal': d_goal, 'img_height': side, 'img_width': side,
                                 'best_path_len': float(best)})
    h, t, d_goal = 90, 20, 60
    flank = t + 2 * (((d_goal - t) / 2) ** 2 + (h / 2) ** 2) ** 0.5
    for h_g in (7, 6, 5, 4, 3):
        for y_g in np.random.randint(20, 70, num_envs):
            cfg['gap'].append({'h': h, 't': t, 'h_g': h_g, 'y_g': int(y_g), 'd_goal': d_goal, 'img_height': 224,
                               'img_width': 224, 'flank_path_len': flank})
    return cfg


def _block_gap_configs(root_dir):
    path = join(root_dir, "data", "block_gap", "block_gap_configs.json")
    if os.path.exists(path):
        with open(path) as f:
            return json.load(f)
    import numpy as np
    state = np.random.get_state()
    try:
        return generate_block_gap_configs(BLOCK_GAP_SYNTHETIC_SEED)   # offline: a fixed synthetic set, caller's RNG untouched
    finally:
        np.random.set_state(state)


def get_block_env_configs(root_dir='.'):
    return _block_gap_configs(root_dir)['block']


def get_gap_env_configs(root_dir='.'):
    return _block_gap_configs(root_dir)['gap']


def _rect_problem(img_height, img_width, rectangles, d_goal, extra):
    from .env import Env
    x_start = (img_width // 2 - d_goal // 2, img_height // 2)
    x_goal = (img_width // 2 + d_goal // 2, img_height // 2)
    env_dict = {'env_dims': (img_height, img_width), 'rectangle_obstacles': rectangles, 'circle_obstacles': [],
                'start': [x_start], 'goal': [x_goal]}
    mask = worlds.rasterize_mask_2d(env_dict['env_dims'], rectangles, [])   # cv2.rectangle(-1): both corners inclusive
    problem = {'x_start': x_start, 'x_goal': x_goal, 'env_dict': env_dict, 'env': Env(env_dict), 'binary_mask': mask}
    problem.update(extra)
    problem['search_radius'] = compute_gamma_rrt_star(mask, dim=2)
    return problem


def get_block_problem_input(block_env_config):
    """One square block of side w centred in the map, start / goal d_goal apart on the horizontal centre line."""
    c = block_env_config
    w, H, W = c['w'], c['img_height'], c['img_width']
    rect = [W // 2 - w // 2, H // 2 - w // 2, w, w]
    return _rect_problem(H, W, [rect], c['d_goal'], {'best_path_len': c['best_path_len']})


def get_gap_problem_input(gap_env_config):
    """A wall of thickness t and height h centred in the map with a gap of height h_g whose lower edge is y_g above the
    wall's bottom: two rectangles (above / below the gap)."""
    c = gap_env_config
    h, t, h_g, y_g, H, W = c['h'], c['t'], c['h_g'], c['y_g'], c['img_height'], c['img_width']
    x0, y0 = W // 2 - t // 2, H // 2 - h // 2
    rects = [[x0, y0, t, h - h_g - y_g], [x0, y0 + (h - y_g), t, y_g]]
    return _rect_problem(H, W, rects, c['d_goal'], {'flank_path_len': c['flank_path_len']})


def compute_gamma_rrt_star(binary_mask, dim=2):
    return worlds.gamma_rrt_star(binary_mask.sum(), dim)


def get_random_3d_env_configs(root_dir='.'):
    path = join(root_dir, "data", "random_3d", "test", "envs.json")
    if os.path.exists(path):
        with open(path) as f:
            maps = json.load(f)
    else:
        maps = [worlds.random_world_3d(i) for i in range(N_SYNTHETIC_WORLDS_3D)]
    return [{'img_idx': i, 'env_dict': copy(ed)} for i, ed in enumerate(maps)]


def get_random_3d_problem_input(random_3d_env_config):
    return worlds.problem_3d(random_3d_env_config['env_dict'])


def compute_gamma_rrt_star_3d(env):
    return worlds.gamma_rrt_star(worlds.approximate_free_vol_3d(env), 3)
