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


def get_random_2d_problem_input(random_2d_env_config):
    return worlds.problem_2d(random_2d_env_config['env_dict'], 0)


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
