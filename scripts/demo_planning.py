#!/usr/bin/env python3
"""Counterpart of the reference's demo_planning_2d.py / demo_planning_3d.py (same flags), running on
the MI355X library.  Problems are synthetic unless data/random_{2d,3d}/test/envs.json exists.

  python scripts/demo_planning.py -p irrt_star --problem random_2d --iter_max 5000
  python scripts/demo_planning.py -p nirrt_star -n pointnet2 -c bfs --problem random_3d --iter_max 2000
"""
import argparse
import importlib
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nirrt_star_amd", "dropin"))
sys.path.insert(1, ROOT)

import numpy as np  # noqa: E402


def arg_parse():
    p = argparse.ArgumentParser()
    p.add_argument('-p', '--path_planner', default='rrt_star', help='rrt_star, irrt_star, nrrt_star, nirrt_star')
    p.add_argument('-n', '--neural_net', default='none', help='none, pointnet2')
    p.add_argument('-c', '--connect', default='none', help='none, bfs')
    p.add_argument('--device', default='cuda')
    p.add_argument('--step_len', type=float, default=10)
    p.add_argument('--iter_max', type=int, default=50000)
    p.add_argument('--clearance', type=float, default=0)
    p.add_argument('--pc_n_points', type=int, default=2048)
    p.add_argument('--pc_over_sample_scale', type=int, default=5)
    p.add_argument('--pc_sample_rate', type=float, default=0.5)
    p.add_argument('--pc_update_cost_ratio', type=float, default=None,
                   help='default: 0.9 for random_2d (demo_planning_2d.py:21), 1.0 for random_3d (demo_planning_3d.py:21)')
    p.add_argument('--connect_max_trial_attempts', type=int, default=5)
    p.add_argument('--problem', default='random_2d', help='random_2d, random_3d')
    p.add_argument('--seed', type=int, default=None)
    p.add_argument('--env_config_index', type=int, default=None)
    p.add_argument('--visualize', action='store_true')
    return p.parse_args()


def main():
    args = arg_parse()
    dim = "2d" if args.problem == "random_2d" else "3d"
    if args.pc_update_cost_ratio is None:      # the reference's two demo scripts differ here
        args.pc_update_cost_ratio = 0.9 if dim == "2d" else 1.0
    name = args.path_planner
    if args.neural_net != 'none':
        assert name in ('nirrt_star', 'nrrt_star')
        name += '_png'
        if args.connect == 'bfs':
            name += '_c'
    pkg = "path_planning_classes" if dim == "2d" else "path_planning_classes_3d"
    get_path_planner = importlib.import_module("%s.%s_%s" % (pkg, name, dim)).get_path_planner
    wrapper = None
    if args.neural_net == 'pointnet2':
        wpkg = "wrapper" if dim == "2d" else "wrapper_3d"
        mod = "pointnet2_wrapper_connect_bfs" if args.connect == 'bfs' else "pointnet2_wrapper"
        PNGWrapper = importlib.import_module("%s.pointnet_pointnet2.%s" % (wpkg, mod)).PNGWrapper
        from nirrt_star_amd import png_wrapper
        ck = png_wrapper.checkpoint_path('.', int(dim[0]))
        if not os.path.exists(ck):
            print("no trained checkpoint at %s: writing seeded synthetic weights (guidance quality is NOT meaningful)" % ck)
            png_wrapper.make_synthetic_checkpoint(ck, dim=int(dim[0]))
        wrapper = PNGWrapper(device=args.device)
    pu = importlib.import_module("datasets%s.planning_problem_utils_%s" % ("" if dim == "2d" else "_3d", dim))
    args.clearance = 3 if dim == "2d" else 2
    cfgs = getattr(pu, "get_random_%s_env_configs" % dim)()
    idx = args.env_config_index if args.env_config_index is not None else np.random.randint(len(cfgs))
    problem = getattr(pu, "get_random_%s_problem_input" % dim)(cfgs[idx])
    if args.seed is not None:
        np.random.seed(args.seed)
        random.seed(args.seed)
        import torch
        torch.manual_seed(args.seed)
    planner = get_path_planner(args, problem, wrapper)
    t0 = time.time()
    planner.planning(visualize=args.visualize)
    dt = time.time() - t0
    print("%s on %s #%d: %d iterations in %.2f s (%.0f it/s), %d vertices, path length %.3f, success %s"
          % (planner.get_path_planner_name(), args.problem, idx, args.iter_max, dt, args.iter_max / dt, planner.num_vertices,
             planner.get_path_len(planner.path), planner.check_success(planner.path)))


if __name__ == "__main__":
    main()
