"""Drop-in module: same import path and names as the reference's `datasets/planning_problem_utils_2d.py`, backed by libnirrt_hip.so.
Put `nirrt_star_amd/dropin` first on sys.path."""
from nirrt_star_amd.problems import (compute_gamma_rrt_star, get_block_env_configs, get_block_problem_input,  # noqa: F401
                                     get_gap_env_configs, get_gap_problem_input, get_random_2d_env_configs,
                                     get_random_2d_problem_input)
