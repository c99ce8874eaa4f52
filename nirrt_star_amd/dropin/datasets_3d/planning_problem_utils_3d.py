"""Drop-in module: same import path and names as the reference's `datasets_3d/planning_problem_utils_3d.py`, backed by libnirrt_hip.so.
Put `nirrt_star_amd/dropin` first on sys.path."""
from nirrt_star_amd.problems import (compute_gamma_rrt_star_3d, get_random_3d_env_configs,  # noqa: F401
                                     get_random_3d_problem_input)
