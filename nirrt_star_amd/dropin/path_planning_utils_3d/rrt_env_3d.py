"""Drop-in module: same import path and names as the reference's `path_planning_utils_3d/rrt_env_3d.py`, backed by libnirrt_hip.so.
Put `nirrt_star_amd/dropin` first on sys.path."""
from nirrt_star_amd.env import Env3D as Env  # noqa: F401
