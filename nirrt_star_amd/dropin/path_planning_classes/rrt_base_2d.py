"""Drop-in module: same import path and names as the reference's `path_planning_classes/rrt_base_2d.py`, backed by libnirrt_hip.so.
Put `nirrt_star_amd/dropin` first on sys.path."""
from nirrt_star_amd.planners import RRTStar2D as RRTBase2D  # noqa: F401
