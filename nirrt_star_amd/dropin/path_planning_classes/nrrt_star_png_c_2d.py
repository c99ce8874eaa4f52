"""Drop-in module: same import path and names as the reference's `path_planning_classes/nrrt_star_png_c_2d.py`, backed by libnirrt_hip.so.
Put `nirrt_star_amd/dropin` first on sys.path."""
from nirrt_star_amd.planners import NRRTStarPNGC2D  # noqa: F401
from nirrt_star_amd.planners import get_nrrt_star_png_c_2d as get_path_planner  # noqa: F401
