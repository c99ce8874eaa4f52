"""Drop-in module: same import path and names as the reference's `path_planning_classes_3d/nrrt_star_png_c_3d.py`, backed by libnirrt_hip.so.
Put `nirrt_star_amd/dropin` first on sys.path."""
from nirrt_star_amd.planners import NRRTStarPNGC3D  # noqa: F401
from nirrt_star_amd.planners import get_nrrt_star_png_c_3d as get_path_planner  # noqa: F401
