"""Drop-in module: same import path and names as the reference's `wrapper/utils/bfs_connect_heuristic.py`, backed by libnirrt_hip.so.
Put `nirrt_star_amd/dropin` first on sys.path."""
from nirrt_star_amd.bfs_connect import *  # noqa: F401,F403
