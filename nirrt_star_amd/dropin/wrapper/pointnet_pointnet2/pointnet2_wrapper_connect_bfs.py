"""Drop-in module: same import path and names as the reference's `wrapper/pointnet_pointnet2/pointnet2_wrapper_connect_bfs.py`, backed by libnirrt_hip.so.
Put `nirrt_star_amd/dropin` first on sys.path."""
from nirrt_star_amd.png_wrapper import PNGWrapper as PNGWrapper  # noqa: F401
