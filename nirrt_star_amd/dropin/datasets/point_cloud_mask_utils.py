"""Drop-in module: same import path and names as the reference's `datasets/point_cloud_mask_utils.py`, backed by libnirrt_hip.so.
Put `nirrt_star_amd/dropin` first on sys.path."""
from nirrt_star_amd.pointcloud import (ellipsoid_point_cloud_sampling, generate_rectangle_point_cloud,  # noqa: F401
                                       get_point_cloud_mask_around_points)
