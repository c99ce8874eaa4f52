"""Drop-in module: same import path and names as the reference's `datasets_3d/point_cloud_mask_utils_3d.py`, backed by libnirrt_hip.so.
Put `nirrt_star_amd/dropin` first on sys.path."""
from nirrt_star_amd.pointcloud import (ellipsoid_point_cloud_sampling_3d,  # noqa: F401
                                       generate_rectangle_point_cloud_3d)
