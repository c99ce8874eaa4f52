"""Drop-in module: same import path and names as the reference's `pointnet_pointnet2/PathPlanDataLoader.py`.
Put `nirrt_star_amd/dropin` first on sys.path."""
from nirrt_star_amd.path_plan_dataset import PathPlanDataset  # noqa: F401
