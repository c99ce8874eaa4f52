"""nirrt_star_amd — MI355X-native RRT*/IRRT*/NIRRT* planning inner loop.

The tree (vertex SoA + parent array) lives in HBM and every iteration's
nearest / Near / segment-collision / cost-walk / choose-parent / rewire runs
in hand-written gfx950 HIP kernels behind a C ABI (``include/nirrt_hip.h``).
The host side here mirrors the reference's Python planner interface
(``path_planning_classes{,_3d}`` of tedhuang96/nirrt_star) so that the
reference's demo/eval drivers can switch over by putting
``nirrt_star_amd/dropin`` first on ``sys.path``.

Nothing in this package imports ``oracle/`` — the oracle is test
infrastructure only, and every product path raises if the HIP library is
missing instead of falling back to a CPU implementation.
"""

__version__ = "0.1.0"
