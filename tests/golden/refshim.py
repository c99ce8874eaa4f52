"""Import shim for the READ-ONLY reference at /root/reference (this container only).

Used solely by ``make_golden.py`` to generate fixtures; nothing under ``tests/`` that runs
on the GPU box imports this (``/root/reference`` does not exist there).

* the reference has no ``__init__.py`` anywhere, and site-packages holds an unrelated
  ``datasets`` (HuggingFace) that shadows its ``datasets/`` namespace package -> pre-seed;
* cv2 and open3d are not installed -> stubbed.  open3d's farthest_point_down_sample is replaced by a RECORDER
  (FPS_CALLS keeps every candidate set the reference hands to it - the fixtures pin those) followed by the
  repo's restatement of the down-sampling (oracle/pointops_ref.py; parity of that one call is unpinned);
* never write bytecode into the reference tree.
"""
import os
import sys
import types

REF = "/root/reference"
FPS_CALLS = []   # (points (n, 3) float64 copy, num_samples) of every open3d farthest_point_down_sample call


def install():
    sys.dont_write_bytecode = True
    os.environ.setdefault("MPLBACKEND", "Agg")
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for name in ("datasets", "datasets_3d"):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, name)]
        sys.modules[name] = m
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    if "open3d" not in sys.modules:
        o3d = types.ModuleType("open3d")

        class _PC:   # stand-in for open3d.geometry.PointCloud: FPS delegated to OUR restatement
            def __init__(self):
                self.points = None

            def farthest_point_down_sample(self, num_samples):
                import numpy as np
                from oracle import pointops_ref
                pts = np.ascontiguousarray(self.points, dtype=np.float64)
                FPS_CALLS.append((pts.copy(), int(num_samples)))
                out = _PC()
                out.points = pts[pointops_ref.farthest_point_down_sample_f64(pts, num_samples)] if num_samples < len(pts) else pts.copy()
                return out

        o3d.geometry = types.SimpleNamespace(PointCloud=_PC)
        o3d.utility = types.SimpleNamespace(Vector3dVector=lambda a: a)
        sys.modules["open3d"] = o3d
    return REF
