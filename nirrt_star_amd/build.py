"""Build libnirrt_hip.so (gfx950) in-tree:  python -m nirrt_star_amd.build

hipcc cross-compiles without a GPU, so this runs in the CPU-only container; the built .so is
git-ignored but travels to the GPU box with the working tree.
-ffp-contract=off is REQUIRED: the kernels restate the reference's float64 formulas op by op
 and spell out the few fused ops with __builtin_fma themselves.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libnirrt_hip.so")
SOURCES = [os.path.join(CSRC, "nirrt_hip.hip"), os.path.join(CSRC, "pointops.hip")]
DEPS = SOURCES + [os.path.join(CSRC, "nirrt_device.hpp"), os.path.join(CSRC, "nirrt_kernels.inc"),
                  os.path.join(os.path.dirname(HERE), "include", "nirrt_hip.h")]
# one module-wide LDS object at the same address in every kernel: the non-inlined loop-body functions then address it
# with constant offsets instead of a per-kernel offset-table lookup (see LdsData in csrc/nirrt_device.hpp)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-mllvm", "-amdgpu-lower-module-lds-strategy=module"]


def needs_build():
    if not os.path.exists(SO):
        return True
    so_m = os.path.getmtime(SO)
    return any(os.path.getmtime(p) > so_m for p in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("NIRRT_EXTRA_FLAGS", "").split()
    cmd = [hipcc] + FLAGS + extra + ["-o", SO] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(SO)
