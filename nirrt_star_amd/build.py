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
DEPS = SOURCES + [os.path.join(CSRC, "nirrt_device.hpp"), os.path.join(CSRC, "nirrt_kernels.inc"), os.path.join(CSRC, "glibc235_libm.inc"), os.path.join(CSRC, "glibc235_device.inc"),
                  os.path.join(os.path.dirname(HERE), "include", "nirrt_hip.h")]
# one module-wide LDS object at the same address in every kernel: the non-inlined loop-body functions then address it
# with constant offsets instead of a per-kernel offset-table lookup (see LdsData in csrc/nirrt_device.hpp)
# -fno-optimize-sibling-calls: keeps LLVM from marking the calls of the loop-body functions `tail`; only then does its
# inter-procedural register allocation drop the callee-saved saves of those local functions (48 VGPRs = 12.8 KB of
# scratch written and read back per wave and iteration otherwise - a quarter of the kernel's measured HBM writes).
# -sink-insts-to-avoid-spills: the persistent loops call the loop's phases with nothing alive in registers (csrc/nirrt_kernels.inc,
# run_tree_body) - except what machine LICM hoists out of the loop: LDS base addresses, zero constants.  Those were saved to
# scratch and reloaded around every call (9 dwords per lane and iteration); with this flag they are re-materialised where they
# are used and the loops of k_run_sample / k_run_pool have no scratch access at all.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-fno-optimize-sibling-calls", "-mllvm", "-amdgpu-lower-module-lds-strategy=module"]
# (an -mllvm option the compiler does not know is a hard error: probed once per build, dropped with a note if this ROCm lacks it -
#  the library is then correct but its persistent loops reload a few hoisted constants from scratch every iteration)
OPTIONAL_MLLVM = ["-sink-insts-to-avoid-spills=true"]


# Test-only second build with tiny compile-time limits, so that the overflow paths of the loop body (parent chains longer
# than the LDS chain cache, Near sets larger than the LDS stash) run in the parity suite (tests/test_hip_variants.py loads it
# through NIRRT_HIP_SO).  Same sources, same flags, different -D limits.
SO_SMALL = os.path.join(HERE, "libnirrt_hip_small.so")
SMALL_FLAGS = ["-DCHAIN_MAX=8", "-DNEAR_STASH=8"]


def needs_build(so=SO):
    if not os.path.exists(so):
        return True
    so_m = os.path.getmtime(so)
    return any(os.path.getmtime(p) > so_m for p in DEPS + [os.path.abspath(__file__)])


_optional = None


def optional_flags(hipcc):
    """the -mllvm options of OPTIONAL_MLLVM this compiler accepts (device-only compile of an empty kernel, a second per option)"""
    global _optional
    if _optional is None:
        import tempfile
        _optional = []
        with tempfile.TemporaryDirectory() as d:
            src = os.path.join(d, "probe.hip")
            with open(src, "w") as f:
                f.write("__global__ void k() {}\n")
            for opt in OPTIONAL_MLLVM:
                r = subprocess.run([hipcc, "--offload-arch=gfx950", "--cuda-device-only", "-c", "-mllvm", opt, "-o", os.path.join(d, "probe.o"), src],
                                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                if r.returncode == 0:
                    _optional += ["-mllvm", opt]
                else:
                    print("nirrt_star_amd.build: this hipcc does not know -mllvm %s: built without it" % opt, file=sys.stderr)
    return _optional


def _compile(so, extra, verbose):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + optional_flags(hipcc) + extra + ["-o", so] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)


def build(force=False, verbose=False, small=True):
    """Compile libnirrt_hip.so (and the small-limits test build) when a source is newer than the library."""
    if force or needs_build(SO):
        _compile(SO, os.environ.get("NIRRT_EXTRA_FLAGS", "").split(), verbose)
    if small and (force or needs_build(SO_SMALL)):
        _compile(SO_SMALL, SMALL_FLAGS, verbose)
    return SO


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(SO)
