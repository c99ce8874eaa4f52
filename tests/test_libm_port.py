"""CPU: csrc/glibc235_libm.inc (glibc 2.35's sin / cos / atan2, FMA variants, restated instruction by instruction - what the
device's 2D steer evaluates) against the libm of the machine the tests run on: bit-identical over random steer arguments
(rrt_star_2d.py:67-78: theta = math.atan2(dy, dx), math.cos(theta), math.sin(theta)) and wide-range angles, and against CPython's
math module itself on a small sample.  Holds on an x86-64 host with FMA + AVX2 and glibc 2.35 (this image, the GPU box's image);
elsewhere the reference itself computes other values and the test says so instead of failing."""
import ctypes as C
import math
import os
import platform
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PORT = os.path.join(ROOT, "scripts", "libm_port")


def _host_matches():
    if platform.machine() != "x86_64" or platform.libc_ver() != ("glibc", "2.35"):
        return False
    with open("/proc/cpuinfo") as f:
        flags = next((l for l in f if l.startswith("flags")), "")
    return " fma " in flags + " " and " avx2 " in flags + " "


pytestmark = pytest.mark.skipif(not _host_matches(), reason="the restated functions are glibc 2.35's x86-64 FMA variants: another "
                                                            "libm / CPU resolves math.sin / cos / atan2 to other code")


def test_restated_functions_equal_this_machines_libm(tmp_path):
    exe = str(tmp_path / "libm_harness")
    subprocess.check_call(["gcc", "-O2", "-fno-builtin", "-ffp-contract=off", "-Wno-unused-label", "-Wno-unused-variable",
                           "-Wno-unused-but-set-variable", os.path.join(PORT, "harness.c"), "-lm", "-o", exe], cwd=PORT)
    r = subprocess.run([exe, "3000000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:]
    assert "atan2 mismatches 0, sin 0, cos 0, wide-range sin/cos 0, unsupported paths hit 0" in r.stdout


def test_restated_steer_equals_cpython_math(tmp_path):
    """the reference's new_state (rrt_star_2d.py:67-78) evaluated with CPython's math module, against the restated functions
    through a small shared library: bit-identical node_new"""
    src = tmp_path / "steer.c"
    src.write_text('''
#include <math.h>
#include <stdint.h>
#include <string.h>
static inline double D(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
static inline uint64_t B(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
#define DB(u) D(u)
#define LIBM_CONST(name, val) static const uint64_t name = val;
#define LIBM_TABLE(name, n) static const uint64_t name[n]
#define LIBM_FN static
#define UNSUPPORTED(msg) return NAN
#define S64(off) stk[(off) / 8]
#define W64(off, v) (stk[(off) / 8] = (v))
#define S32(off) ((uint32_t)(stk[(off) / 8] >> (((off) & 4) * 8)))
#define W32(off, v) (stk[(off) / 8] = (stk[(off) / 8] & ~(0xffffffffull << (((off) & 4) * 8))) | ((uint64_t)(uint32_t)(v) << (((off) & 4) * 8)))
static inline uint64_t LD64(int64_t a);
#define LD32(a) ((uint32_t)LD64(a))
#include "%s"
static inline uint64_t LD64(int64_t a)
{
    if (a >= LIBM_T_SINCOS_BASE && a < LIBM_T_SINCOS_BASE + 8 * 440) return T_sincos[(a - LIBM_T_SINCOS_BASE) / 8];
    if (a >= LIBM_T_ATAN_BASE && a < LIBM_T_ATAN_BASE + 8 * 241 * 7) return T_atan[(a - LIBM_T_ATAN_BASE) / 8];
    return 0;
}
void steer2d(double fx, double fy, double dx, double dy, double m, double *out)
{
    double theta = glibc_atan2(dy, dx);
    out[0] = fx + m * glibc_cos(theta, 0.);
    out[1] = fy + m * glibc_sin(theta, 0.);
}
''' % os.path.join(ROOT, "nirrt_star_amd", "csrc", "glibc235_libm.inc"))
    so = str(tmp_path / "steer.so")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-fno-builtin", "-ffp-contract=off", "-Wno-unused-label", "-Wno-unused-variable",
                           "-Wno-unused-but-set-variable", str(src), "-lm", "-o", so])
    L = C.CDLL(so)
    L.steer2d.argtypes = [C.c_double] * 5 + [C.POINTER(C.c_double)]
    rng = np.random.default_rng(5)
    out = (C.c_double * 2)()
    for _ in range(20000):
        fx, fy, tx, ty = rng.uniform(0, 224, 4)
        dx, dy = tx - fx, ty - fy
        m = min(math.hypot(dx, dy), 10.0)
        theta = math.atan2(dy, dx)
        ex, ey = fx + m * math.cos(theta), fy + m * math.sin(theta)
        L.steer2d(fx, fy, dx, dy, m, out)
        assert out[0] == ex and out[1] == ey
