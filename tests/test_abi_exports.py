"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU and exports every
symbol include/nirrt_hip.h declares; the product refuses to run (loudly) when no device exists."""
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols(header="nirrt_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nirrt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from nirrt_star_amd import _hip, build
    build.build()
    L = _hip.load()
    syms = _declared_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(L, s), "missing export %s" % s
    assert set(_hip.EXPORTS) == set(syms)
    pn = _declared_symbols("nirrt_pointops.h")
    assert len(pn) == 18      # (round 6: + the four _ragged entry points)
    for s in pn:
        assert hasattr(L, s), "missing export %s" % s


def test_struct_layouts_match_header():
    import ctypes as C
    from nirrt_star_amd import _hip
    # nirrt_step_result: 2*i32, 2*i64, 4*i32, i64, 3*f64, f64, i64, i64, 2*i32
    assert C.sizeof(_hip.StepResult) == 8 + 16 + 16 + 8 + 24 + 8 + 8 + 8 + 8
    assert _hip.StepResult.c_best.offset == 72
    assert C.sizeof(_hip.Config) == 8 + 8 + 24 + 24 + 24 + 24 + 24 + 8 + 8 + 8 + 8
    assert C.sizeof(_hip.RunArgs) == 16 + 8 + 8 * 18
    assert _hip.RunArgs.struct_size.offset == 0 and _hip.RunArgs.iters.offset == 16


def test_run_args_size_and_version_match_the_c_header(tmp_path):
    """the ctypes mirror against the header itself: a C compiler's sizeof / offsetof of nirrt_run_args and NIRRT_ABI_VERSION, and the
    version the built library reports (nirrt_run refuses a struct of another size; _hip.load() a library of another version)"""
    import ctypes as C
    import subprocess
    from nirrt_star_amd import _hip
    src = tmp_path / "abi.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "nirrt_hip.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu %d\\n", sizeof(nirrt_run_args), offsetof(nirrt_run_args, iters), '
                   'offsetof(nirrt_run_args, run_ahead), sizeof(nirrt_step_result), NIRRT_ABI_VERSION); return 0; }\n')
    exe = str(tmp_path / "abi")
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe])
    size, off_iters, off_ahead, step, ver = (int(x) for x in subprocess.check_output([exe], text=True).split())
    assert size == C.sizeof(_hip.RunArgs) and off_iters == _hip.RunArgs.iters.offset and off_ahead == _hip.RunArgs.run_ahead.offset
    assert step == C.sizeof(_hip.StepResult)
    assert ver == _hip.ABI_VERSION == _hip.load().nirrt_abi_version()


def test_no_silent_cpu_fallback_without_device():
    from nirrt_star_amd import _hip, worlds
    if _hip.device_count() > 0:
        pytest.skip("a GPU is visible")
    pr = worlds.problem_2d(worlds.random_world_2d(0), 0)
    with pytest.raises(_hip.NirrtError):
        _hip.HipTree(2, 100, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], 3, pr["env"])
