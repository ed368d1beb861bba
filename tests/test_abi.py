"""The C-ABI library loads on a machine without a GPU and exports every symbol include/linevis_hip.h declares
(no compute calls here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from common import ROOT
from linevis_amd import capi, host_api


def header_symbols():
    text = open(capi.HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lv_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_what_the_binding_binds():
    assert header_symbols() == sorted(capi.SYMBOLS)


def test_library_exports_every_declared_symbol():
    L = capi.load()
    for name in header_symbols():
        assert hasattr(L, name), name
    assert L.lv_version().startswith(b"linevis_hip")


def test_header_cites_reference_interfaces():
    text = open(capi.HEADER_PATH).read()
    for needle in ("LineRenderer.hpp", "LineRenderData.hpp:99-106", "LineData.cpp:1057-1075", "InternalState.hpp",
                   "VulkanRayTracer.cpp", "PerPixelLinkedListLineRenderer.cpp", "RenderingModes.hpp"):
        assert needle in text


def test_struct_layouts_match_the_header(tmp_path):
    """sizeof / offsetof of every field of lv_stats as a C99 compiler lays out include/linevis_hip.h = the ctypes mirror."""
    import subprocess
    fields = [n for n, _ in capi.Stats._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "linevis_hip.h"\nint main(void) {\n'
                   '  printf("%zu\\n", sizeof(lv_stats));\n'
                   + "".join('  printf("%%zu\\n", offsetof(lv_stats, %s));\n' % n for n in fields)
                   + '  printf("%zu %zu %zu\\n", sizeof(lv_line_point), sizeof(lv_tube_vertex), sizeof(lv_streamline_settings));\n'
                   '  return 0;\n}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.dirname(capi.HEADER_PATH), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True).split()
    assert int(out[0]) == ctypes.sizeof(capi.Stats)
    for n, off in zip(fields, out[1:1 + len(fields)]):
        assert int(off) == getattr(capi.Stats, n).offset, n
    assert [int(x) for x in out[1 + len(fields):]] == [48, 32, ctypes.sizeof(capi.StreamlineSettings)]
    assert capi.LINE_POINT_DTYPE.itemsize == 48
    assert capi.LINE_POINT_DTYPE.fields["lineNormal"][1] == 32
    assert capi.LINE_POINT_DTYPE.fields["lineStartIndex"][1] == 44


def test_no_cpu_fallback_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the failure path needs a GPU-less machine")
    with pytest.raises(capi.LineVisError):
        capi.Context(0)
    with pytest.raises(capi.LineVisError):
        host_api.HeadlessLineRenderer()


def test_host_library_loads():
    L = host_api.load()
    for name in ("lvh_flow_create", "lvh_flow_build_render_data", "lvh_flow_build_triangle_data", "lvh_grid_create",
                 "lvh_grid_trace", "lvh_renderer_create", "lvh_renderer_render"):
        assert hasattr(L, name)


def test_product_never_touches_the_oracle():
    """The product path must not import, link or call anything under oracle/."""
    bad = []
    for base in ("linevis_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            if "_lib" in dirpath or "__pycache__" in dirpath:
                continue
            for f in files:
                if f.endswith((".py", ".h", ".hpp", ".cpp", ".hip")):
                    txt = open(os.path.join(dirpath, f), errors="replace").read()
                    if re.search(r"\boracle\b|lv_oracle|lvo_", txt) and f != "LvMath.hpp":
                        bad.append(os.path.join(dirpath, f))
    # LvMath.hpp mentions the word in a comment only
    assert bad == [], bad


def _build_abi_smoke(tmp_path):
    import subprocess
    lib_dir = os.path.dirname(capi.LIB_PATH)
    exe = str(tmp_path / "abi_smoke")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.dirname(capi.HEADER_PATH),
                           os.path.join(ROOT, "tests", "abi_smoke.c"), "-L", lib_dir, "-llinevis_hip", "-Wl,-rpath," + lib_dir, "-o", exe])
    return exe


def test_compiled_c_program_links_the_boundary_and_fails_cleanly_without_a_device(tmp_path):
    """tests/abi_smoke.c is plain C99 (-pedantic -Werror), links nothing but liblinevis_hip.so, and without a GPU lv_create reports
    LV_E_HIP instead of falling back to anything."""
    import subprocess
    import torch
    capi.load()
    exe = _build_abi_smoke(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the -m gpu test runs the program")
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "abi_smoke.bin")], capture_output=True, text=True)
    assert r.returncode == 4 and "lv_create(0) failed with -2" in r.stderr and r.stdout.startswith("linevis_hip")


@pytest.mark.gpu
def test_compiled_c_program_renders_the_fixture(tmp_path):
    """The boundary driven from C without Python in between: both renderers on the small fixture, frames within 2 LSB of the CPU
    checker's (tests/golden/abi_smoke.bin, written by tests/golden/make_golden.py)."""
    import subprocess
    capi.load()
    exe = _build_abi_smoke(tmp_path)
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "abi_smoke.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "abi_smoke ok" in r.stdout and "mode 11: max difference" in r.stdout and "mode 2: max difference" in r.stdout
