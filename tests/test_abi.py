"""CPU tests of the boundary: the HIP library is built, loads, and exports every symbol that
include/ygz_hip.h declares.  No compute call is made (there is no GPU in the build container)."""
import ctypes
import os
import re
from conftest import ROOT


def _declared():
    h = open(os.path.join(ROOT, "include", "ygz_hip.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(ygz_hip_\w+)\s*\(", h)))


def test_library_exports_every_declared_symbol(hip_lib):
    lib = hip_lib.load()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "libygz_hip.so does not export " + n
    assert set(hip_lib.ABI_SYMBOLS) == set(names)


def test_default_params_and_error_strings(hip_lib):
    lib = hip_lib.load()
    p = hip_lib.Params()
    lib.ygz_hip_default_params(ctypes.byref(p))
    assert (p.image_width, p.image_height, p.pyramid_levels, p.cell_size, p.fast_threshold) == (640, 480, 3, 10, 15)
    assert abs(p.fx - 520.9) < 1e-3 and abs(p.cy - 249.7) < 1e-3        # config/default.yaml:32-35
    assert lib.ygz_hip_error_string(0) == b"ok"
    assert b"capacity" in lib.ygz_hip_error_string(-4)
    k = hip_lib.KltParams()
    lib.ygz_hip_default_klt_params(ctypes.byref(k))
    assert (k.win, k.max_level, k.max_iter, k.use_initial_flow) == (21, 4, 30, 1)   # Tracker.h:25-27, Tracker.cpp:97


def test_create_rejects_bad_arguments_without_device(hip_lib):
    lib = hip_lib.load()
    p = hip_lib.Params()
    lib.ygz_hip_default_params(ctypes.byref(p))
    p.image_width = 0
    ctx = ctypes.c_void_p()
    assert lib.ygz_hip_create(ctypes.byref(ctx), 0, ctypes.byref(p), None) == hip_lib.E_INVALID
    assert not ctx
    assert lib.ygz_hip_create(None, 0, ctypes.byref(p), None) == hip_lib.E_INVALID


def test_product_package_does_not_touch_the_oracle():
    """the product path must never route through oracle/ (tier rule 3)"""
    pkg = os.path.join(ROOT, "ygz_slam_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "pyoracle" not in txt and "ygz_oracle" not in txt and "libygz_oracle" not in txt, f
    for dp, _, fs in os.walk(os.path.join(ROOT, "include")):
        for f in fs:
            txt = open(os.path.join(dp, f), errors="ignore").read()
            assert "ygz_oracle" not in txt, f


def test_header_is_plain_c_and_matches_the_loader():
    """include/ygz_hip.h is the boundary a cgo / JNI / ctypes binding reads: it has to compile as C99 without any C++ or HIP header,
    and every function it declares has to be in the loader's symbol list (and the other way round)."""
    import re
    import subprocess
    hdr = os.path.join(ROOT, "include", "ygz_hip.h")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", hdr], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    declared = set(re.findall(r"\b(ygz_hip_[a-z0-9_]+)\s*\(", open(hdr).read()))
    from ygz_slam_amd import _lib
    assert declared == set(_lib.ABI_SYMBOLS), (sorted(declared - set(_lib.ABI_SYMBOLS)), sorted(set(_lib.ABI_SYMBOLS) - declared))
