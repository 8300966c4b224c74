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


def test_offline_header_is_plain_c_and_the_host_library_exports_it():
    """include/ygz_offline.h (the C++ offline driver's boundary, libygz_host.so) compiles as C99, every function it declares is exported and
    bound by ygz_slam_amd/offline.py, and the ctypes mirror of its parameter block has the header's defaults and size"""
    import subprocess
    from ygz_slam_amd import offline
    hdr = os.path.join(ROOT, "include", "ygz_offline.h")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", "-I", os.path.join(ROOT, "include"), hdr], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    txt = re.sub(r"/\*.*?\*/", "", open(hdr).read(), flags=re.S)
    declared = set(re.findall(r"\b(ygz_offline_[a-z0-9_]+)\s*\(", txt)) - {"ygz_offline_chunk_fn"}
    assert declared == set(offline.OFFLINE_SYMBOLS), (sorted(declared - set(offline.OFFLINE_SYMBOLS)), sorted(set(offline.OFFLINE_SYMBOLS) - declared))
    lib = offline.host_lib()
    for n in declared:
        assert hasattr(lib, n), "libygz_host.so does not export " + n
    p = offline.OffParams()
    lib.ygz_offline_default_params(ctypes.byref(p))
    assert (p.width, p.height, p.levels, p.chunk, p.kf_stride, p.window_kfs, p.max_points, p.ba_iterations, p.lanes) == (1280, 720, 3, 128, 8, 8, 2000, 20, 3)
    assert (p.obs_mode, p.ba_rounds, p.frame_channels, p.depth_kind, p.pipeline_ba, p.defer_gaps, p.ramp, p.kf_tail) == (1, 1, 3, 1, 1, -1, 1, 1)
    assert abs(p.outlier_chi2 - 5.991) < 1e-12 and abs(p.depth_scale - 1 / 5000.0) < 1e-15
    # sizeof through a C probe: the mirror has the header's layout
    src = '#include "ygz_offline.h"\n#include <stdio.h>\nint main(void){printf("%zu %zu %zu", sizeof(ygz_offline_params), sizeof(ygz_offline_results), sizeof(ygz_offline_exchange));return 0;}'
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        sizes = [int(x) for x in subprocess.check_output([os.path.join(d, "s")]).split()]
    assert sizes == [ctypes.sizeof(offline.OffParams), ctypes.sizeof(offline.OffResults), ctypes.sizeof(offline.OffExchange)]
    # bad arguments are refused before any device is touched
    h = ctypes.c_void_p()
    p.world, p.rank = 2, 0
    assert lib.ygz_offline_create(ctypes.byref(h), ctypes.byref(p), None, None) == -1 and not h          # world > 1 needs an id or a hook
    p.world, p.window_kfs = 1, 1
    assert lib.ygz_offline_create(ctypes.byref(h), ctypes.byref(p), None, None) == -1 and not h


def test_abi_version_is_checked_at_load(hip_lib):
    lib = hip_lib.load()
    assert lib.ygz_hip_abi_version() == hip_lib.ABI_VERSION
    hdr = open(os.path.join(ROOT, "include", "ygz_hip.h")).read()
    assert int(re.search(r"#define YGZ_HIP_ABI_VERSION\s+(\d+)", hdr).group(1)) == hip_lib.ABI_VERSION


def test_frame_covisibility_members():
    """ygz::Frame::UpdateConnections / GetBestCovisibilityKeyframes / AddConnection / UpdateBestCovisibles (src/Basic/Frame.cpp:73-176; what
    src/Module/LocalMapping.cpp:256,345,378,586 reads) on a hand-made map -- host bookkeeping, no device (tests/cpp/test_surface covis)."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "cpp", "test_surface")
    assert os.path.exists(exe), "tests/cpp/test_surface is not built (run __graft_entry__.build())"
    out = subprocess.check_output([exe, "covis"], timeout=60).decode().strip().split("\n")
    # keyframe 3 shares 20 / 16 / 5 good map points with keyframes 0 / 1 / 2 (7 bad ones with keyframe 2 do not count): threshold 15, heaviest first
    assert out[0] == "kf3 cov 0:20 1:16 | connected 0:20 1:16 2:5 | best1 1:0 best10 2"
    # keyframe 2 (5 with keyframe 3, 3 with keyframe 0): below the threshold everywhere -> the single best neighbour, which is told about it
    assert out[1] == "kf2 cov 3:5 | kf3 now sees kf2 with 5 | in frustum 1"
    # a second UpdateConnections replaces the lists; UpdateBestCovisibles appends all connections, heaviest first; no map points: nothing
    assert out[2] == "kf3 again 2 then 0:20 1:16 2:5 | empty 0 0"
