"""ctypes loader of libygz_hip.so (the C ABI declared in include/ygz_hip.h).

This is harness plumbing for tests/ and bench.py: numpy arrays in, numpy arrays out, every
call goes straight through the C ABI.  There is NO fallback path: if the HIP library is
missing or no gfx950 device is usable the calls raise.
"""
import ctypes as C
import os
import sys
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libygz_hip.so")
MAX_LEVELS = 8

OK, E_INVALID, E_HIP, E_NO_DEVICE, E_CAPACITY, E_STATE = 0, -1, -2, -3, -4, -5
ABI_VERSION = 6                     # YGZ_HIP_ABI_VERSION of include/ygz_hip.h this file mirrors


class YgzHipError(RuntimeError):
    def __init__(self, code, what, hip_err=0):
        super().__init__("%s failed: code %d (hip error %d)" % (what, code, hip_err))
        self.code = code


class Params(C.Structure):
    _fields_ = [("image_width", C.c_int), ("image_height", C.c_int), ("pyramid_levels", C.c_int),
                ("cell_size", C.c_int), ("fast_threshold", C.c_int), ("nms_tie_suppress", C.c_int),
                ("max_frames", C.c_int), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("debug_maps", C.c_int)]


class KptSoa(C.Structure):
    _fields_ = [("px", C.POINTER(C.c_double)), ("level", C.POINTER(C.c_int32)), ("score", C.POINTER(C.c_float)),
                ("angle", C.POINTER(C.c_float)), ("desc", C.POINTER(C.c_uint8))]


class AlignPair(C.Structure):
    _fields_ = [("ref_slot", C.c_int), ("cur_slot", C.c_int), ("T_ref", C.c_double * 7), ("T_cur", C.c_double * 7)]


class LocalMap(C.Structure):
    _fields_ = [("n_points", C.c_int), ("pos_world", C.POINTER(C.c_double)), ("point_bad", C.POINTER(C.c_uint8)),
                ("n_keyframes", C.c_int), ("kf_slot", C.POINTER(C.c_int32)), ("kf_T", C.POINTER(C.c_double)),
                ("n_candidates", C.c_int), ("cand_point", C.POINTER(C.c_int32)), ("cand_kf", C.POINTER(C.c_int32)),
                ("cand_level", C.POINTER(C.c_int32)), ("cand_px_ref", C.POINTER(C.c_double))]


class KltParams(C.Structure):
    _fields_ = [("win", C.c_int), ("max_level", C.c_int), ("max_iter", C.c_int), ("eps", C.c_double),
                ("min_eig_threshold", C.c_double), ("use_initial_flow", C.c_int)]


class BaProblem(C.Structure):
    _fields_ = [("n_poses", C.c_int), ("n_points", C.c_int), ("n_edges", C.c_int),
                ("poses", C.POINTER(C.c_double)), ("pose_fixed", C.POINTER(C.c_uint8)),
                ("points", C.POINTER(C.c_double)), ("edge_pose", C.POINTER(C.c_int32)),
                ("edge_point", C.POINTER(C.c_int32)), ("obs", C.POINTER(C.c_double)),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("huber_delta", C.c_double), ("formulation", C.c_int),
                ("point_fixed", C.POINTER(C.c_uint8)), ("edge_huber", C.POINTER(C.c_double)),
                ("edge_enable", C.POINTER(C.c_uint8))]


class CeresOptions(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int), ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
                ("parameter_tolerance", C.c_double), ("initial_trust_region_radius", C.c_double),
                ("max_trust_region_radius", C.c_double), ("min_trust_region_radius", C.c_double),
                ("min_relative_decrease", C.c_double), ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
                ("jacobi_scaling", C.c_int), ("max_num_consecutive_invalid_steps", C.c_int), ("fail_behind_camera", C.c_int),
                ("trust_region_strategy", C.c_int)]


class CeresSummary(C.Structure):
    _fields_ = [("iterations", C.c_int), ("successful_steps", C.c_int), ("unsuccessful_steps", C.c_int),
                ("termination", C.c_int), ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("final_radius", C.c_double)]


class BaStats(C.Structure):
    _fields_ = [("iterations", C.c_int), ("lm_trials", C.c_int), ("chi2_initial", C.c_double), ("chi2_final", C.c_double),
                ("lambda_final", C.c_double)]


# every symbol include/ygz_hip.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "ygz_hip_default_params", "ygz_hip_create", "ygz_hip_destroy", "ygz_hip_synchronize", "ygz_hip_join", "ygz_hip_set_overlap", "ygz_hip_error_string",
    "ygz_hip_last_hip_error", "ygz_hip_max_keypoints", "ygz_hip_timer_begin", "ygz_hip_timer_end", "ygz_hip_probe_begin", "ygz_hip_probe_end",
    "ygz_hip_upload_bgr", "ygz_hip_upload_gray", "ygz_hip_build_pyramid", "ygz_hip_download_level", "ygz_hip_download_framed_level", "ygz_hip_level_size",
    "ygz_hip_detect", "ygz_hip_keypoint_count", "ygz_hip_get_keypoints", "ygz_hip_describe", "ygz_hip_describe_given_angle", "ygz_hip_get_fast_maps",
    "ygz_hip_match_slots", "ygz_hip_match_slots_again", "ygz_hip_get_matches", "ygz_hip_hamming_match",
    "ygz_hip_find_direct_projection", "ygz_hip_align2d", "ygz_hip_sparse_align", "ygz_hip_sparse_align_residuals",
    "ygz_hip_default_klt_params", "ygz_hip_klt_track", "ygz_hip_klt_track_filtered",
    "ygz_hip_set_keypoint_depths", "ygz_hip_track_begin", "ygz_hip_track_reload", "ygz_hip_track_klt", "ygz_hip_track_klt_prepare", "ygz_hip_track_direct",
    "ygz_hip_track_sparse_align", "ygz_hip_track_get_klt", "ygz_hip_track_get_direct", "ygz_hip_track_get_pose",
    "ygz_hip_ba_linearize", "ygz_hip_ba_upload", "ygz_hip_ba_set_state", "ygz_hip_ba_set_state_device", "ygz_hip_ba_linearize_resident", "ygz_hip_ba_download", "ygz_hip_ba_optimize",
    "ygz_hip_ba_optimize_resident", "ygz_hip_ba_set_team_budget", "ygz_hip_ba_get_state", "ygz_hip_ba_behind_camera", "ygz_hip_ba_set_enable", "ygz_hip_ba_light_barrier", "ygz_hip_ceres_default_options", "ygz_hip_ba_solve_ceres", "ygz_hip_ba_solve_ceres_resident", "ygz_hip_optimize_pose_only",
    "ygz_hip_vocab_load", "ygz_hip_vocab_info", "ygz_hip_compute_bow", "ygz_hip_get_bow", "ygz_hip_bow_transform", "ygz_hip_search_by_bow_slots", "ygz_hip_search_by_bow", "ygz_hip_depth_from_triangulation", "ygz_hip_track_local_map", "ygz_hip_find_direct_projection_mp", "ygz_hip_find_direct_projection_mp_begin", "ygz_hip_find_direct_projection_mp_end", "ygz_hip_set_wait_hook",
    "ygz_hip_match_postfilter", "ygz_hip_get_good_matches", "ygz_hip_match_postfilter_host", "ygz_hip_match_sets", "ygz_hip_check_frame_descriptors",
    "ygz_hip_check_descriptor_pairs", "ygz_hip_track_adopt_pose", "ygz_hip_track_pose_only", "ygz_hip_track_get_pose_only",
    "ygz_hip_pinned_alloc", "ygz_hip_pinned_free", "ygz_hip_upload_bgr_batch", "ygz_hip_upload_gray_batch", "ygz_hip_get_keypoint_pixels_batch",
    "ygz_hip_get_keypoints_batch", "ygz_hip_set_keypoint_depths_batch", "ygz_hip_track_get_summary", "ygz_hip_create_map_points", "ygz_hip_depth_filter_update",
    "ygz_hip_get_keypoint_counts", "ygz_hip_get_keypoint_depths", "ygz_hip_upload_depth_batch", "ygz_hip_keypoint_depths_from_image", "ygz_hip_ba_get_stats", "ygz_hip_se3_chain", "ygz_hip_stream_wait", "ygz_hip_mark", "ygz_hip_wait_mark",
    "ygz_hip_kf_row_bytes", "ygz_hip_kf_store_create", "ygz_hip_kf_store_info", "ygz_hip_kf_store_put", "ygz_hip_kf_store_put_trel",
    "ygz_hip_kf_store_set_trel", "ygz_hip_kf_store_refresh", "ygz_hip_ba_reserve_windows", "ygz_hip_ba_build_windows", "ygz_hip_ba_pack_states", "ygz_hip_ba_mark_outliers", "ygz_hip_ba_get_outlier_stats", "ygz_hip_bow_orientation", "ygz_hip_bow_orientation_slots", "ygz_hip_ba_last_path",
    "ygz_hip_abi_version", "ygz_hip_ba_optimize_chi2", "ygz_hip_ba_set_team_placement", "ygz_hip_get_stream", "ygz_hip_get_device", "ygz_hip_make_current", "ygz_hip_device_alloc", "ygz_hip_device_free", "ygz_hip_copy",
]

SUMMARY_FIELDS = 32

_lib = None


def load():
    """Load the in-tree HIP library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libygz_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        # PyTorch-ROCm wheels bundle their own HIP runtime (torch/lib/libamdhip64.so).  If this library is mapped first it pulls in
        # /opt/rocm's copy, a later `import torch` maps the second one, and torch then reports "No HIP GPUs are available".  Mapped
        # after torch, libygz_hip.so binds to the runtime that is already there.  This harness (tests / bench.py, which use
        # torch.distributed) therefore imports torch first when it is installed; a C / C++ caller is not concerned.
        if "torch" not in sys.modules and os.environ.get("YGZ_NO_TORCH_PRELOAD") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        _lib = C.CDLL(LIB_PATH)
        # C has no name mangling: a library built from an older header would still load.  Refuse it here.
        if not hasattr(_lib, "ygz_hip_abi_version") or _lib.ygz_hip_abi_version() != ABI_VERSION:
            got = _lib.ygz_hip_abi_version() if hasattr(_lib, "ygz_hip_abi_version") else None
            _lib = None
            raise ImportError("libygz_hip.so has ABI version %r, this binding was written for %d: rebuild (python -c 'import __graft_entry__ as g; g.build()')" % (got, ABI_VERSION))
        _lib.ygz_hip_error_string.restype = C.c_char_p
        _lib.ygz_hip_kf_row_bytes.restype = C.c_size_t
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def se3_chain(T_rel):
    """T[0] = identity, T[i] = T_rel[i] * T[i - 1] (host code of the library)"""
    T = np.ascontiguousarray(T_rel, np.float64).reshape(-1, 7)
    out = np.empty_like(T)
    rc = load().ygz_hip_se3_chain(_p(T, C.c_double), len(T), _p(out, C.c_double))
    if rc != OK:
        raise YgzHipError(rc, "se3_chain")
    return out


class PinnedArray:
    """numpy view of page-locked host memory (ygz_hip_pinned_alloc) -- asynchronous copies need it"""

    def __init__(self, shape, dtype):
        lib = load()
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self._ptr = C.c_void_p()
        lib.ygz_hip_pinned_alloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        rc = lib.ygz_hip_pinned_alloc(C.byref(self._ptr), max(self.nbytes, 64))
        if rc != OK:
            raise YgzHipError(rc, "ygz_hip_pinned_alloc")
        buf = (C.c_uint8 * max(self.nbytes, 64)).from_address(self._ptr.value)
        self.array = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def free(self):
        if self._ptr:
            self.array = None
            load().ygz_hip_pinned_free.argtypes = [C.c_void_p]
            load().ygz_hip_pinned_free(self._ptr)
            self._ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class HipContext:
    """One ygz_hip_ctx: owns HBM frame slots on one GPU and one HIP stream."""

    def __init__(self, width=640, height=480, levels=3, max_frames=8, device=0, debug_maps=False, stream=None,
                 fast_threshold=15, cell_size=10, nms_tie_suppress=0):
        self.lib = load()
        p = Params()
        self.lib.ygz_hip_default_params(C.byref(p))
        p.image_width, p.image_height, p.pyramid_levels = width, height, levels
        p.max_frames, p.debug_maps = max_frames, int(debug_maps)
        p.fast_threshold, p.cell_size, p.nms_tie_suppress = fast_threshold, cell_size, nms_tie_suppress
        self.params = p
        self._ctx = C.c_void_p()
        rc = self.lib.ygz_hip_create(C.byref(self._ctx), device, C.byref(p), C.c_void_p(stream))
        if rc != OK:
            self._ctx = C.c_void_p()
            raise YgzHipError(rc, "ygz_hip_create")
        self.width, self.height, self.levels, self.max_frames = width, height, levels, max_frames
        self.cells = self.lib.ygz_hip_max_keypoints(self._ctx)

    @classmethod
    def from_handle(cls, ptr, width, height, levels=3):
        """a NON-OWNING view of a context somebody else created (the lanes / BA context of the C++ offline driver): close() leaves it alone"""
        self = cls.__new__(cls)
        self.lib = load()
        p = Params()
        self.lib.ygz_hip_default_params(C.byref(p))
        p.image_width, p.image_height, p.pyramid_levels = width, height, levels
        self.params = p
        self._ctx = C.c_void_p(ptr)
        self._borrowed = True
        self.width, self.height, self.levels, self.max_frames = width, height, levels, 0
        self.cells = self.lib.ygz_hip_max_keypoints(self._ctx)
        return self

    def _chk(self, rc, what):
        if rc != OK:
            raise YgzHipError(rc, what, self.lib.ygz_hip_last_hip_error(self._ctx))

    def close(self):
        if self._ctx and not getattr(self, "_borrowed", False):
            self.lib.ygz_hip_destroy(self._ctx)
        self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        self._chk(self.lib.ygz_hip_synchronize(self._ctx), "synchronize")

    def join(self):
        """the context's stream waits for the stages pending on side streams; the host does not block"""
        self._chk(self.lib.ygz_hip_join(self._ctx), "join")

    def set_overlap(self, enable=True):
        self._chk(self.lib.ygz_hip_set_overlap(self._ctx, int(enable)), "set_overlap")

    def timer_begin(self):
        self._chk(self.lib.ygz_hip_timer_begin(self._ctx), "timer_begin")

    def timer_end(self):
        ms = C.c_float(0)
        self._chk(self.lib.ygz_hip_timer_end(self._ctx, C.byref(ms)), "timer_end")
        return ms.value

    def probe_begin(self, kernel, max_launches=4096):
        self._chk(self.lib.ygz_hip_probe_begin(self._ctx, kernel.encode(), max_launches), "probe_begin")

    def probe_end(self):
        ms, n = C.c_double(0), C.c_int(0)
        self._chk(self.lib.ygz_hip_probe_end(self._ctx, C.byref(ms), C.byref(n)), "probe_end")
        return ms.value, n.value

    # ---- frames
    def upload_gray(self, slot, gray):
        gray = np.ascontiguousarray(gray, np.uint8)
        assert gray.shape == (self.height, self.width)
        self._chk(self.lib.ygz_hip_upload_gray(self._ctx, slot, _p(gray, C.c_uint8), self.width), "upload_gray")

    def upload_bgr(self, slot, bgr):
        bgr = np.ascontiguousarray(bgr, np.uint8)
        assert bgr.shape == (self.height, self.width, 3)
        self._chk(self.lib.ygz_hip_upload_bgr(self._ctx, slot, _p(bgr, C.c_uint8), self.width * 3), "upload_bgr")

    # ---- bulk traffic (arrays must be C-contiguous; wait=False needs page-locked memory, see PinnedArray)
    def upload_bgr_batch(self, slot_begin, bgr, wait=True):
        assert bgr.dtype == np.uint8 and bgr.flags["C_CONTIGUOUS"] and bgr.shape[1:] == (self.height, self.width, 3)
        self._chk(self.lib.ygz_hip_upload_bgr_batch(self._ctx, slot_begin, len(bgr), _p(bgr, C.c_uint8), int(wait)), "upload_bgr_batch")

    def upload_gray_batch(self, slot_begin, gray, wait=True):
        assert gray.dtype == np.uint8 and gray.flags["C_CONTIGUOUS"] and gray.shape[1:] == (self.height, self.width)
        self._chk(self.lib.ygz_hip_upload_gray_batch(self._ctx, slot_begin, len(gray), _p(gray, C.c_uint8), int(wait)), "upload_gray_batch")

    def get_keypoint_pixels_batch(self, slot_begin, n_slots, px=None, count=None, wait=True):
        px = np.empty((n_slots, self.cells, 2), np.float64) if px is None else px
        count = np.empty(n_slots, np.int32) if count is None else count
        self._chk(self.lib.ygz_hip_get_keypoint_pixels_batch(self._ctx, slot_begin, n_slots, _p(px, C.c_double), _p(count, C.c_int32), int(wait)),
                  "get_keypoint_pixels_batch")
        return px, count

    def get_keypoints_batch(self, slot_begin, n_slots, out=None, wait=True):
        if out is None:
            out = dict(px=np.empty((n_slots, self.cells, 2), np.float64), level=np.empty((n_slots, self.cells), np.int32),
                       score=np.empty((n_slots, self.cells), np.float32), angle=np.empty((n_slots, self.cells), np.float32),
                       desc=np.empty((n_slots, self.cells, 32), np.uint8), count=np.empty(n_slots, np.int32))
        f = lambda k, t: _p(out[k], t) if out.get(k) is not None else None
        soa = KptSoa(f("px", C.c_double), f("level", C.c_int32), f("score", C.c_float), f("angle", C.c_float), f("desc", C.c_uint8))
        self._chk(self.lib.ygz_hip_get_keypoints_batch(self._ctx, slot_begin, n_slots, C.byref(soa), _p(out["count"], C.c_int32), int(wait)),
                  "get_keypoints_batch")
        return out

    def set_keypoint_depths_batch(self, slot_begin, depth, has_mp, wait=True):
        assert depth.dtype == np.float64 and has_mp.dtype == np.uint8 and depth.shape[1] == self.cells and has_mp.shape == depth.shape
        self._chk(self.lib.ygz_hip_set_keypoint_depths_batch(self._ctx, slot_begin, len(depth), _p(depth, C.c_double), _p(has_mp, C.c_uint8),
                                                             int(wait)), "set_keypoint_depths_batch")

    def get_keypoint_counts(self, slot_begin, n_slots, out=None, wait=True):
        out = np.empty(n_slots, np.int32) if out is None else out
        self._chk(self.lib.ygz_hip_get_keypoint_counts(self._ctx, slot_begin, n_slots, _p(out, C.c_int32), int(wait)), "get_keypoint_counts")
        return out

    def upload_depth_batch(self, slot_begin, depth, scale=1.0, wait=True):
        """depth [n, dh, dw]: float32 / float64 metres, or uint16 with depth = value * scale"""
        assert depth.ndim == 3 and depth.flags["C_CONTIGUOUS"] and depth.dtype in (np.float32, np.uint16, np.float64)
        kind = {np.dtype(np.float32): 0, np.dtype(np.uint16): 1, np.dtype(np.float64): 2}[depth.dtype]
        self._chk(self.lib.ygz_hip_upload_depth_batch(self._ctx, slot_begin, len(depth), C.c_void_p(depth.ctypes.data), depth.shape[2], depth.shape[1],
                                                      kind, C.c_double(scale), int(wait)), "upload_depth_batch")

    def get_keypoint_depths(self, slot):
        d, m, n = np.empty(self.cells, np.float64), np.empty(self.cells, np.uint8), C.c_int(0)
        self._chk(self.lib.ygz_hip_get_keypoint_depths(self._ctx, slot, _p(d, C.c_double), _p(m, C.c_uint8), self.cells, C.byref(n)), "get_keypoint_depths")
        return d[:n.value].copy(), m[:n.value].copy()

    def keypoint_depths_from_image(self, slot_begin, n_slots):
        self._chk(self.lib.ygz_hip_keypoint_depths_from_image(self._ctx, slot_begin, n_slots), "keypoint_depths_from_image")

    # ---- keyframe store / device-built BA windows (configs[4])
    def stream_wait(self, signaler):
        self._chk(self.lib.ygz_hip_stream_wait(self._ctx, signaler._ctx), "stream_wait")

    def mark(self):
        self._chk(self.lib.ygz_hip_mark(self._ctx), "mark")

    def wait_mark(self, signaler):
        self._chk(self.lib.ygz_hip_wait_mark(self._ctx, signaler._ctx), "wait_mark")

    def kf_row_bytes(self, with_images=False):
        return int(self.lib.ygz_hip_kf_row_bytes(self._ctx, int(with_images)))

    def kf_store_create(self, n_keyframes, n_frames, max_windows, rows_ptr=None, rows_bytes=0, with_images=False):
        """with_images: the rows also hold the keyframes' pyramids (ba_build_windows(..., obs_mode=1) reads them)"""
        self._chk(self.lib.ygz_hip_kf_store_create(self._ctx, n_keyframes, n_frames, max_windows, C.c_void_p(rows_ptr) if rows_ptr else None,
                                                   C.c_size_t(rows_bytes), int(with_images)), "kf_store_create")

    def kf_store_info(self):
        rows, trel, rb, nk, nf = C.c_void_p(0), C.c_void_p(0), C.c_size_t(0), C.c_int(0), C.c_int(0)
        self._chk(self.lib.ygz_hip_kf_store_info(self._ctx, C.byref(rows), C.byref(rb), C.byref(trel), C.byref(nk), C.byref(nf)), "kf_store_info")
        return dict(rows=rows.value, row_bytes=rb.value, trel=trel.value, n_keyframes=nk.value, n_frames=nf.value)

    def kf_store_put(self, src, src_slots, kf_index):
        a = np.ascontiguousarray(src_slots, np.int32); b = np.ascontiguousarray(kf_index, np.int32)
        assert len(a) == len(b)
        self._chk(self.lib.ygz_hip_kf_store_put(self._ctx, src._ctx, len(a), _p(a, C.c_int32), _p(b, C.c_int32)), "kf_store_put")

    def kf_store_put_trel(self, src, first_pair, n_pairs, first_frame):
        self._chk(self.lib.ygz_hip_kf_store_put_trel(self._ctx, src._ctx, first_pair, n_pairs, first_frame), "kf_store_put_trel")

    def kf_store_set_trel(self, first_frame, T_rel):
        T = np.ascontiguousarray(T_rel, np.float64).reshape(-1, 7)
        self._chk(self.lib.ygz_hip_kf_store_set_trel(self._ctx, first_frame, len(T), _p(T, C.c_double)), "kf_store_set_trel")

    def kf_store_refresh(self):
        self._chk(self.lib.ygz_hip_kf_store_refresh(self._ctx), "kf_store_refresh")

    def ba_reserve_windows(self, window_begin, n_windows, K, max_points, huber_delta=5.991):
        self._chk(self.lib.ygz_hip_ba_reserve_windows(self._ctx, window_begin, n_windows, K, max_points, C.c_double(huber_delta)), "ba_reserve_windows")

    def ba_build_windows(self, window_begin, kf_index, kf_frame, n_kfs, obs_mode=0):
        """kf_index / kf_frame [n, K] (entries past n_kfs[i] ignored), n_kfs [n]; obs_mode 0: observations = good Hamming matches, 1: direct
        projection of the map points into the keyframes (FindCandidates + FindDirectProjection; the store must hold the images)"""
        a = np.ascontiguousarray(kf_index, np.int32); b = np.ascontiguousarray(kf_frame, np.int32); c = np.ascontiguousarray(n_kfs, np.int32)
        assert a.shape == b.shape and a.ndim == 2 and len(c) == len(a)
        self._chk(self.lib.ygz_hip_ba_build_windows(self._ctx, window_begin, len(a), _p(a, C.c_int32), _p(b, C.c_int32), _p(c, C.c_int32), int(obs_mode)),
                  "ba_build_windows")

    def ba_mark_outliers(self, window_begin, n_windows, chi2_threshold=5.991, disable=False):
        """BA.cpp:503-515 on the windows' current state (asynchronous, behind the resident LM)"""
        self._chk(self.lib.ygz_hip_ba_mark_outliers(self._ctx, window_begin, n_windows, C.c_double(chi2_threshold), int(disable)), "ba_mark_outliers")

    def ba_get_outlier_stats(self, window_begin, n_windows):
        """[n, 4]: edges tested, outliers, chi2 of the tested edges, chi2 of the inliers"""
        out = np.empty((n_windows, 4), np.float64)
        self._chk(self.lib.ygz_hip_ba_get_outlier_stats(self._ctx, window_begin, n_windows, _p(out, C.c_double)), "ba_get_outlier_stats")
        return out

    def ba_pack_states(self, window_begin, n_windows, row_doubles, dst_ptr=None, out=None, wait=True):
        """state rows into device memory at dst_ptr, or into (and returning) a host array [n_windows, row_doubles]"""
        if dst_ptr is not None:
            self._chk(self.lib.ygz_hip_ba_pack_states(self._ctx, window_begin, n_windows, C.c_void_p(dst_ptr), C.c_size_t(row_doubles), 1, int(wait)), "ba_pack_states")
            return None
        out = np.empty((n_windows, row_doubles), np.float64) if out is None else out
        self._chk(self.lib.ygz_hip_ba_pack_states(self._ctx, window_begin, n_windows, C.c_void_p(out.ctypes.data), C.c_size_t(row_doubles), 0, int(wait)), "ba_pack_states")
        return out

    def ba_get_stats(self, window_begin, n_windows, want_stats=True):
        """(statistics of the last resident LM run per window, dims [n, 4] = poses, points, edges, free poses)"""
        st = (BaStats * n_windows)()
        dims = np.empty((n_windows, 4), np.int32)
        self._chk(self.lib.ygz_hip_ba_get_stats(self._ctx, window_begin, n_windows, st if want_stats else None, _p(dims, C.c_int32)), "ba_get_stats")
        return (list(st) if want_stats else None), dims

    def ba_lm_iterations(self, window_begin, n_windows):
        """iterations of the last resident LM run per window WITHOUT raising: < 0 = the team timed out at a barrier (or no run yet)"""
        st = (BaStats * n_windows)()
        rc = self.lib.ygz_hip_ba_get_stats(self._ctx, window_begin, n_windows, st, None)
        if rc not in (OK, E_HIP, E_STATE):
            self._chk(rc, "ba_get_stats")
        return [s.iterations for s in st]

    def track_get_summary(self, out=None, wait=True):
        if out is None:
            out = np.empty((self.max_frames, SUMMARY_FIELDS), np.float64)
        n = C.c_int(0)
        self._chk(self.lib.ygz_hip_track_get_summary(self._ctx, _p(out, C.c_double), len(out), C.byref(n), int(wait)), "track_get_summary")
        return out[:n.value]

    def build_pyramid(self, slot_begin=0, n_slots=1, from_bgr=False):
        self._chk(self.lib.ygz_hip_build_pyramid(self._ctx, slot_begin, n_slots, int(from_bgr)), "build_pyramid")

    def level_size(self, level):
        w, h = C.c_int(0), C.c_int(0)
        self._chk(self.lib.ygz_hip_level_size(self._ctx, level, C.byref(w), C.byref(h)), "level_size")
        return w.value, h.value

    def download_level(self, slot, level):
        w, h = self.level_size(level)
        out = np.empty((h, w), np.uint8)
        self._chk(self.lib.ygz_hip_download_level(self._ctx, slot, level, _p(out, C.c_uint8)), "download_level")
        return out

    # ---- extractor
    def download_framed_level(self, slot, level):
        """the tracker's working image of a level (24-pixel reflect-101 frame), or None when the slot has no current framed copy"""
        w, h = self.level_size(level)
        out = np.empty((h + 48, w + 48), np.uint8)
        rc = self.lib.ygz_hip_download_framed_level(self._ctx, slot, level, _p(out, C.c_uint8))
        if rc == E_STATE:
            return None
        self._chk(rc, "download_framed_level")
        return out

    def detect(self, slot_begin=0, n_slots=1, occupied=None):
        occ = None
        if occupied is not None:
            occupied = np.ascontiguousarray(occupied, np.uint8).reshape(n_slots, self.cells)
            occ = _p(occupied, C.c_uint8)
        self._chk(self.lib.ygz_hip_detect(self._ctx, slot_begin, n_slots, occ), "detect")

    def keypoint_count(self, slot):
        n = C.c_int(0)
        self._chk(self.lib.ygz_hip_keypoint_count(self._ctx, slot, C.byref(n)), "keypoint_count")
        return n.value

    def get_keypoints(self, slot):
        cap = self.cells
        px = np.empty((cap, 2), np.float64)
        level = np.empty(cap, np.int32)
        score = np.empty(cap, np.float32)
        angle = np.empty(cap, np.float32)
        desc = np.empty((cap, 32), np.uint8)
        soa = KptSoa(_p(px, C.c_double), _p(level, C.c_int32), _p(score, C.c_float), _p(angle, C.c_float), _p(desc, C.c_uint8))
        n = C.c_int(0)
        self._chk(self.lib.ygz_hip_get_keypoints(self._ctx, slot, C.byref(soa), cap, C.byref(n)), "get_keypoints")
        n = n.value
        return dict(px=px[:n].copy(), level=level[:n].copy(), score=score[:n].copy(), angle=angle[:n].copy(), desc=desc[:n].copy())

    def describe(self, slot, px, level):
        px = np.ascontiguousarray(px, np.float64).reshape(-1, 2)
        level = np.ascontiguousarray(level, np.int32)
        self._chk(self.lib.ygz_hip_describe(self._ctx, slot, _p(px, C.c_double), _p(level, C.c_int32), len(level)), "describe")

    def describe_given_angle(self, slot, px, level, angle):
        """ComputeOrbDescriptor alone with the caller's angles (degrees) -- what FeatureDetector::ComputeDescriptor(Feature*) calls"""
        px = np.ascontiguousarray(px, np.float64).reshape(-1, 2)
        level = np.ascontiguousarray(level, np.int32)
        angle = np.ascontiguousarray(angle, np.float32)
        self._chk(self.lib.ygz_hip_describe_given_angle(self._ctx, slot, _p(px, C.c_double), _p(level, C.c_int32), _p(angle, C.c_float), len(level)),
                  "describe_given_angle")

    def get_device(self):
        dev, cus = C.c_int(-1), C.c_int(0)
        self._chk(self.lib.ygz_hip_get_device(self._ctx, C.byref(dev), C.byref(cus)), "get_device")
        return dev.value, cus.value

    def get_fast_maps(self, slot, level):
        w, h = self.level_size(level)
        score = np.empty((h, w), np.uint8)
        nms = np.empty((h, w), np.uint8)
        self._chk(self.lib.ygz_hip_get_fast_maps(self._ctx, slot, level, _p(score, C.c_uint8), _p(nms, C.c_uint8)), "get_fast_maps")
        return score, nms

    # ---- matcher
    def match_slots(self, query_slots, train_slots, cross_check=1):
        q = np.ascontiguousarray(query_slots, np.int32)
        t = np.ascontiguousarray(train_slots, np.int32)
        self._chk(self.lib.ygz_hip_match_slots(self._ctx, _p(q, C.c_int32), _p(t, C.c_int32), len(q), cross_check), "match_slots")

    def match_slots_again(self, cross_check=1):
        self._chk(self.lib.ygz_hip_match_slots_again(self._ctx, cross_check), "match_slots_again")

    def get_matches(self, pair):
        idx = np.empty(self.cells, np.int32)
        dist = np.empty(self.cells, np.int32)
        n = C.c_int(0)
        self._chk(self.lib.ygz_hip_get_matches(self._ctx, pair, _p(idx, C.c_int32), _p(dist, C.c_int32), self.cells, C.byref(n)), "get_matches")
        return idx[:n.value].copy(), dist[:n.value].copy()

    def hamming_match(self, q, t, cross_check=1, want_second=False):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        idx = np.empty(len(q), np.int32)
        dist = np.empty(len(q), np.int32)
        d2 = np.empty(len(q), np.int32) if want_second else None
        self._chk(self.lib.ygz_hip_hamming_match(self._ctx, _p(q, C.c_uint8), len(q), _p(t, C.c_uint8), len(t), cross_check,
                                                 _p(idx, C.c_int32), _p(dist, C.c_int32),
                                                 _p(d2, C.c_int32) if want_second else None), "hamming_match")
        return (idx, dist, d2) if want_second else (idx, dist)

    # ---- M3 / M6 filters
    def match_postfilter(self, min_floor=20.0, min_ceil=50.0, factor=3.0):
        self._chk(self.lib.ygz_hip_match_postfilter(self._ctx, C.c_double(min_floor), C.c_double(min_ceil), C.c_double(factor)), "match_postfilter")

    def get_good_matches(self, pair):
        good = np.zeros(self.cells, np.uint8)
        n, ng, md = C.c_int(0), C.c_int(0), C.c_double(0)
        self._chk(self.lib.ygz_hip_get_good_matches(self._ctx, pair, _p(good, C.c_uint8), self.cells, C.byref(n), C.byref(ng), C.byref(md)),
                  "get_good_matches")
        return good[:n.value].astype(bool), ng.value, md.value

    def match_postfilter_host(self, train_idx, dist, min_floor=20.0, min_ceil=50.0, factor=3.0):
        ti = np.ascontiguousarray(train_idx, np.int32); di = np.ascontiguousarray(dist, np.int32)
        good = np.zeros(max(len(ti), 1), np.uint8)
        ng, md = C.c_int(0), C.c_double(0)
        self._chk(self.lib.ygz_hip_match_postfilter_host(self._ctx, _p(ti, C.c_int32), _p(di, C.c_int32), len(ti), C.c_double(min_floor),
                                                         C.c_double(min_ceil), C.c_double(factor), _p(good, C.c_uint8), C.byref(ng), C.byref(md)),
                  "match_postfilter_host")
        return good[:len(ti)].astype(bool), ng.value, md.value

    def match_sets(self, descs, pair_q, pair_t, cross_check=1, good_filter=True, min_floor=20.0, min_ceil=50.0, factor=3.0):
        """M1-M3 over host descriptor sets in ONE call: descs = list of [n_s, 32] uint8 arrays; returns per pair (rows of the query set)
        train_idx, dist and -- with good_filter -- the kept flags, their count and the clamped minimum distance"""
        ds = [np.ascontiguousarray(d, np.uint8).reshape(-1, 32) for d in descs]
        n_sets, n_pairs = len(ds), len(pair_q)
        ptrs = (C.POINTER(C.c_uint8) * n_sets)(*[_p(d, C.c_uint8) for d in ds])
        cnt = np.array([len(d) for d in ds], np.int32)
        pq = np.ascontiguousarray(pair_q, np.int32); pt = np.ascontiguousarray(pair_t, np.int32)
        idx = np.empty((n_pairs, self.cells), np.int32); dist = np.empty((n_pairs, self.cells), np.int32)
        good = np.zeros((n_pairs, self.cells), np.uint8) if good_filter else None
        ng = np.zeros(n_pairs, np.int32); md = np.zeros(n_pairs, np.float64)
        self.lib.ygz_hip_match_sets.argtypes = None
        self._chk(self.lib.ygz_hip_match_sets(self._ctx, n_sets, ptrs, _p(cnt, C.c_int32), n_pairs, _p(pq, C.c_int32), _p(pt, C.c_int32), cross_check,
                                              _p(idx, C.c_int32), _p(dist, C.c_int32), _p(good, C.c_uint8) if good_filter else None,
                                              _p(ng, C.c_int32) if good_filter else None, _p(md, C.c_double) if good_filter else None,
                                              C.c_double(min_floor), C.c_double(min_ceil), C.c_double(factor)), "match_sets")
        out = []
        for p in range(n_pairs):
            n = int(cnt[pq[p]])
            r = dict(idx=idx[p, :n], dist=dist[p, :n])
            if good_filter:
                r.update(good=good[p, :n].astype(bool), n_good=int(ng[p]), min_dis=float(md[p]))
            out.append(r)
        return out

    def check_frame_descriptors(self, slot1, slot2, idx1, idx2, init_low=30, init_high=80, ratio=3.0):
        i1 = np.ascontiguousarray(idx1, np.int32); i2 = np.ascontiguousarray(idx2, np.int32)
        n = len(i1)
        dist = np.zeros(max(n, 1), np.int32); keep = np.zeros(max(n, 1), np.uint8)
        ng, best = C.c_int(0), C.c_int(0)
        self._chk(self.lib.ygz_hip_check_frame_descriptors(self._ctx, slot1, slot2, _p(i1, C.c_int32), _p(i2, C.c_int32), n, init_low, init_high,
                                                           C.c_float(ratio), _p(dist, C.c_int32), _p(keep, C.c_uint8), C.byref(ng), C.byref(best)),
                  "check_frame_descriptors")
        return dist[:n].copy(), keep[:n].astype(bool), ng.value, best.value

    def check_descriptor_pairs(self, desc1, desc2, init_low=30, init_high=80, ratio=3.0):
        d1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32); d2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
        n = len(d1)
        dist = np.zeros(max(n, 1), np.int32); keep = np.zeros(max(n, 1), np.uint8)
        ng, best = C.c_int(0), C.c_int(0)
        self._chk(self.lib.ygz_hip_check_descriptor_pairs(self._ctx, _p(d1, C.c_uint8), _p(d2, C.c_uint8), n, init_low, init_high, C.c_float(ratio),
                                                          _p(dist, C.c_int32), _p(keep, C.c_uint8), C.byref(ng), C.byref(best)),
                  "check_descriptor_pairs")
        return dist[:n].copy(), keep[:n].astype(bool), ng.value, best.value

    # ---- alignment
    def sparse_align_residuals(self, ref_slot, cur_slot, T_cur_from_ref, px, depth, has_mp, level):
        """one SparseImgAlign::computeResiduals(model, linearize = true) at `level`: (chi2 float sum, n_meas, H [6][6], Jres [6])"""
        T = np.ascontiguousarray(T_cur_from_ref, np.float64)
        px = np.ascontiguousarray(px, np.float64).reshape(-1, 2); depth = np.ascontiguousarray(depth, np.float64)
        has_mp = np.ascontiguousarray(has_mp, np.uint8)
        chi2, nm = C.c_double(0), C.c_int(0)
        H, J = np.zeros(36), np.zeros(6)
        self.lib.ygz_hip_sparse_align_residuals.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                                            C.POINTER(C.c_uint8), C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int),
                                                            C.POINTER(C.c_double), C.POINTER(C.c_double)]
        self._chk(self.lib.ygz_hip_sparse_align_residuals(self._ctx, int(ref_slot), int(cur_slot), _p(T, C.c_double), _p(px, C.c_double), _p(depth, C.c_double),
                                                          _p(has_mp, C.c_uint8), len(depth), int(level), C.byref(chi2), C.byref(nm), _p(H, C.c_double),
                                                          _p(J, C.c_double)), "sparse_align_residuals")
        return chi2.value, nm.value, H.reshape(6, 6), J

    def find_direct_projection_mp(self, cur_slot, T_cur, kf_slots, kf_T, cand_kf, pos_world, px_ref, level_ref, px_in=None):
        """Matcher::FindDirectProjection, MapPoint overload, per candidate over several reference keyframes in one launch.
        Returns dict(in_view, px_proj, ok, px, level); px_in given: no FindCandidates projection, every candidate evaluated."""
        T_cur = np.ascontiguousarray(T_cur, np.float64); kf_slots = np.ascontiguousarray(kf_slots, np.int32)
        kf_T = np.ascontiguousarray(kf_T, np.float64).reshape(-1, 7); cand_kf = np.ascontiguousarray(cand_kf, np.int32)
        pos_world = np.ascontiguousarray(pos_world, np.float64).reshape(-1, 3); px_ref = np.ascontiguousarray(px_ref, np.float64).reshape(-1, 2)
        level_ref = np.ascontiguousarray(level_ref, np.int32)
        n = len(cand_kf)
        vis = np.zeros(n, np.uint8); proj = np.zeros((n, 2)); ok = np.zeros(n, np.uint8); px = np.zeros((n, 2)); sl = np.zeros(n, np.int32)
        pin = None
        if px_in is not None:
            pin = np.ascontiguousarray(px_in, np.float64).reshape(-1, 2)
        self.lib.ygz_hip_find_direct_projection_mp.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_double),
                                                               C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32),
                                                               C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.POINTER(C.c_double), C.POINTER(C.c_uint8),
                                                               C.POINTER(C.c_double), C.POINTER(C.c_int32)]
        self._chk(self.lib.ygz_hip_find_direct_projection_mp(self._ctx, int(cur_slot), _p(T_cur, C.c_double), len(kf_slots), _p(kf_slots, C.c_int32),
                                                             _p(kf_T, C.c_double), n, _p(cand_kf, C.c_int32), _p(pos_world, C.c_double), _p(px_ref, C.c_double),
                                                             _p(level_ref, C.c_int32), None if pin is None else _p(pin, C.c_double), _p(vis, C.c_uint8),
                                                             _p(proj, C.c_double), _p(ok, C.c_uint8), _p(px, C.c_double), _p(sl, C.c_int32)),
                  "find_direct_projection_mp")
        return dict(in_view=vis.astype(bool), px_proj=proj, ok=ok.astype(bool), px=px, level=sl)

    def find_direct_projection_mp_begin(self, cur_slot, T_cur, kf_slots, kf_T, cand_kf, pos_world, px_ref, level_ref):
        """first half of find_direct_projection_mp (px_in = None form): queued, not waited for; returns the candidate count for ..._end"""
        T_cur = np.ascontiguousarray(T_cur, np.float64); kf_slots = np.ascontiguousarray(kf_slots, np.int32)
        kf_T = np.ascontiguousarray(kf_T, np.float64).reshape(-1, 7); cand_kf = np.ascontiguousarray(cand_kf, np.int32)
        pos_world = np.ascontiguousarray(pos_world, np.float64).reshape(-1, 3); px_ref = np.ascontiguousarray(px_ref, np.float64).reshape(-1, 2)
        level_ref = np.ascontiguousarray(level_ref, np.int32)
        self.lib.ygz_hip_find_direct_projection_mp_begin.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_double),
                                                                     C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)]
        self._chk(self.lib.ygz_hip_find_direct_projection_mp_begin(self._ctx, int(cur_slot), _p(T_cur, C.c_double), len(kf_slots), _p(kf_slots, C.c_int32),
                                                                   _p(kf_T, C.c_double), len(cand_kf), _p(cand_kf, C.c_int32), _p(pos_world, C.c_double),
                                                                   _p(px_ref, C.c_double), _p(level_ref, C.c_int32)), "find_direct_projection_mp_begin")
        return len(cand_kf)

    def find_direct_projection_mp_end(self, n):
        vis = np.zeros(n, np.uint8); proj = np.zeros((n, 2)); ok = np.zeros(n, np.uint8); px = np.zeros((n, 2)); sl = np.zeros(n, np.int32)
        self.lib.ygz_hip_find_direct_projection_mp_end.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_double), C.POINTER(C.c_uint8),
                                                                   C.POINTER(C.c_double), C.POINTER(C.c_int32)]
        self._chk(self.lib.ygz_hip_find_direct_projection_mp_end(self._ctx, int(n), _p(vis, C.c_uint8), _p(proj, C.c_double), _p(ok, C.c_uint8),
                                                                 _p(px, C.c_double), _p(sl, C.c_int32)), "find_direct_projection_mp_end")
        return dict(in_view=vis.astype(bool), px_proj=proj, ok=ok.astype(bool), px=px, level=sl)

    def find_direct_projection(self, ref_slot, T_ref, cur_slot, T_cur, px_ref, depth_ref, level_ref, px_cur):
        pair = AlignPair(ref_slot, cur_slot, (C.c_double * 7)(*T_ref), (C.c_double * 7)(*T_cur))
        px_ref = np.ascontiguousarray(px_ref, np.float64).reshape(-1, 2)
        depth_ref = np.ascontiguousarray(depth_ref, np.float64)
        level_ref = np.ascontiguousarray(level_ref, np.int32)
        px_cur = np.ascontiguousarray(px_cur, np.float64).reshape(-1, 2).copy()
        n = len(depth_ref)
        sl = np.empty(n, np.int32)
        ok = np.empty(n, np.uint8)
        self._chk(self.lib.ygz_hip_find_direct_projection(self._ctx, C.byref(pair), _p(px_ref, C.c_double), _p(depth_ref, C.c_double),
                                                          _p(level_ref, C.c_int32), _p(px_cur, C.c_double), _p(sl, C.c_int32),
                                                          _p(ok, C.c_uint8), n), "find_direct_projection")
        return ok.astype(bool), px_cur, sl

    def track_local_map(self, cur_slot, T_cur, kf_slot, kf_T, pos_world, point_bad, cand_point, cand_kf, cand_px_ref, cand_level):
        """LocalMapping::FindCandidates + ProjectMapPoints (LocalMapping.cpp:47-120) -> n, in_view, px_proj, match_cand, px_match, level"""
        pw = np.ascontiguousarray(pos_world, np.float64).reshape(-1, 3)
        P = pw.shape[0]
        bad = None if point_bad is None else np.ascontiguousarray(point_bad, np.uint8)
        ks = np.ascontiguousarray(kf_slot, np.int32); kT = np.ascontiguousarray(kf_T, np.float64).reshape(-1, 7)
        cp = np.ascontiguousarray(cand_point, np.int32); ck = np.ascontiguousarray(cand_kf, np.int32)
        cl = np.ascontiguousarray(cand_level, np.int32); cx = np.ascontiguousarray(cand_px_ref, np.float64).reshape(-1, 2)
        m = LocalMap(P, _p(pw, C.c_double), _p(bad, C.c_uint8) if bad is not None else None, len(ks), _p(ks, C.c_int32), _p(kT, C.c_double),
                     len(cp), _p(cp, C.c_int32), _p(ck, C.c_int32), _p(cl, C.c_int32), _p(cx, C.c_double))
        in_view = np.zeros(P, np.uint8); px_proj = np.zeros((P, 2)); match = np.full(P, -1, np.int32)
        px_match = np.zeros((P, 2)); lvl = np.zeros(P, np.int32)
        n = C.c_int32(0)
        T = (C.c_double * 7)(*[float(x) for x in T_cur])
        self.lib.ygz_hip_track_local_map.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(LocalMap), C.POINTER(C.c_uint8),
                                                     C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_int32),
                                                     C.POINTER(C.c_int32)]
        self._chk(self.lib.ygz_hip_track_local_map(self._ctx, int(cur_slot), T, C.byref(m), _p(in_view, C.c_uint8), _p(px_proj, C.c_double),
                                                   _p(match, C.c_int32), _p(px_match, C.c_double), _p(lvl, C.c_int32), C.byref(n)), "track_local_map")
        return int(n.value), in_view, px_proj, match, px_match, lvl

    def align2d(self, cur_slot, level, pwb, uv, n_iter=10):
        pwb = np.ascontiguousarray(pwb, np.uint8).reshape(-1, 100)
        uv = np.ascontiguousarray(uv, np.float64).reshape(-1, 2).copy()
        n = len(pwb)
        ok = np.empty(n, np.uint8)
        chi2 = np.empty(n, np.float32)
        patch = np.ascontiguousarray(pwb.reshape(n, 10, 10)[:, 1:9, 1:9]).reshape(n, 64)
        self._chk(self.lib.ygz_hip_align2d(self._ctx, cur_slot, level, _p(pwb, C.c_uint8), _p(patch, C.c_uint8), _p(uv, C.c_double),
                                           _p(ok, C.c_uint8), _p(chi2, C.c_float), n, n_iter), "align2d")
        return ok.astype(bool), uv, chi2

    def sparse_align(self, ref_slot, T_ref, cur_slot, T_cur, px, depth, has_mp, max_level=2, min_level=0, n_iter=30):
        Tr = (C.c_double * 7)(*T_ref)
        Tc = (C.c_double * 7)(*T_cur)
        px = np.ascontiguousarray(px, np.float64).reshape(-1, 2)
        depth = np.ascontiguousarray(depth, np.float64)
        has_mp = np.ascontiguousarray(has_mp, np.uint8)
        nm = C.c_int(0)
        iters = (C.c_int * MAX_LEVELS)()
        self._chk(self.lib.ygz_hip_sparse_align(self._ctx, ref_slot, Tr, cur_slot, Tc, _p(px, C.c_double), _p(depth, C.c_double),
                                                _p(has_mp, C.c_uint8), len(depth), max_level, min_level, n_iter, C.byref(nm), iters),
                  "sparse_align")
        return nm.value, np.array(list(Tc)), list(iters)[:self.levels]

    # ---- KLT
    def klt_params(self):
        p = KltParams()
        self.lib.ygz_hip_default_klt_params(C.byref(p))
        return p

    def klt_track(self, prev_slot, cur_slot, prev_pts, next_pts_init, params=None):
        prm = params or self.klt_params()
        pp = np.ascontiguousarray(prev_pts, np.float32).reshape(-1, 2)
        npts = np.ascontiguousarray(next_pts_init, np.float32).reshape(-1, 2).copy()
        st = np.empty(len(pp), np.uint8)
        err = np.empty(len(pp), np.float32)
        self._chk(self.lib.ygz_hip_klt_track(self._ctx, prev_slot, cur_slot, _p(pp, C.c_float), _p(npts, C.c_float), len(pp),
                                             C.byref(prm), _p(st, C.c_uint8), _p(err, C.c_float)), "klt_track")
        return npts, st, err

    def klt_track_filtered(self, prev_slot, cur_slot, prev_pts, next_pts_init, border=20, params=None):
        prm = params or self.klt_params()
        pp = np.ascontiguousarray(prev_pts, np.float32).reshape(-1, 2)
        npts = np.ascontiguousarray(next_pts_init, np.float32).reshape(-1, 2).copy()
        n = len(pp)
        st = np.empty(max(n, 1), np.uint8); err = np.empty(max(n, 1), np.float32); keep = np.zeros(max(n, 1), np.uint8); nk = C.c_int(0)
        self._chk(self.lib.ygz_hip_klt_track_filtered(self._ctx, prev_slot, cur_slot, _p(pp, C.c_float), _p(npts, C.c_float), n, C.byref(prm),
                                                      int(border), _p(st, C.c_uint8), _p(err, C.c_float), _p(keep, C.c_uint8), C.byref(nk)),
                  "klt_track_filtered")
        return npts, st[:n], err[:n], keep[:n].astype(bool), nk.value

    # ---- resident batched tracking
    def set_keypoint_depths(self, slot, depth, has_mp):
        depth = np.ascontiguousarray(depth, np.float64)
        has_mp = np.ascontiguousarray(has_mp, np.uint8)
        self._chk(self.lib.ygz_hip_set_keypoint_depths(self._ctx, slot, _p(depth, C.c_double), _p(has_mp, C.c_uint8), len(depth)),
                  "set_keypoint_depths")

    def track_begin(self, cur_slots, ref_slots, T_cur, T_ref, predict=True):
        c = np.ascontiguousarray(cur_slots, np.int32)
        r = np.ascontiguousarray(ref_slots, np.int32)
        Tc = np.ascontiguousarray(T_cur, np.float64).reshape(-1, 7)
        Tr = np.ascontiguousarray(T_ref, np.float64).reshape(-1, 7)
        self._chk(self.lib.ygz_hip_track_begin(self._ctx, _p(c, C.c_int32), _p(r, C.c_int32), _p(Tc, C.c_double), _p(Tr, C.c_double),
                                               len(c), int(predict)), "track_begin")

    def track_reload(self, predict=True):
        self._chk(self.lib.ygz_hip_track_reload(self._ctx, int(predict)), "track_reload")

    def track_klt_prepare(self):
        self._chk(self.lib.ygz_hip_track_klt_prepare(self._ctx), "track_klt_prepare")

    def track_klt(self, params=None):
        prm = params or self.klt_params()
        self._chk(self.lib.ygz_hip_track_klt(self._ctx, C.byref(prm)), "track_klt")

    def track_direct(self):
        self._chk(self.lib.ygz_hip_track_direct(self._ctx), "track_direct")

    def track_sparse_align(self, max_level=2, min_level=0, n_iter=30):
        self._chk(self.lib.ygz_hip_track_sparse_align(self._ctx, max_level, min_level, n_iter), "track_sparse_align")

    def track_adopt_pose(self):
        self._chk(self.lib.ygz_hip_track_adopt_pose(self._ctx), "track_adopt_pose")

    def track_pose_only(self):
        self._chk(self.lib.ygz_hip_track_pose_only(self._ctx), "track_pose_only")

    def track_get_pose_only(self, pair):
        pose = (C.c_double * 6)(); T = (C.c_double * 7)()
        inl, rnd, n = C.c_int(0), C.c_int(0), C.c_int(0)
        bad = np.zeros(self.cells, np.uint8); depth = np.zeros(self.cells, np.float64)
        self._chk(self.lib.ygz_hip_track_get_pose_only(self._ctx, pair, pose, T, C.byref(inl), C.byref(rnd), _p(bad, C.c_uint8),
                                                       _p(depth, C.c_double), self.cells, C.byref(n)), "track_get_pose_only")
        return dict(pose=np.array(list(pose)), T=np.array(list(T)), inliers=inl.value, rounds=rnd.value,
                    bad=bad[:n.value].astype(bool), depth=depth[:n.value].copy())

    def track_get_klt(self, pair):
        pts = np.empty((self.cells, 2), np.float32)
        st = np.empty(self.cells, np.uint8)
        err = np.empty(self.cells, np.float32)
        n = C.c_int(0)
        self._chk(self.lib.ygz_hip_track_get_klt(self._ctx, pair, _p(pts, C.c_float), _p(st, C.c_uint8), _p(err, C.c_float), self.cells,
                                                 C.byref(n)), "track_get_klt")
        return pts[:n.value].copy(), st[:n.value].copy(), err[:n.value].copy()

    def track_get_direct(self, pair):
        px = np.empty((self.cells, 2), np.float64)
        lvl = np.empty(self.cells, np.int32)
        ok = np.empty(self.cells, np.uint8)
        n = C.c_int(0)
        self._chk(self.lib.ygz_hip_track_get_direct(self._ctx, pair, _p(px, C.c_double), _p(lvl, C.c_int32), _p(ok, C.c_uint8), self.cells,
                                                    C.byref(n)), "track_get_direct")
        return ok[:n.value].astype(bool), px[:n.value].copy(), lvl[:n.value].copy()

    def track_get_pose(self, pair):
        T = (C.c_double * 7)()
        nm = C.c_int(0)
        iters = (C.c_int * MAX_LEVELS)()
        self._chk(self.lib.ygz_hip_track_get_pose(self._ctx, pair, T, C.byref(nm), iters), "track_get_pose")
        return nm.value, np.array(list(T)), list(iters)[:self.levels]

    def depth_filter_update(self, cur_slot, T_cur, ref_slot, T_refs, seeds, batch_counter, max_n_kfs=5, conv_thresh=100.0):
        """seeds: dict of kp [n,2] f32, octave, ref (index into ref_slot), frame_id (u64), a, b, mu, z_range, sigma2 (f32); a/b/mu/sigma2 come
        back updated (copies) together with state, z, matched_px, pos_world"""
        rs = np.ascontiguousarray(ref_slot, np.int32); Tr = np.ascontiguousarray(T_refs, np.float64).reshape(-1, 7)
        kp = np.ascontiguousarray(seeds["kp"], np.float32).reshape(-1, 2); n = len(kp)
        oc = np.ascontiguousarray(seeds["octave"], np.int32); sr = np.ascontiguousarray(seeds["ref"], np.int32)
        fid = np.ascontiguousarray(seeds["frame_id"], np.uint64)
        f = {k: np.ascontiguousarray(seeds[k], np.float32).copy() for k in ("a", "b", "mu", "z_range", "sigma2")}
        st = np.zeros(max(n, 1), np.int32); z = np.zeros(max(n, 1)); mp = np.zeros((max(n, 1), 2)); pw = np.zeros((max(n, 1), 3)); nu = C.c_int(0)
        fp = lambda k: _p(f[k], C.c_float)
        self._chk(self.lib.ygz_hip_depth_filter_update(self._ctx, cur_slot, (C.c_double * 7)(*T_cur), len(rs), _p(rs, C.c_int32), _p(Tr, C.c_double),
                                                       int(batch_counter), int(max_n_kfs), C.c_double(conv_thresh), n, _p(kp, C.c_float), _p(oc, C.c_int32),
                                                       _p(sr, C.c_int32), _p(fid, C.c_uint64), fp("a"), fp("b"), fp("mu"), fp("z_range"), fp("sigma2"),
                                                       _p(st, C.c_int32), _p(z, C.c_double), _p(mp, C.c_double), _p(pw, C.c_double), C.byref(nu)),
                  "depth_filter_update")
        return dict(a=f["a"], b=f["b"], mu=f["mu"], sigma2=f["sigma2"], state=st[:n], z=z[:n], matched_px=mp[:n], pos_world=pw[:n], updated=nu.value)

    def create_map_points(self, slot1, T1, slot2, T2, px1, level1, px2):
        p1 = np.ascontiguousarray(px1, np.float64).reshape(-1, 2); l1 = np.ascontiguousarray(level1, np.int32)
        p2 = np.ascontiguousarray(px2, np.float64).reshape(-1, 2).copy()
        n = len(l1)
        code = np.zeros(max(n, 1), np.int32); d1 = np.zeros(max(n, 1)); d2 = np.zeros(max(n, 1)); pw = np.zeros((max(n, 1), 3))
        sl = np.zeros(max(n, 1), np.int32); nc = C.c_int(0)
        self._chk(self.lib.ygz_hip_create_map_points(self._ctx, slot1, (C.c_double * 7)(*T1), slot2, (C.c_double * 7)(*T2), n, _p(p1, C.c_double),
                                                     _p(l1, C.c_int32), _p(p2, C.c_double), _p(code, C.c_int32), _p(d1, C.c_double),
                                                     _p(d2, C.c_double), _p(pw, C.c_double), _p(sl, C.c_int32), C.byref(nc)), "create_map_points")
        return dict(px2=p2, code=code[:n], depth1=d1[:n], depth2=d2[:n], pos_world=pw[:n], search_level=sl[:n], created=nc.value)

    def depth_from_triangulation(self, T_search_ref, f_ref, f_cur, determinant_th=1e-5):
        T = (C.c_double * 7)(*T_search_ref)
        fr = np.ascontiguousarray(f_ref, np.float64).reshape(-1, 3); fc = np.ascontiguousarray(f_cur, np.float64).reshape(-1, 3)
        n = len(fr)
        d1, d2, ok = np.full(max(n, 1), np.nan), np.full(max(n, 1), np.nan), np.zeros(max(n, 1), np.uint8)
        self.lib.ygz_hip_depth_from_triangulation.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int,
                                                              C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint8)]
        self._chk(self.lib.ygz_hip_depth_from_triangulation(self._ctx, T, _p(fr, C.c_double), _p(fc, C.c_double), n, float(determinant_th),
                                                            _p(d1, C.c_double), _p(d2, C.c_double), _p(ok, C.c_uint8)), "depth_from_triangulation")
        return d1[:n], d2[:n], ok[:n]

    # ---- BoW
    def vocab_load(self, blob):
        buf = np.frombuffer(blob, np.uint8).copy()
        self._chk(self.lib.ygz_hip_vocab_load(self._ctx, _p(buf, C.c_uint8), C.c_size_t(len(buf))), "vocab_load")
        k, L, nn, nw = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self._chk(self.lib.ygz_hip_vocab_info(self._ctx, C.byref(k), C.byref(L), C.byref(nn), C.byref(nw)), "vocab_info")
        return k.value, L.value, nn.value, nw.value

    def compute_bow(self, slot_begin, n_slots, levelsup=4):
        self._chk(self.lib.ygz_hip_compute_bow(self._ctx, slot_begin, n_slots, levelsup), "compute_bow")

    def get_bow(self, slot):
        word, weight, node = np.empty(self.cells, np.int32), np.empty(self.cells), np.empty(self.cells, np.int32)
        n = C.c_int(0)
        self._chk(self.lib.ygz_hip_get_bow(self._ctx, slot, _p(word, C.c_int32), _p(weight, C.c_double), _p(node, C.c_int32), self.cells,
                                           C.byref(n)), "get_bow")
        return word[:n.value].copy(), weight[:n.value].copy(), node[:n.value].copy()

    def bow_transform(self, desc, levelsup=4):
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        n = len(desc)
        word, weight, node = np.empty(max(n, 1), np.int32), np.empty(max(n, 1)), np.empty(max(n, 1), np.int32)
        self._chk(self.lib.ygz_hip_bow_transform(self._ctx, _p(desc, C.c_uint8), n, levelsup, _p(word, C.c_int32), _p(weight, C.c_double),
                                                 _p(node, C.c_int32)), "bow_transform")
        return word[:n], weight[:n], node[:n]

    def search_by_bow_slots(self, slot1, slot2, mode=0, E12=None, th_low=65, knn_ratio=0.7, epipolar_dsqr=1e-4):
        s1 = np.ascontiguousarray(slot1, np.int32); s2 = np.ascontiguousarray(slot2, np.int32)
        n = len(s1)
        m = np.empty((n, self.cells), np.int32); cnt = np.empty(n, np.int32)
        E = None if E12 is None else np.ascontiguousarray(E12, np.float64).reshape(n, 9)
        self.lib.ygz_hip_search_by_bow_slots.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double),
                                                         C.c_int, C.c_float, C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        self._chk(self.lib.ygz_hip_search_by_bow_slots(self._ctx, mode, n, _p(s1, C.c_int32), _p(s2, C.c_int32), None if E is None else _p(E, C.c_double),
                                                       int(th_low), float(knn_ratio), float(epipolar_dsqr), _p(m, C.c_int32), _p(cnt, C.c_int32)),
                  "search_by_bow_slots")
        return m, cnt

    def search_by_bow(self, desc1, node1, desc2, node2, mode=0, px1=None, px2=None, E12=None, th_low=65, knn_ratio=0.7, epipolar_dsqr=1e-4):
        d1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32); d2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
        n1a = np.ascontiguousarray(node1, np.int32); n2a = np.ascontiguousarray(node2, np.int32)
        p1 = None if px1 is None else np.ascontiguousarray(px1, np.float64); p2 = None if px2 is None else np.ascontiguousarray(px2, np.float64)
        E = None if E12 is None else np.ascontiguousarray(E12, np.float64).reshape(9)
        m = np.empty(max(len(d1), 1), np.int32); cnt = C.c_int(0)
        f64 = lambda a: None if a is None else _p(a, C.c_double)
        self.lib.ygz_hip_search_by_bow.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_int,
                                                   C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double),
                                                   C.c_int, C.c_float, C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_int)]
        self._chk(self.lib.ygz_hip_search_by_bow(self._ctx, mode, _p(d1, C.c_uint8), _p(n1a, C.c_int32), f64(p1), len(d1), _p(d2, C.c_uint8),
                                                 _p(n2a, C.c_int32), f64(p2), len(d2), f64(E), int(th_low), float(knn_ratio), float(epipolar_dsqr),
                                                 _p(m, C.c_int32), C.byref(cnt)), "search_by_bow")
        return m[:len(d1)], cnt.value

    def bow_orientation(self, angle1, angle2, match12):
        """Matcher::Options::checkOrientation on a SearchByBoW result: (count, hist [30], maxima [3])"""
        a1 = np.ascontiguousarray(angle1, np.float64); a2 = np.ascontiguousarray(angle2, np.float64); m = np.ascontiguousarray(match12, np.int32)
        kept = C.c_int(0); hist = np.zeros(30, np.int32); ind = np.zeros(3, np.int32)
        self.lib.ygz_hip_bow_orientation.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int32),
                                                     C.POINTER(C.c_int), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        self._chk(self.lib.ygz_hip_bow_orientation(self._ctx, _p(a1, C.c_double), len(a1), _p(a2, C.c_double), len(a2), _p(m, C.c_int32), C.byref(kept),
                                                   _p(hist, C.c_int32), _p(ind, C.c_int32)), "bow_orientation")
        return kept.value, hist, ind

    def bow_orientation_slots(self, slot1, slot2, match12):
        s1 = np.ascontiguousarray(slot1, np.int32); s2 = np.ascontiguousarray(slot2, np.int32); m = np.ascontiguousarray(match12, np.int32)
        n = len(s1)
        assert m.shape == (n, self.cells)
        kept = np.zeros(n, np.int32); hist = np.zeros((n, 30), np.int32); ind = np.zeros((n, 3), np.int32)
        self.lib.ygz_hip_bow_orientation_slots.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                                           C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        self._chk(self.lib.ygz_hip_bow_orientation_slots(self._ctx, n, _p(s1, C.c_int32), _p(s2, C.c_int32), _p(m, C.c_int32), _p(kept, C.c_int32),
                                                         _p(hist, C.c_int32), _p(ind, C.c_int32)), "bow_orientation_slots")
        return kept, hist, ind

    # ---- BA
    def _ba_problem(self, poses, fixed, points, edge_pose, edge_point, obs, huber_delta, formulation, cam=None,
                    point_fixed=None, edge_huber=None, edge_enable=None):
        if fixed is None:
            fixed = np.zeros(len(poses), np.uint8)
        self._keep = [np.ascontiguousarray(poses, np.float64).reshape(-1, 6), np.ascontiguousarray(fixed, np.uint8),
                      np.ascontiguousarray(points, np.float64).reshape(-1, 3), np.ascontiguousarray(edge_pose, np.int32),
                      np.ascontiguousarray(edge_point, np.int32), np.ascontiguousarray(obs, np.float64),
                      None if point_fixed is None else np.ascontiguousarray(point_fixed, np.uint8),
                      None if edge_huber is None else np.ascontiguousarray(edge_huber, np.float64),
                      None if edge_enable is None else np.ascontiguousarray(edge_enable, np.uint8)]
        a = self._keep
        fx, fy, cx, cy = cam if cam else (float(self.params.fx), float(self.params.fy), float(self.params.cx), float(self.params.cy))
        opt = lambda v, t: C.POINTER(t)() if v is None else _p(v, t)
        return BaProblem(len(a[0]), len(a[2]), len(a[3]), _p(a[0], C.c_double), _p(a[1], C.c_uint8), _p(a[2], C.c_double),
                         _p(a[3], C.c_int32), _p(a[4], C.c_int32), _p(a[5], C.c_double), fx, fy, cx, cy, float(huber_delta), formulation,
                         opt(a[6], C.c_uint8), opt(a[7], C.c_double), opt(a[8], C.c_uint8))

    @staticmethod
    def _ba_out(K, P, E):
        return dict(Hpp=np.empty((K, 6, 6)), bp=np.empty((K, 6)), Hll=np.empty((P, 3, 3)), bl=np.empty((P, 3)),
                    Hpl=np.empty((E, 6, 3)), err=np.empty((E, 2)), chi2_edge=np.empty(E), chi2=np.empty(1))

    def ba_linearize(self, poses, fixed, points, edge_pose, edge_point, obs, huber_delta=5.991, formulation=0, cam=None,
                     point_fixed=None, edge_huber=None, edge_enable=None):
        pb = self._ba_problem(poses, fixed, points, edge_pose, edge_point, obs, huber_delta, formulation, cam, point_fixed,
                              edge_huber, edge_enable)
        o = self._ba_out(pb.n_poses, pb.n_points, pb.n_edges)
        d = lambda k: _p(o[k], C.c_double)
        self._chk(self.lib.ygz_hip_ba_linearize(self._ctx, C.byref(pb), d("Hpp"), d("bp"), d("Hll"), d("bl"), d("Hpl"), d("err"),
                                                d("chi2_edge"), d("chi2")), "ba_linearize")
        o["chi2"] = float(o["chi2"][0])
        nb = C.c_int(0)
        self._chk(self.lib.ygz_hip_ba_behind_camera(self._ctx, 1023, C.byref(nb)), "ba_behind_camera")
        o["n_behind"] = nb.value
        return o

    def ceres_options(self, **kw):
        o = CeresOptions()
        self.lib.ygz_hip_ceres_default_options(C.byref(o))
        for k, v in kw.items():
            setattr(o, k, v)
        return o

    def ba_solve_ceres(self, poses, fixed, points, edge_pose, edge_point, obs_n, point_fixed=None, edge_huber=None,
                       edge_enable=None, options=None):
        """ceres::Solve around the GPU linearisation of a formulation-2 problem; returns (poses, points, summary dict)."""
        pb = self._ba_problem(poses, fixed, points, edge_pose, edge_point, obs_n, 0.0, 2, None, point_fixed, edge_huber, edge_enable)
        po = self._keep[0].copy()
        pt = self._keep[2].copy()
        sm = CeresSummary()
        opt = options or self.ceres_options()
        self._chk(self.lib.ygz_hip_ba_solve_ceres(self._ctx, C.byref(pb), _p(po, C.c_double), _p(pt, C.c_double), C.byref(opt),
                                                  C.byref(sm)), "ba_solve_ceres")
        return po, pt, {k: getattr(sm, k) for k, _ in CeresSummary._fields_}

    def ba_solve_ceres_resident(self, window_begin, n_windows, options=None):
        sm = (CeresSummary * n_windows)()
        opt = options or self.ceres_options()
        self._chk(self.lib.ygz_hip_ba_solve_ceres_resident(self._ctx, window_begin, n_windows, C.byref(opt), sm), "ba_solve_ceres_resident")
        return [{k: getattr(s, k) for k, _ in CeresSummary._fields_} for s in sm]

    def optimize_pose_only(self, frame_off, px, pw, poses, depth=None):
        """ba::OptimizeCurrentPoseOnly for a batch of frames; returns (poses, bad, depth, inliers, rounds)."""
        off = np.ascontiguousarray(frame_off, np.int32)
        px = np.ascontiguousarray(px, np.float64).reshape(-1, 2)
        pw = np.ascontiguousarray(pw, np.float64).reshape(-1, 3)
        po = np.ascontiguousarray(poses, np.float64).reshape(-1, 6).copy()
        nf, n = len(off) - 1, len(px)
        bad = np.zeros(max(n, 1), np.uint8)
        dep = np.full(max(n, 1), np.nan) if depth is None else np.ascontiguousarray(depth, np.float64).copy()
        inl, rounds = np.zeros(max(nf, 1), np.int32), np.zeros(max(nf, 1), np.int32)
        self._chk(self.lib.ygz_hip_optimize_pose_only(self._ctx, nf, _p(off, C.c_int32), _p(px, C.c_double), _p(pw, C.c_double),
                                                      _p(po, C.c_double), _p(bad, C.c_uint8), _p(dep, C.c_double),
                                                      _p(inl, C.c_int32), _p(rounds, C.c_int32)), "optimize_pose_only")
        return po, bad[:n], dep[:n], inl[:nf], rounds[:nf]

    def ba_optimize(self, poses, fixed, points, edge_pose, edge_point, obs, iterations=20, huber_delta=5.991, cam=None):
        pb = self._ba_problem(poses, fixed, points, edge_pose, edge_point, obs, huber_delta, 0, cam)
        po = np.ascontiguousarray(poses, np.float64).copy()
        pt = np.ascontiguousarray(points, np.float64).copy()
        st = BaStats()
        self._chk(self.lib.ygz_hip_ba_optimize(self._ctx, C.byref(pb), _p(po, C.c_double), _p(pt, C.c_double), iterations, C.byref(st)),
                  "ba_optimize")
        return po, pt, st

    def ba_optimize_chi2(self, poses, fixed, points, edge_pose, edge_point, obs, iterations=20, huber_delta=5.991, cam=None):
        """ba_optimize + the per-edge chi2 at the optimised state (what ba::LocalBAG2O marks its outliers with): one upload, one transfer back"""
        pb = self._ba_problem(poses, fixed, points, edge_pose, edge_point, obs, huber_delta, 0, cam)
        po = np.ascontiguousarray(poses, np.float64).copy()
        pt = np.ascontiguousarray(points, np.float64).copy()
        chi = np.zeros(max(len(np.atleast_1d(edge_pose)), 1), np.float64)
        st = BaStats()
        self._chk(self.lib.ygz_hip_ba_optimize_chi2(self._ctx, C.byref(pb), _p(po, C.c_double), _p(pt, C.c_double), iterations, C.byref(st),
                                                    _p(chi, C.c_double)), "ba_optimize_chi2")
        return po, pt, st, chi[:len(np.atleast_1d(edge_pose))]

    def ba_optimize_resident(self, window_begin, n_windows, iterations=20, want_stats=True):
        st = (BaStats * n_windows)()
        self._chk(self.lib.ygz_hip_ba_optimize_resident(self._ctx, window_begin, n_windows, iterations, st if want_stats else None),
                  "ba_optimize_resident")
        return list(st) if want_stats else None

    def ba_last_path(self):
        """(resident?, reasons) of the last ba_optimize / ba_solve_ceres: the resident kernel, or the ~10x slower host loop and why"""
        v = int(self.lib.ygz_hip_ba_last_path(self._ctx))
        return bool(v & 1), [n for b, n in ((16, "more than 20 free poses"), (32, "repeated (point, pose) edges"), (64, "YGZ_BA_HOST_LOOP=1")) if v & b]

    def ba_light_barrier(self):
        """1: the same-XCD barrier of the resident LM passed its self-test on this device and is in use; 0: failed (full barriers); -1: not run yet"""
        return int(self.lib.ygz_hip_ba_light_barrier(self._ctx))

    def ba_set_team_placement(self, spread):
        self._chk(self.lib.ygz_hip_ba_set_team_placement(self._ctx, int(spread)), "ba_set_team_placement")

    def ba_set_team_budget(self, workgroups):
        self._chk(self.lib.ygz_hip_ba_set_team_budget(self._ctx, int(workgroups)), "ba_set_team_budget")

    def ba_get_state(self, window, K, P):
        po, pt = np.empty((K, 6)), np.empty((P, 3))
        self._chk(self.lib.ygz_hip_ba_get_state(self._ctx, window, _p(po, C.c_double), _p(pt, C.c_double)), "ba_get_state")
        return po, pt

    def ba_upload(self, window, poses, fixed, points, edge_pose, edge_point, obs, huber_delta=5.991, formulation=0, cam=None,
                  point_fixed=None, edge_huber=None, edge_enable=None):
        pb = self._ba_problem(poses, fixed, points, edge_pose, edge_point, obs, huber_delta, formulation, cam, point_fixed, edge_huber, edge_enable)
        self._chk(self.lib.ygz_hip_ba_upload(self._ctx, window, C.byref(pb)), "ba_upload")
        return pb.n_poses, pb.n_points, pb.n_edges

    def ba_set_enable(self, window, edge_enable):
        en = np.ascontiguousarray(edge_enable, np.uint8)
        self._chk(self.lib.ygz_hip_ba_set_enable(self._ctx, window, _p(en, C.c_uint8)), "ba_set_enable")

    def ba_set_state(self, window, poses, points):
        poses = np.ascontiguousarray(poses, np.float64)
        points = np.ascontiguousarray(points, np.float64)
        self._chk(self.lib.ygz_hip_ba_set_state(self._ctx, window, _p(poses, C.c_double), _p(points, C.c_double)), "ba_set_state")

    def ba_set_state_device(self, window, d_poses_ptr, d_points_ptr):
        self._chk(self.lib.ygz_hip_ba_set_state_device(self._ctx, window, C.c_void_p(d_poses_ptr), C.c_void_p(d_points_ptr)),
                  "ba_set_state_device")

    def ba_linearize_resident(self, window_begin=0, n_windows=1):
        self._chk(self.lib.ygz_hip_ba_linearize_resident(self._ctx, window_begin, n_windows), "ba_linearize_resident")

    def ba_download(self, window, K, P, E):
        o = self._ba_out(K, P, E)
        d = lambda k: _p(o[k], C.c_double)
        self._chk(self.lib.ygz_hip_ba_download(self._ctx, window, d("Hpp"), d("bp"), d("Hll"), d("bl"), d("Hpl"), d("err"),
                                               d("chi2_edge"), d("chi2")), "ba_download")
        o["chi2"] = float(o["chi2"][0])
        return o
