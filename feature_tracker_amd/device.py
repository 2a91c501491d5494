"""ctypes binding of the C ABI in include/ft_device.h (libft_device.so, hand-written gfx950 kernels).

There is no CPU fallback: loading fails loudly when the library is missing, and every call raises
``FtError`` on a negative return code.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_lib", "libft_device.so")

FT_LK_USE_INITIAL_FLOW = 4
FT_LK_GET_MIN_EIGENVALS = 8
FT_LK_NO_ERROR = 1024
FT_LK_FORWARD_BACKWARD = 256
FT_LK_REJECT_OUTLIERS = 512
FT_MAX_LEVELS = 8
FT_MAX_WIN = 32
FT_MAX_BATCH = 256


class FtError(RuntimeError):
    def __init__(self, code, what, detail=""):
        super().__init__(f"{what}: error {code}" + (f" ({detail})" if detail else ""))
        self.code = code


class ft_limits(C.Structure):
    _fields_ = [("max_width", C.c_int), ("max_height", C.c_int), ("max_levels", C.c_int), ("n_streams", C.c_int),
                ("n_slots", C.c_int), ("max_points", C.c_int), ("max_grid_cells", C.c_int), ("max_per_cell", C.c_int)]


class ft_lk_params(C.Structure):
    _fields_ = [("win_w", C.c_int), ("win_h", C.c_int), ("max_level", C.c_int), ("max_iter", C.c_int),
                ("eps", C.c_double), ("min_eig", C.c_double), ("flags", C.c_int), ("fb_thresh", C.c_float),
                ("ransac_hypotheses", C.c_int), ("ransac_seed", C.c_uint), ("stereo_threshold", C.c_double)]


class ft_lk_job(C.Structure):
    _fields_ = [("stream_prev", C.c_int), ("slot_prev", C.c_int), ("stream_next", C.c_int), ("slot_next", C.c_int),
                ("n", C.c_int), ("prev_xy", C.c_void_p), ("next_xy", C.c_void_p), ("status", C.c_void_p),
                ("err", C.c_void_p), ("H", C.c_void_p), ("iters", C.c_void_p), ("calib_prev", C.c_void_p), ("calib_next", C.c_void_p),
                ("stereo_Rt", C.c_void_p)]


class ft_match_set(C.Structure):
    _fields_ = [("n", C.c_int), ("prev_xy", C.c_void_p), ("next_xy", C.c_void_p), ("status", C.c_void_p),
                ("calib_prev", C.c_void_p), ("calib_next", C.c_void_p)]


class ft_match(C.Structure):
    _fields_ = [("query", C.c_int), ("train", C.c_int), ("distance", C.c_float)]


class ft_keypoint(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("response", C.c_float)]


class ft_kernel_time(C.Structure):
    _fields_ = [("pyramid_ms", C.c_double), ("pyramid_launches", C.c_int), ("lk_ms", C.c_double), ("lk_launches", C.c_int),
                ("fast_ms", C.c_double), ("fast_launches", C.c_int),
                ("lk_points", C.c_longlong), ("lk_level_visits", C.c_longlong), ("lk_iterations", C.c_longlong),
                ("reject_ms", C.c_double), ("reject_launches", C.c_int)]


# every symbol include/ft_device.h declares
SYMBOLS = ["ft_ctx_create", "ft_ctx_destroy", "ft_upload", "ft_set_frame_device", "ft_set_frames_device",
           "ft_set_frames_device_pyramid", "ft_build_pyramid",
           "ft_build_gradients", "ft_lk_track", "ft_lk_track_async", "ft_reject_outliers", "ft_reject_outliers_async", "ft_match_descriptors", "ft_compute_descriptors", "ft_stage_frame", "ft_set_frames_staged_pyramid", "ft_fast_grid", "ft_fast_grid_async",
           "ft_homography", "ft_predict_flow", "ft_level_info", "ft_download_level", "ft_download_grad", "ft_sync", "ft_fence", "ft_wait", "ft_set_front_queue", "ft_flush",
           "ft_profile", "ft_kernel_times", "ft_queue_time_begin", "ft_queue_time_end", "ft_strerror",
           "ft_last_hip_error", "ft_device_count"]

_lib = None


def _preload_torch_hip_runtime():
    """PyTorch wheels bundle their own libamdhip64.so.  Two HIP runtimes in one process cannot both own the GPU, so
    when torch is installed its copy is mapped first: this library's libamdhip64.so.7 dependency then resolves to it
    (same SONAME) and a later ``import torch`` finds its copy already loaded."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        C.CDLL(path, mode=C.RTLD_GLOBAL)


def load_library():
    """Load libft_device.so; raise if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build the HIP extension first (__graft_entry__.build()); "
                          "there is no CPU fallback for the device path")
    _preload_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    vp, ip = C.c_void_p, C.POINTER(C.c_int)
    lib.ft_ctx_create.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(ft_limits)]
    lib.ft_ctx_destroy.argtypes = [vp]
    lib.ft_ctx_destroy.restype = None
    lib.ft_upload.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_size_t]
    lib.ft_set_frame_device.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_size_t]
    lib.ft_set_frames_device.argtypes = [vp, C.c_int, ip, ip, C.POINTER(vp), C.c_int, C.c_int, C.c_size_t]
    lib.ft_set_frames_device_pyramid.argtypes = [vp, C.c_int, ip, ip, C.POINTER(vp), C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int, ip]
    lib.ft_build_pyramid.argtypes = [vp, C.c_int, ip, ip, C.c_int, C.c_int, C.c_int, C.c_int, ip]
    lib.ft_build_gradients.argtypes = [vp, C.c_int, ip, ip]
    lib.ft_lk_track.argtypes = [vp, C.c_int, C.POINTER(ft_lk_job), C.POINTER(ft_lk_params)]
    lib.ft_lk_track_async.argtypes = [vp, C.c_int, C.POINTER(ft_lk_job), C.POINTER(ft_lk_params)]
    lib.ft_stage_frame.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_size_t]
    lib.ft_set_frames_staged_pyramid.argtypes = [vp, C.c_int, ip, ip, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, ip]
    lib.ft_match_descriptors.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_float, C.POINTER(ft_match), C.c_int, ip]
    lib.ft_compute_descriptors.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, vp, vp]
    lib.ft_reject_outliers.argtypes = [vp, C.c_int, C.POINTER(ft_match_set), C.c_int, C.c_uint]
    lib.ft_reject_outliers_async.argtypes = [vp, C.c_int, C.POINTER(ft_match_set), C.c_int, C.c_uint]
    fg = [vp, C.c_int, ip, ip, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(ft_keypoint), C.c_int, ip, ip]
    lib.ft_fast_grid.argtypes = fg
    lib.ft_fast_grid_async.argtypes = fg
    fp = C.POINTER(C.c_float)
    lib.ft_homography.argtypes = [fp, fp, fp]
    lib.ft_homography.restype = None
    lib.ft_predict_flow.argtypes = [vp, fp, fp, fp, C.c_int]
    lib.ft_level_info.argtypes = [vp, C.c_int, C.c_int, C.c_int, ip, ip]
    lib.ft_download_level.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_size_t]
    lib.ft_download_grad.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_size_t]
    lib.ft_sync.argtypes = [vp]
    lib.ft_fence.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.ft_wait.argtypes = [vp, C.c_uint64]
    lib.ft_set_front_queue.argtypes = [vp, C.c_int]
    lib.ft_flush.argtypes = [vp]
    lib.ft_profile.argtypes = [vp, C.c_int]
    lib.ft_kernel_times.argtypes = [vp, C.POINTER(ft_kernel_time), C.c_int]
    lib.ft_queue_time_begin.argtypes = [vp]
    lib.ft_queue_time_end.argtypes = [vp, fp]
    lib.ft_strerror.argtypes = [C.c_int]
    lib.ft_strerror.restype = C.c_char_p
    lib.ft_last_hip_error.argtypes = [vp]
    lib.ft_last_hip_error.restype = C.c_char_p
    lib.ft_device_count.argtypes = []
    _lib = lib
    return lib


def _ints(v):
    a = np.ascontiguousarray(v, dtype=np.int32)
    return a, a.ctypes.data_as(C.POINTER(C.c_int))


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class LKParams:
    """Window, depth and termination of one calcOpticalFlowPyrLK call (defaults: src/track/TrackKLT.h:121-122,
    src/track/TrackKLT.cpp:305,:401)."""

    def __init__(self, win=15, max_level=3, max_iter=15, eps=0.01, min_eig=1e-4, flags=FT_LK_USE_INITIAL_FLOW, fb_thresh=1.0,
                 ransac_hypotheses=0, ransac_seed=0, stereo_threshold=3.0):
        self.win_w, self.win_h = (win, win) if np.isscalar(win) else win
        self.max_level, self.max_iter, self.eps, self.min_eig = max_level, max_iter, eps, min_eig
        self.flags, self.fb_thresh = flags, fb_thresh
        self.ransac_hypotheses, self.ransac_seed = ransac_hypotheses, ransac_seed
        self.stereo_threshold = stereo_threshold

    def c_struct(self):
        return ft_lk_params(self.win_w, self.win_h, self.max_level, self.max_iter, self.eps, self.min_eig, self.flags, self.fb_thresh,
                            self.ransac_hypotheses, self.ransac_seed, self.stereo_threshold)


class DeviceContext:
    """Owns one ft_ctx: device pyramids for ``n_streams`` camera streams x ``n_slots`` ring slots."""

    def __init__(self, max_width, max_height, max_levels=4, n_streams=1, n_slots=2, max_points=4096, max_grid_cells=64,
                 max_per_cell=32, device=0):
        self.lib = load_library()
        self._ctx = C.c_void_p()
        lim = ft_limits(max_width, max_height, max_levels, n_streams, n_slots, max_points, max_grid_cells, max_per_cell)
        rc = self.lib.ft_ctx_create(C.byref(self._ctx), device, C.byref(lim))
        if rc != 0:
            self._ctx = C.c_void_p()
            raise FtError(rc, "ft_ctx_create", self.lib.ft_strerror(rc).decode())
        self.limits = lim
        self._keep = []   # host arrays borrowed by async calls until the next sync

    # -- plumbing --
    def _check(self, rc, what):
        if rc != 0:
            detail = self.lib.ft_strerror(rc).decode()
            if rc == -3:
                detail += ": " + self.lib.ft_last_hip_error(self._ctx).decode()
            raise FtError(rc, what, detail)

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self.lib.ft_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- frames and pyramids --
    def upload(self, stream, slot, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        self._check(self.lib.ft_upload(self._ctx, stream, slot, img.ctypes.data, w, h, img.strides[0]), "ft_upload")

    def set_frame_device(self, stream, slot, dev_ptr, w, h, stride):
        self._check(self.lib.ft_set_frame_device(self._ctx, stream, slot, C.c_void_p(dev_ptr), w, h, stride), "ft_set_frame_device")

    def set_frames_device(self, streams, slots, dev_ptrs, w, h, stride):
        s, sp = _ints(streams)
        l, lp = _ints(slots)
        ptrs = (C.c_void_p * len(s))(*[int(p) for p in dev_ptrs])
        self._check(self.lib.ft_set_frames_device(self._ctx, len(s), sp, lp, ptrs, w, h, stride), "ft_set_frames_device")

    def set_frames_device_pyramid(self, streams, slots, dev_ptrs, w, h, stride, max_level=3, win=15):
        """Ingest device-resident frames and build their pyramids (no gradient images) in one pass."""
        s, sp = _ints(streams)
        l, lp = _ints(slots)
        win_w, win_h = (win, win) if np.isscalar(win) else win
        ptrs = (C.c_void_p * len(s))(*[int(p) for p in dev_ptrs])
        ach = np.zeros(len(s), dtype=np.int32)
        self._check(self.lib.ft_set_frames_device_pyramid(self._ctx, len(s), sp, lp, ptrs, w, h, stride, max_level, win_w, win_h,
                                                          ach.ctypes.data_as(C.POINTER(C.c_int))), "ft_set_frames_device_pyramid")
        return ach

    def set_frames_host_pyramid(self, streams, slots, frames, max_level=3, win=15):
        """Frames in host memory (list of (h, w) uint8 arrays of one size): staged into pinned memory, then level 0 and
        the pyramids in one pass that reads the staging over the bus (no copy command)."""
        s, sp = _ints(streams)
        l, lp = _ints(slots)
        win_w, win_h = (win, win) if np.isscalar(win) else win
        h, w = frames[0].shape
        for st, sl, f in zip(s, l, frames):
            f = np.ascontiguousarray(f, dtype=np.uint8)
            self._check(self.lib.ft_stage_frame(self._ctx, int(st), int(sl), f.ctypes.data, w, h, f.strides[0]), "ft_stage_frame")
        ach = np.zeros(len(s), dtype=np.int32)
        self._check(self.lib.ft_set_frames_staged_pyramid(self._ctx, len(s), sp, lp, w, h, max_level, win_w, win_h,
                                                          ach.ctypes.data_as(C.POINTER(C.c_int))), "ft_set_frames_staged_pyramid")
        return ach

    def build_pyramid(self, streams, slots, max_level=3, win=15, with_gradients=True):
        s, sp = _ints(streams)
        l, lp = _ints(slots)
        win_w, win_h = (win, win) if np.isscalar(win) else win
        ach = np.zeros(len(s), dtype=np.int32)
        self._check(self.lib.ft_build_pyramid(self._ctx, len(s), sp, lp, max_level, win_w, win_h, int(with_gradients),
                                              ach.ctypes.data_as(C.POINTER(C.c_int))), "ft_build_pyramid")
        return ach

    def build_gradients(self, streams, slots):
        s, sp = _ints(streams)
        l, lp = _ints(slots)
        self._check(self.lib.ft_build_gradients(self._ctx, len(s), sp, lp), "ft_build_gradients")

    def level_info(self, stream, slot, level):
        w, h = C.c_int(), C.c_int()
        self._check(self.lib.ft_level_info(self._ctx, stream, slot, level, C.byref(w), C.byref(h)), "ft_level_info")
        return w.value, h.value

    def download_level(self, stream, slot, level):
        w, h = self.level_info(stream, slot, level)
        out = np.empty((h, w), dtype=np.uint8)
        self._check(self.lib.ft_download_level(self._ctx, stream, slot, level, out.ctypes.data, out.strides[0]), "ft_download_level")
        return out

    def download_grad(self, stream, slot, level):
        w, h = self.level_info(stream, slot, level)
        out = np.empty((h, w, 2), dtype=np.int16)
        self._check(self.lib.ft_download_grad(self._ctx, stream, slot, level, out.ctypes.data, 2 * w), "ft_download_grad")
        return out

    # -- LK --
    def lk_track(self, jobs, params, want_iters=False, sync=True):
        """jobs: list of dicts {prev:(stream,slot), next:(stream,slot), prev_xy:(n,2) float32, guess:(n,2)|None, H:(3,3)|None,
        calib_prev / calib_next:(8,)|None, stereo_Rt:(12,)|None}.
        Returns a list of (next_xy, status, err[, iters]) per job (filled after the sync)."""
        cj = (ft_lk_job * len(jobs))()
        outs = []
        keep = []
        nlev = params.max_level + 1
        for i, j in enumerate(jobs):
            prev_xy = np.ascontiguousarray(j["prev_xy"], dtype=np.float32).reshape(-1, 2)
            n = prev_xy.shape[0]
            guess = j.get("guess")
            nxt = np.array(prev_xy if guess is None else guess, dtype=np.float32).reshape(-1, 2).copy()
            st = np.zeros(n, dtype=np.uint8)
            er = np.zeros(n, dtype=np.float32)
            it = np.full((n, nlev), -1, dtype=np.int32) if want_iters else None
            H = None if j.get("H") is None else np.ascontiguousarray(j["H"], dtype=np.float32).reshape(9)
            c0 = None if j.get("calib_prev") is None else np.ascontiguousarray(j["calib_prev"], dtype=np.float64).reshape(8)
            c1 = c0 if j.get("calib_next") is None else np.ascontiguousarray(j["calib_next"], dtype=np.float64).reshape(8)
            rt = None if j.get("stereo_Rt") is None else np.ascontiguousarray(j["stereo_Rt"], dtype=np.float64).reshape(12)
            cj[i] = ft_lk_job(j["prev"][0], j["prev"][1], j["next"][0], j["next"][1], n, prev_xy.ctypes.data, nxt.ctypes.data,
                              st.ctypes.data, er.ctypes.data, None if H is None else H.ctypes.data,
                              None if it is None else it.ctypes.data, None if c0 is None else c0.ctypes.data,
                              None if c1 is None else c1.ctypes.data, None if rt is None else rt.ctypes.data)
            keep += [prev_xy, nxt, st, er, it, H, c0, c1, rt]
            outs.append((nxt, st, er, it) if want_iters else (nxt, st, er))
        p = params.c_struct()
        fn = self.lib.ft_lk_track if sync else self.lib.ft_lk_track_async
        self._check(fn(self._ctx, len(jobs), cj, C.byref(p)), "ft_lk_track")
        if not sync:
            self._keep += keep + [cj]
        return outs

    def reject_outliers(self, sets, hypotheses=256, seed=0):
        """sets: list of dicts {prev_xy:(n,2), next_xy:(n,2), status:(n,) uint8 | None, calib_prev:(8,), calib_next:(8,) | None}.
        Returns the list of output status arrays."""
        cs = (ft_match_set * len(sets))()
        keep, outs = [], []
        for i, q in enumerate(sets):
            a = np.ascontiguousarray(q["prev_xy"], dtype=np.float32).reshape(-1, 2)
            b = np.ascontiguousarray(q["next_xy"], dtype=np.float32).reshape(-1, 2)
            n = a.shape[0]
            st = np.ones(n, dtype=np.uint8) if q.get("status") is None else np.array(q["status"], dtype=np.uint8).copy()
            c0 = np.ascontiguousarray(q["calib_prev"], dtype=np.float64).reshape(8)
            c1 = c0 if q.get("calib_next") is None else np.ascontiguousarray(q["calib_next"], dtype=np.float64).reshape(8)
            cs[i] = ft_match_set(n, a.ctypes.data, b.ctypes.data, st.ctypes.data, c0.ctypes.data, c1.ctypes.data)
            keep += [a, b, c0, c1]
            outs.append(st)
        self._check(self.lib.ft_reject_outliers(self._ctx, len(sets), cs, hypotheses, seed), "ft_reject_outliers")
        return outs

    def match_descriptors(self, desc0, desc1, knn_ratio=0.7):
        """desc0 (n0, 32), desc1 (n1, 32) uint8.  Returns (m, 3) int32 rows (query, train, distance) in query order."""
        a = np.ascontiguousarray(desc0, dtype=np.uint8).reshape(-1, 32)
        b = np.ascontiguousarray(desc1, dtype=np.uint8).reshape(-1, 32)
        cap = max(1, a.shape[0])
        out = (ft_match * cap)()
        n = C.c_int(0)
        self._check(self.lib.ft_match_descriptors(self._ctx, a.ctypes.data, a.shape[0], b.ctypes.data, b.shape[0], knn_ratio, out, cap,
                                                  C.byref(n)), "ft_match_descriptors")
        return np.array([(out[i].query, out[i].train, int(out[i].distance)) for i in range(n.value)], dtype=np.int32).reshape(-1, 3)

    def compute_descriptors(self, stream, slot, xy):
        """xy (n, 2) keypoint positions on level 0.  Returns (desc (n, 32) uint8, valid (n,) uint8)."""
        p = np.ascontiguousarray(xy, dtype=np.float32).reshape(-1, 2)
        n = p.shape[0]
        desc, valid = np.zeros((n, 32), dtype=np.uint8), np.zeros(n, dtype=np.uint8)
        self._check(self.lib.ft_compute_descriptors(self._ctx, stream, slot, p.ctypes.data, n, desc.ctypes.data, valid.ctypes.data), "ft_compute_descriptors")
        return desc, valid

    # -- FAST --
    def fast_grid(self, streams, slots, threshold=20, grid_x=5, grid_y=5, per_cell=7, cap=None, sync=True):
        s, sp = _ints(streams)
        l, lp = _ints(slots)
        n = len(s)
        if cap is None:
            cap = self.limits.max_grid_cells * per_cell
        out = np.zeros((n, cap, 3), dtype=np.float32)
        n_out = np.zeros(n, dtype=np.int32)
        cells = np.full((n, self.limits.max_grid_cells), -1, dtype=np.int32)
        # cell_counts is written densely as n*cells entries: fetch through a flat scratch and reshape by the caller
        flat = np.zeros(n * self.limits.max_grid_cells, dtype=np.int32)
        fn = self.lib.ft_fast_grid if sync else self.lib.ft_fast_grid_async
        self._check(fn(self._ctx, n, sp, lp, threshold, grid_x, grid_y, per_cell,
                       out.ctypes.data_as(C.POINTER(ft_keypoint)), cap, n_out.ctypes.data_as(C.POINTER(C.c_int)),
                       flat.ctypes.data_as(C.POINTER(C.c_int))), "ft_fast_grid")
        if not sync:
            self._keep += [out, n_out, flat, s, l]
        return out, n_out, flat

    # -- gyro seed --
    def homography(self, K4, R):
        K4 = np.ascontiguousarray(K4, dtype=np.float32)
        R = np.ascontiguousarray(R, dtype=np.float32).reshape(9)
        H = np.zeros(9, dtype=np.float32)
        self.lib.ft_homography(_fptr(K4), _fptr(R), _fptr(H))
        return H.reshape(3, 3)

    def predict_flow(self, H, xy):
        H = np.ascontiguousarray(H, dtype=np.float32).reshape(9)
        xy = np.ascontiguousarray(xy, dtype=np.float32).reshape(-1, 2)
        out = np.zeros_like(xy)
        self._check(self.lib.ft_predict_flow(self._ctx, _fptr(H), _fptr(xy), _fptr(out), xy.shape[0]), "ft_predict_flow")
        return out

    # -- sync / timing --
    def sync(self):
        self._check(self.lib.ft_sync(self._ctx), "ft_sync")
        self._keep.clear()

    def fence(self):
        """Mark the current end of the queue; returns a token for wait()."""
        tok = C.c_uint64()
        self._check(self.lib.ft_fence(self._ctx, C.byref(tok)), "ft_fence")
        return tok.value

    def wait(self, token):
        """Block until the queue has passed the fence and deliver the outputs of the async calls made before it.
        Output arrays of later async calls are still owned by the library until their own wait or sync()."""
        self._check(self.lib.ft_wait(self._ctx, C.c_uint64(token)), "ft_wait")

    def set_front_queue(self, on):
        """Ingest, pyramids and detector on a device queue of their own, beside the tracking work (see ft_device.h)."""
        self._check(self.lib.ft_set_front_queue(self._ctx, 1 if on else 0), "ft_set_front_queue")

    def flush(self):
        """Launch the pyramid steps the library holds back for a detector call that may follow (see ft_device.h)."""
        self._check(self.lib.ft_flush(self._ctx), "ft_flush")

    def profile(self, enable):
        self._check(self.lib.ft_profile(self._ctx, int(enable)), "ft_profile")

    def kernel_times(self, reset=True):
        t = ft_kernel_time()
        self._check(self.lib.ft_kernel_times(self._ctx, C.byref(t), int(reset)), "ft_kernel_times")
        return {"pyramid_ms": t.pyramid_ms, "pyramid_launches": t.pyramid_launches, "lk_ms": t.lk_ms,
                "lk_launches": t.lk_launches, "fast_ms": t.fast_ms, "fast_launches": t.fast_launches,
                "lk_points": t.lk_points, "lk_level_visits": t.lk_level_visits, "lk_iterations": t.lk_iterations,
                "reject_ms": t.reject_ms, "reject_launches": t.reject_launches}

    def queue_time_begin(self):
        self._check(self.lib.ft_queue_time_begin(self._ctx), "ft_queue_time_begin")

    def queue_time_end(self):
        ms = C.c_float()
        self._check(self.lib.ft_queue_time_end(self._ctx, C.byref(ms)), "ft_queue_time_end")
        return ms.value


def device_count():
    return load_library().ft_device_count()
