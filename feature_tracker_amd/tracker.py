"""ctypes binding of include/ft_tracker.h: the host tracker layer (TrackKLT / TrackKLTBatch) on the gfx950 device
library.  ``load_library`` knows one library only, the product's; there is no fallback.  The classes accept an already
bound library object (``bind``) so that a harness can drive the identical wrapper over another build of the same C
interface."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_lib", "libft_tracker.so")


class ftk_options(C.Structure):
    _fields_ = [("num_pts", C.c_int), ("fast_threshold", C.c_int), ("grid_x", C.c_int), ("grid_y", C.c_int),
                ("min_px_dist", C.c_int), ("pyr_levels", C.c_int), ("win_size", C.c_int), ("lk_max_iter", C.c_int),
                ("lk_eps", C.c_double), ("lk_min_eig", C.c_double), ("use_gyro_seed", C.c_int), ("forward_backward", C.c_int),
                ("fb_thresh", C.c_float), ("speculative_detect", C.c_int),
                ("reject_outliers", C.c_int), ("ransac_hypotheses", C.c_int),
                ("stereo_extrinsic_seed", C.c_int), ("stereo_threshold", C.c_double), ("drop_fraction", C.c_double),
                ("use_klt", C.c_int), ("knn_ratio", C.c_double), ("descriptor_ransac_on_pairs", C.c_int)]


SYMBOLS = ["ftk_default_options", "ftk_tracker_create", "ftk_tracker_destroy", "ftk_tracker_set_calibration",
           "ftk_tracker_feed_imu", "ftk_tracker_feed_stereo", "ftk_tracker_feed_monocular", "ftk_tracker_get_points",
           "ftk_tracker_database_size", "ftk_tracker_get_track", "ftk_batch_create", "ftk_batch_create_grouped", "ftk_batch_destroy",
           "ftk_batch_set_calibration", "ftk_batch_feed_imu", "ftk_batch_feed", "ftk_batch_feed_pipelined", "ftk_batch_flush", "ftk_batch_get_points", "ftk_batch_database_size", "ftk_batch_get_track", "ftk_batch_groups", "ftk_batch_group_ctx",
           "ftk_batch_stage_times", "ftk_batch_device_ctx", "ftk_backend_name"]

_lib = None


def load_library():
    """The product library (libft_tracker.so on libft_device.so); raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it first (__graft_entry__.build()); there is no fallback")
        from . import device
        device.load_library()      # maps the HIP runtime and libft_device.so first
        _lib = bind(C.CDLL(LIB_PATH))
    return _lib


def bind(L):
    """Declare the argument types of the ftk_* interface on a loaded library and return it."""
    vp, dp = C.c_void_p, C.POINTER(C.c_double)
    L.ftk_default_options.argtypes = [C.POINTER(ftk_options)]
    L.ftk_default_options.restype = None
    L.ftk_tracker_create.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(ftk_options), C.c_int, C.c_int]
    L.ftk_tracker_destroy.argtypes = [vp]
    L.ftk_tracker_destroy.restype = None
    L.ftk_tracker_set_calibration.argtypes = [vp, C.c_int, dp, dp]
    L.ftk_tracker_feed_imu.argtypes = [vp, C.c_double, dp, dp]
    L.ftk_tracker_feed_stereo.argtypes = [vp, C.c_double, vp, vp, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int]
    L.ftk_tracker_feed_monocular.argtypes = [vp, C.c_double, vp, C.c_int, C.c_int, C.c_size_t, C.c_int]
    L.ftk_tracker_get_points.argtypes = [vp, C.c_int, vp, vp, C.c_int]
    L.ftk_tracker_database_size.argtypes = [vp]
    L.ftk_tracker_get_track.argtypes = [vp, C.c_uint64, C.c_int, vp, vp, C.c_int]
    L.ftk_batch_create.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(ftk_options), C.c_int, C.c_int, C.c_int]
    L.ftk_batch_create_grouped.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(ftk_options), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.ftk_batch_destroy.argtypes = [vp]
    L.ftk_batch_destroy.restype = None
    L.ftk_batch_set_calibration.argtypes = [vp, C.c_int, dp, dp]
    L.ftk_batch_feed_imu.argtypes = [vp, C.c_int, C.c_double, dp, dp]
    L.ftk_batch_feed.argtypes = [vp, dp, C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_size_t, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    L.ftk_batch_feed_pipelined.argtypes = L.ftk_batch_feed.argtypes
    L.ftk_batch_flush.argtypes = [vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    L.ftk_batch_get_points.argtypes = [vp, C.c_int, vp, vp, C.c_int]
    L.ftk_batch_database_size.argtypes = [vp, C.c_int]
    L.ftk_batch_get_track.argtypes = [vp, C.c_int, C.c_uint64, vp, vp, C.c_int]
    L.ftk_batch_groups.argtypes = [vp]
    L.ftk_batch_group_ctx.argtypes = [vp, C.c_int]
    L.ftk_batch_group_ctx.restype = vp
    L.ftk_batch_stage_times.argtypes = [vp, dp, C.POINTER(C.c_longlong), C.c_int]
    L.ftk_batch_device_ctx.argtypes = [vp]
    L.ftk_batch_device_ctx.restype = vp
    L.ftk_backend_name.argtypes = []
    L.ftk_backend_name.restype = C.c_char_p
    return L


def make_options(lib, **kw):
    o = ftk_options()
    lib.ftk_default_options(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise TypeError(f"unknown tracker option {k}")
        setattr(o, k, v)
    return o


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class _Base:
    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed with code {rc}")


class Tracker(_Base):
    """One TrackKLT with cameras 0 and 1 (reference driver usage: src/test_tracking.cpp:229-231,:148,:207)."""

    def __init__(self, max_width, max_height, device=0, lib=None, **options):
        self.lib = lib if lib is not None else load_library()
        self.opts = make_options(self.lib, **options)
        self._h = C.c_void_p()
        self._check(self.lib.ftk_tracker_create(C.byref(self._h), device, C.byref(self.opts), max_width, max_height), "ftk_tracker_create")

    def close(self):
        if self._h:
            self.lib.ftk_tracker_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_calibration(self, calib8, T_cam_imu=None):
        k = np.ascontiguousarray(calib8, dtype=np.float64).reshape(-1, 8)
        T = None if T_cam_imu is None else np.ascontiguousarray(T_cam_imu, dtype=np.float64).reshape(-1, 16)
        self._check(self.lib.ftk_tracker_set_calibration(self._h, k.shape[0], _dp(k), None if T is None else _dp(T)), "set_calibration")

    def feed_imu(self, t, am, wm):
        a, w = np.ascontiguousarray(am, dtype=np.float64), np.ascontiguousarray(wm, dtype=np.float64)
        self._check(self.lib.ftk_tracker_feed_imu(self._h, t, _dp(a), _dp(w)), "feed_imu")

    def feed_stereo(self, t, left, right, cam_left=0, cam_right=1):
        left, right = np.ascontiguousarray(left, dtype=np.uint8), np.ascontiguousarray(right, dtype=np.uint8)
        h, w = left.shape
        self._check(self.lib.ftk_tracker_feed_stereo(self._h, t, left.ctypes.data, right.ctypes.data, w, h, left.strides[0], cam_left, cam_right), "feed_stereo")

    def feed_monocular(self, t, img, cam=0):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        self._check(self.lib.ftk_tracker_feed_monocular(self._h, t, img.ctypes.data, w, h, img.strides[0], cam), "feed_monocular")

    def points(self, cam=0):
        n = self.lib.ftk_tracker_get_points(self._h, cam, None, None, 0)
        xy, ids = np.zeros((n, 2), dtype=np.float32), np.zeros(n, dtype=np.uint64)
        if n:
            self.lib.ftk_tracker_get_points(self._h, cam, xy.ctypes.data, ids.ctypes.data, n)
        return xy, ids

    def database_size(self):
        return self.lib.ftk_tracker_database_size(self._h)

    def track(self, feat_id, cam=0):
        n = self.lib.ftk_tracker_get_track(self._h, int(feat_id), cam, None, None, 0)
        uv, ts = np.zeros((n, 4), dtype=np.float32), np.zeros(n, dtype=np.float64)
        if n > 0:
            self.lib.ftk_tracker_get_track(self._h, int(feat_id), cam, uv.ctypes.data, ts.ctypes.data, n)
        return uv, ts


class BatchTracker(_Base):
    """n independent monocular streams through shared batched launches (TrackKLTBatch)."""

    def __init__(self, n_streams, max_width, max_height, device=0, lib=None, n_groups=0, host_threads=0, **options):
        self.lib = lib if lib is not None else load_library()
        self.opts = make_options(self.lib, **options)
        self.n = n_streams
        self._h = C.c_void_p()
        self._check(self.lib.ftk_batch_create_grouped(C.byref(self._h), device, C.byref(self.opts), n_streams, max_width, max_height,
                                                      n_groups, host_threads), "ftk_batch_create")
        self._ptrs = (C.c_void_p * n_streams)()
        self._ts = np.zeros(n_streams, dtype=np.float64)

    def close(self):
        if self._h:
            self.lib.ftk_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_calibration(self, stream, calib8, T_cam_imu=None):
        k = np.ascontiguousarray(calib8, dtype=np.float64).reshape(8)
        T = None if T_cam_imu is None else np.ascontiguousarray(T_cam_imu, dtype=np.float64).reshape(16)
        self._check(self.lib.ftk_batch_set_calibration(self._h, stream, _dp(k), None if T is None else _dp(T)), "batch_set_calibration")

    def feed_imu(self, stream, t, am, wm):
        a, w = np.ascontiguousarray(am, dtype=np.float64), np.ascontiguousarray(wm, dtype=np.float64)
        self._check(self.lib.ftk_batch_feed_imu(self._h, stream, t, _dp(a), _dp(w)), "batch_feed_imu")

    def feed_host(self, timestamps, frames, pipelined=False):
        """frames: sequence of n (h, w) uint8 arrays (copied into pinned staging during the call)."""
        keep = [np.ascontiguousarray(f, dtype=np.uint8) for f in frames]
        h, w = keep[0].shape
        for i, f in enumerate(keep):
            self._ptrs[i] = f.ctypes.data
        # a pipelined set may complete up to three calls later: keep the arrays of the last four sets alive
        self._keep = (getattr(self, "_keep", []) + [keep])[-6:]
        return self._feed(timestamps, 0, w, h, keep[0].strides[0], pipelined)

    def feed_host_prepared(self, timestamps, host_ptrs, w, h, stride, pipelined=False):
        """feed_host without per-call marshalling: ``timestamps`` a contiguous float64 array of n, ``host_ptrs`` a
        ctypes (c_void_p * n) array of frame addresses (the caller keeps the frames alive)."""
        return self._feed(timestamps, 0, w, h, stride, pipelined, ptrs=host_ptrs)

    def feed_device(self, timestamps, dev_ptrs, w, h, stride, pipelined=False):
        """dev_ptrs: n device addresses of frames resident in HBM.  pipelined=True leaves the last group's frame in
        flight (finish with flush()); the returned counts are those of the frames completed by the call."""
        if isinstance(dev_ptrs, C.Array):     # a prepared (c_void_p * n) array: no per-frame marshalling
            return self._feed(timestamps, 1, w, h, stride, pipelined, ptrs=dev_ptrs)
        for i, p in enumerate(dev_ptrs):
            self._ptrs[i] = int(p)
        return self._feed(timestamps, 1, w, h, stride, pipelined)

    def _feed(self, timestamps, on_device, w, h, stride, pipelined=False, ptrs=None):
        if isinstance(timestamps, np.ndarray) and timestamps.dtype == np.float64 and timestamps.flags.c_contiguous:
            ts = timestamps                   # a prepared array: used as it is
        else:
            self._ts[:] = timestamps
            ts = self._ts
        tracked, attempted = C.c_longlong(), C.c_longlong()
        fn = self.lib.ftk_batch_feed_pipelined if pipelined else self.lib.ftk_batch_feed
        self._check(fn(self._h, _dp(ts), self._ptrs if ptrs is None else ptrs, on_device, w, h, stride, C.byref(tracked), C.byref(attempted)), "ftk_batch_feed")
        return tracked.value, attempted.value

    def flush(self):
        tracked, attempted = C.c_longlong(), C.c_longlong()
        self._check(self.lib.ftk_batch_flush(self._h, C.byref(tracked), C.byref(attempted)), "ftk_batch_flush")
        return tracked.value, attempted.value

    def database_size(self, stream):
        return self.lib.ftk_batch_database_size(self._h, stream)

    def track(self, stream, feat_id):
        n = self.lib.ftk_batch_get_track(self._h, stream, int(feat_id), None, None, 0)
        uv, ts = np.zeros((max(n, 0), 4), dtype=np.float32), np.zeros(max(n, 0), dtype=np.float64)
        if n > 0:
            self.lib.ftk_batch_get_track(self._h, stream, int(feat_id), uv.ctypes.data, ts.ctypes.data, n)
        return uv, ts

    def groups(self):
        return self.lib.ftk_batch_groups(self._h)

    def group_ctx(self, g):
        return self.lib.ftk_batch_group_ctx(self._h, g)

    def points(self, stream):
        n = self.lib.ftk_batch_get_points(self._h, stream, None, None, 0)
        xy, ids = np.zeros((n, 2), dtype=np.float32), np.zeros(n, dtype=np.uint64)
        if n:
            self.lib.ftk_batch_get_points(self._h, stream, xy.ctypes.data, ids.ctypes.data, n)
        return xy, ids

    STAGES = ["frame_ingest", "pyramid_enqueue", "detection", "klt_prepare", "klt_enqueue", "device_wait", "db_update", "total"]

    def stage_times(self, reset=True):
        """Mean milliseconds per feed for every stage since the last reset."""
        out = np.zeros(8, dtype=np.float64)
        feeds = C.c_longlong()
        self._check(self.lib.ftk_batch_stage_times(self._h, _dp(out), C.byref(feeds), int(reset)), "ftk_batch_stage_times")
        n = max(1, feeds.value)
        return {k: 1e3 * v / n for k, v in zip(self.STAGES, out)}

    def device_ctx(self):
        return self.lib.ftk_batch_device_ctx(self._h)
