"""Seeded synthetic camera sequences (libft_synth.so) with matching gyro samples.

A stream's camera translates along a smooth path (<= ~6 px/frame) over a textured canvas and rolls about its
optical axis (<= 0.5 deg/frame); the gyro samples are the analytic roll rate at 200 Hz, so the gyro-predicted
flow carries a true signal.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_lib", "libft_synth.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run __graft_entry__.build()")
        L = C.CDLL(LIB_PATH)
        L.fts_canvas.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_int]
        L.fts_canvas.restype = None
        L.fts_canvas_blobs.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_double, C.c_double, C.c_int]
        L.fts_canvas_blobs.restype = None
        L.fts_frame.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_double, C.c_double,
                                C.c_double, C.c_double, C.c_uint32, C.c_int]
        L.fts_frame.restype = None
        _lib = L
    return _lib


def canvas(cw, ch, seed, n_waves=64, n_shapes=400, threads=8):
    out = np.empty((ch, cw), dtype=np.uint8)
    _load().fts_canvas(out.ctypes.data, cw, ch, seed, n_waves, n_shapes, threads)
    return out


def canvas_blobs(cw, ch, seed, n_waves=64, spacing=18.0, jitter=3.5, threads=8):
    """Canvas for the feature-dense configurations: a jittered lattice of small blobs, one strong corner each."""
    out = np.empty((ch, cw), dtype=np.uint8)
    _load().fts_canvas_blobs(out.ctypes.data, cw, ch, seed, n_waves, float(spacing), float(jitter), threads)
    return out


def frame(cnv, w, h, cx, cy, theta, sigma=2.0, noise_seed=0, threads=8, out=None):
    if out is None:
        out = np.empty((h, w), dtype=np.uint8)
    ch, cw = cnv.shape
    _load().fts_frame(cnv.ctypes.data, cw, ch, out.ctypes.data, w, h, out.strides[0], float(cx), float(cy), float(theta), float(sigma),
                      noise_seed, threads)
    return out


class Sequence:
    """One camera stream: frames, timestamps, gyro samples and calibration."""

    def __init__(self, w, h, n_frames, stream_id=0, fps=20.0, sigma=2.0, imu_hz=200.0, threads=8, shapes_per_mpx=280,
                 scene="shapes", blob_spacing=18.0, blob_jitter=3.5):
        """scene "shapes": random filled rectangles and discs (a few hundred corners per 752x480 frame);
        scene "blobs": a jittered lattice of small blobs for the configurations that track thousands of features."""
        self.w, self.h, self.n_frames, self.fps = w, h, n_frames, fps
        seed = 12345 + stream_id
        rng = np.random.default_rng(seed)
        cw, ch = 2 * w, 2 * h
        if scene == "blobs":
            cnv = canvas_blobs(cw, ch, seed, 64, blob_spacing, blob_jitter, threads)
        else:
            n_shapes = max(50, int(shapes_per_mpx * cw * ch / 1e6))
            cnv = canvas(cw, ch, seed, 64, n_shapes, threads)
        # smooth trajectory: two incommensurate sinusoids per axis, peak speed about 6 px/frame
        k = np.arange(n_frames)
        w1, w2 = 2 * np.pi / rng.uniform(55, 75), 2 * np.pi / rng.uniform(30, 45)
        p1, p2, p3, p4 = rng.uniform(0, 2 * np.pi, 4)
        vbound = 0.7 * w1 + 0.3 * w2                               # per-frame speed bound for unit amplitude
        ax, ay = min(0.22 * w, 6.0 / vbound), min(0.22 * h, 6.0 / vbound)   # fastest step <= 6 px, whatever n_frames
        cx = cw / 2 + ax * (0.7 * np.sin(w1 * k + p1) + 0.3 * np.sin(w2 * k + p2))
        cy = ch / 2 + ay * (0.7 * np.cos(w1 * k + p3) + 0.3 * np.sin(w2 * k + p4))
        # roll: theta(t) = A sin(w t); peak rate 0.5 deg/frame
        wr = 2 * np.pi / rng.uniform(50, 80)                      # rad/frame
        amp = np.deg2rad(0.5) / wr
        pr = rng.uniform(0, 2 * np.pi)
        self._roll = (amp, wr * fps, pr)                           # angle = amp*sin(w t + p), t in seconds
        self.timestamps = 1000.0 + k / fps
        theta = amp * np.sin(wr * k + pr)
        self.frames = np.empty((n_frames, h, w), dtype=np.uint8)
        for i in range(n_frames):
            frame(cnv, w, h, cx[i], cy[i], theta[i], sigma, (seed * 1000003 + i) & 0xFFFFFFFF, threads, out=self.frames[i])
        self.center = np.stack([cx, cy], axis=1)
        self.theta = theta
        # calibration: pinhole, no distortion, principal point at the image centre, camera frame = IMU frame
        f = 0.61 * w
        self.calib8 = np.array([f, f, (w - 1) / 2.0, (h - 1) / 2.0, 0, 0, 0, 0], dtype=np.float64)
        self.T_cam_imu = np.eye(4, dtype=np.float64)
        # gyro at imu_hz: the frame shows the canvas rotated by +theta, i.e. image content rotates by -theta:
        # the camera's angular velocity about +z (optical axis) is d(theta)/dt
        t0, t1 = self.timestamps[0] - 0.1, self.timestamps[-1] + 0.1
        self.imu_t = np.arange(t0, t1, 1.0 / imu_hz)
        rate = amp * (wr * fps) * np.cos(wr * fps * (self.imu_t - 1000.0) + pr)
        self.gyro = np.stack([np.zeros_like(rate), np.zeros_like(rate), rate], axis=1)
        self.accel = np.tile([0.0, 0.0, 9.81], (len(rate), 1))

    def true_position(self, xy, frame_from, frame_to):
        """Where the scene points seen at pixels xy (n, 2) in frame `frame_from` appear in frame `frame_to`: frame i
        shows the canvas point c_i + R(theta_i) (p - h) at pixel p, h = the frame centre."""
        h = np.array([(self.w - 1) * 0.5, (self.h - 1) * 0.5])

        def rot(t):
            return np.array([[np.cos(t), -np.sin(t)], [np.sin(t), np.cos(t)]])

        on_canvas = self.center[frame_from] + (np.asarray(xy, dtype=np.float64) - h) @ rot(self.theta[frame_from]).T
        return h + (on_canvas - self.center[frame_to]) @ rot(self.theta[frame_to])     # R(-theta) = R(theta)^T
