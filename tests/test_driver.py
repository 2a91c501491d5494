"""The stereo driver program (feature_tracker_amd/src/test_tracking.cpp, flow of the reference's
src/test_tracking.cpp:220-246): calibration file reader, dataset readers, and whole runs.

CPU: the driver built on the CPU checker must give what the Python-driven tracker gives on the same files.
GPU: the product driver (gfx950 kernels) must give what the checker-backed driver gives, frame by frame."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from feature_tracker_amd import tracker

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECKER = os.path.join(ROOT, "oracle", "test_tracking_checker")
PRODUCT = os.path.join(ROOT, "feature_tracker_amd", "_lib", "test_tracking")
ORC = os.path.join(ROOT, "oracle", "libft_oracle_tracker.so")

# a calibration file in the layout of the reference's build_pc/config/test.yaml (comments, `data:[` without a
# blank, a matrix spread over several lines)
YAML = """%YAML:1.0

#--------------------------------------------------------------------------------------------
# Stereo rig
#--------------------------------------------------------------------------------------------
LEFT.height: 640
LEFT.width: 480
LEFT.D: !!opencv-matrix
   rows: 1
   cols: 5
   dt: d
   data:[0.01, -0.02, 0.003, 0.004, 0.0]
LEFT.K: !!opencv-matrix
   rows: 3
   cols: 3
   dt: d
   data: [500.5, 0.0, 240.25, 0.0, 501.5, 320.75, 0.0, 0.0, 1.0]
LEFT.R:  !!opencv-matrix
   rows: 4
   cols: 4
   dt: d
   data: [
          0.0, 0.0, 1.0, 0.001,
          1.0, 0.0, 0.0, -0.002,   # a comment inside the list
          0.0, 1.0, 0.0, 0.003,
          0.0, 0.0, 0.0, 1.0
         ]

RIGHT.height: 640
RIGHT.width: 480
RIGHT.D: !!opencv-matrix
   rows: 1
   cols: 5
   dt: d
   data:[-0.0, 0.0, 0.0, 0.0, 0.0]
RIGHT.K: !!opencv-matrix
   rows: 3
   cols: 3
   dt: d
   data: [510.0, 0.0, 241.0, 0.0, 511.0, 311.0, 0.0, 0.0, 1]
RIGHT.R:  !!opencv-matrix
   rows: 4
   cols: 4
   dt: d
   data: [ 0.0, 0.0, 1.0, 0.0005, 1.0, 0.0, 0.0, 0.0585, 0.0, 1.0, 0.0, -0.0017, 0.0, 0.0, 0.0, 1.0 ]
"""


def _run(exe, *args):
    r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=300)
    return r.returncode, r.stdout + r.stderr


def _read_pgm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"P5"
        w, h = map(int, f.readline().split())
        assert f.readline().strip() == b"255"
        return np.frombuffer(f.read(), dtype=np.uint8).reshape(h, w)


def _read_dump(path):
    """-> list of (timestamp, {cam: (ids, xy)})"""
    frames = []
    with open(path) as f:
        for line in f:
            p = line.split()
            if p[0] == "frame":
                frames.append((float(p[2]), {0: ([], []), 1: ([], [])}))
            else:
                ids, xy = frames[-1][1][int(p[0])]
                ids.append(int(p[1]))
                xy.append((float(p[2]), float(p[3])))
    return [(t, {c: (np.array(i, dtype=np.uint64), np.array(x, dtype=np.float32).reshape(-1, 2)) for c, (i, x) in cams.items()})
            for t, cams in frames]


def _dataset(tmp_path, w=320, h=240, frames=6):
    d = str(tmp_path / "ds")
    rc, out = _run(CHECKER, "--write-dataset", d, "--size", str(w), str(h), "--frames", str(frames), "--disparity", "4")
    assert rc == 0, out
    return d


def _same(a, b):
    assert len(a) == len(b) and len(a) > 0
    for (ta, ca), (tb, cb) in zip(a, b):
        assert ta == tb
        for cam in (0, 1):
            assert np.array_equal(ca[cam][0], cb[cam][0])                                  # ids (status flags) agree
            assert np.abs(ca[cam][1] - cb[cam][1]).max(initial=0) <= 0.01                  # px


def test_calibration_file_reader(tmp_path):
    p = tmp_path / "rig.yaml"
    p.write_text(YAML)
    rc, out = _run(CHECKER, "--check-config", str(p))
    assert rc == 0, out
    assert "cam_0_wh: 480 x 640" in out and "cam_1_wh: 480 x 640" in out
    assert "cam_0_intrinsic: 500.5 501.5 240.25 320.75 | 0.01 -0.02 0.003 0.004" in out
    assert "cam_1_intrinsic: 510 511 241 311 | -0 0 0 0" in out
    assert "cam_0_T_cam_imu: 0 0 1 0.001 1 0 0 -0.002 0 1 0 0.003 0 0 0 1" in out
    assert "cam_1_T_cam_imu: 0 0 1 0.0005 1 0 0 0.0585 0 1 0 -0.0017 0 0 0 1" in out


def test_missing_files_are_reported_like_the_reference(tmp_path):
    rc, out = _run(CHECKER, "--check-config", str(tmp_path / "nope.yaml"))
    assert rc != 0 and "ERROR: Wrong path to settings" in out                   # src/test_tracking.cpp:33
    p = tmp_path / "half.yaml"
    p.write_text(YAML.split("RIGHT.height")[0])
    rc, out = _run(CHECKER, "--check-config", str(p))
    assert rc != 0 and "ERROR: Calibration parameters to rectify stereo are missing!" in out   # :55
    d = _dataset(tmp_path, frames=1)
    os.remove(os.path.join(d, "imu.txt"))
    os.remove(os.path.join(d, "timestamp.txt"))
    rc, out = _run(CHECKER, "--data", d)
    assert "imu data path is not open!" in out and "Image path is not open!" in out            # :127, :167


@pytest.mark.parametrize("flags,opts", [((), {}), (("--gyro-seed", "--win", "21"), dict(use_gyro_seed=1, win_size=21))])
def test_checker_driver_equals_python_driven_tracker(tmp_path, flags, opts):
    d = _dataset(tmp_path)
    dump = str(tmp_path / "dump.txt")
    rc, out = _run(CHECKER, "--data", d, "--dump", dump, *flags)
    assert rc == 0 and "backend: cpu-oracle" in out, out
    run = _read_dump(dump)

    # the same files through the C interface of the tracker
    t = tracker.Tracker(320, 240, lib=O.checker_tracker_lib(), **opts)
    f = 0.61 * 320
    k = [f, f, 159.5, 119.5, 0, 0, 0, 0]
    T0, T1 = np.eye(4), np.eye(4)
    T1[0, 3] = 0.11
    t.set_calibration(np.array([k, k]), np.stack([T0, T1]))
    for line in open(os.path.join(d, "imu.txt")):
        v = [float(x) for x in line.split()]
        t.feed_imu(v[0] * 1e-6, v[1:4], v[4:7])
    mine = []
    for line in open(os.path.join(d, "timestamp.txt")):
        us = int(line)
        t.feed_stereo(us * 1e-6, _read_pgm(f"{d}/left/{us}.pgm"), _read_pgm(f"{d}/right/{us}.pgm"))
        mine.append((round(us * 1e-6, 6), {0: t.points(0)[::-1], 1: t.points(1)[::-1]}))
    _same(run, mine)
    assert len(run[-1][1][0][0]) > 40
    assert f"database: {t.database_size()} features" in out


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [(), ("--gyro-seed", "--win", "21", "--forward-backward"), ("--reject-outliers", "--stereo-extrinsic-seed"), ("--no-reject-outliers",)])
def test_product_driver_equals_checker_driver(tmp_path, flags):
    d = _dataset(tmp_path, 752, 480, 8)
    a, b = str(tmp_path / "gpu.txt"), str(tmp_path / "cpu.txt")
    rc, out = _run(PRODUCT, "--data", d, "--dump", a, *flags)
    assert rc == 0 and "backend: hip-gfx950" in out, out
    rc, out_c = _run(CHECKER, "--data", d, "--dump", b, *flags)
    assert rc == 0, out_c
    _same(_read_dump(a), _read_dump(b))
    assert out.splitlines()[-1] == out_c.splitlines()[-1]          # the database summary line
    assert len(_read_dump(a)[-1][1][0][0]) > 100
