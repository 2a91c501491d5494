"""The product's stereo orchestration against an independent Python restatement of the reference's
TrackKLT::feed_stereo (tests/orchestration_ref.py, written from src/track/TrackKLT.cpp:27-432 of the reference and
driven by kernel-level oracle calls): active ids, positions and the feature database, frame by frame."""
import numpy as np
import pytest

import oracle_lib as O
from orchestration_ref import StereoTrackerRef
from feature_tracker_amd import synth, tracker

W, H = 640, 400


def _frames(n, blank=()):
    q = synth.Sequence(W, H, n, stream_id=7)
    left = [q.frames[i] if i not in blank else np.full((H, W), 97, np.uint8) for i in range(n)]
    right = [np.roll(f, -4, axis=1) for f in left]          # a fronto-parallel plane: 4 px of disparity
    return q, left, right


def _compare(lib, n_frames, blank=(), pos_tol=0.0, **opts):
    q, left, right = _frames(n_frames, blank)
    calib = np.stack([q.calib8, q.calib8 + np.array([1.5, -2.0, 3.0, -1.0, -0.05, 0.01, 1e-4, -2e-4])])
    prod = tracker.Tracker(W, H, lib=lib, win_size=15, **opts)
    prod.set_calibration(calib, np.stack([q.T_cam_imu, q.T_cam_imu]))
    ref = StereoTrackerRef(num_features=opts.get("num_pts", 150), min_px_dist=opts.get("min_px_dist", 10), win=15, levels=3, calib=calib,
                           reject_outliers=bool(opts.get("reject_outliers", 1)))
    counts = []
    for i in range(n_frames):
        t = float(q.timestamps[i])
        prod.feed_stereo(t, left[i], right[i])
        ref.feed_stereo(t, left[i], right[i])
        for cam in (0, 1):
            xy_p, id_p = prod.points(cam)
            xy_r, id_r = ref.points(cam)
            assert np.array_equal(id_p, id_r), f"frame {i} camera {cam}: active ids differ"
            assert np.abs(xy_p - xy_r).max(initial=0) <= pos_tol, f"frame {i} camera {cam}: positions differ"
        assert prod.database_size() == len(ref.database), f"frame {i}: database sizes differ"
        counts.append((len(ref.ids[0]), len(ref.ids[1]), len(np.intersect1d(ref.ids[0], ref.ids[1]))))
    # every feature's observations: same timestamps, pixels and normalized coordinates
    for fid in list(ref.database)[::7]:
        for cam, obs in ref.database[fid].items():
            uv, ts = prod.track(fid, cam)
            want = np.array(obs, dtype=np.float64)
            assert np.array_equal(ts, want[:, 0])
            assert np.abs(uv - want[:, 1:]).max() <= max(pos_tol, 1e-6)
    return counts


@pytest.mark.parametrize("opts", [dict(), dict(speculative_detect=0, reject_outliers=0), dict(num_pts=60, min_px_dist=25)])
def test_stereo_orchestration_equals_python_restatement_of_the_reference(opts):
    counts = _compare(O.checker_tracker_lib(), 6, **opts)
    n = opts.get("num_pts", 150)
    assert counts[0][0] >= 0.5 * n and counts[0][2] >= 0.3 * n      # features, and stereo features among them
    assert counts[-1][2] >= 0.25 * n                                  # stereo tracks survive six frames


def test_stereo_orchestration_reset_and_redetection_equal_python_restatement():
    """A textureless pair in the middle: nearly every track is lost there, the next pair has fewer than ten points per
    camera and gets all-zero masks (src/track/TrackKLT.cpp:392-396), and the pair after that starts from nothing and
    detects on the current images like a first call (:62-72)."""
    counts = _compare(O.checker_tracker_lib(), 7, blank=(3,))
    assert counts[2][0] > 60 and 0 < counts[3][0] < 10 and counts[4][0] == 0 and counts[5][0] > 60 and counts[6][2] > 30


@pytest.mark.gpu
@pytest.mark.parametrize("opts,blank", [(dict(), ()), (dict(use_gyro_seed=0, num_pts=100), (2,))])
def test_gpu_stereo_orchestration_equals_python_restatement_of_the_reference(opts, blank):
    """The same comparison with the gfx950 kernels behind the product tracker (positions to the parity bar of 0.01 px;
    in practice they are identical)."""
    counts = _compare(None, 6, blank=blank, pos_tol=0.01, **opts)
    assert counts[-1][0] > 40
