"""GPU end-to-end parity: whole tracking runs of the product tracker (gfx950 kernels behind TrackKLT /
TrackKLTBatch) against the same orchestration on the CPU checker, frame by frame."""
import os

import numpy as np
import pytest

import oracle_lib as O
import synth as tsynth
from feature_tracker_amd import synth, tracker

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC = os.path.join(ROOT, "oracle", "libft_oracle_tracker.so")


def _feed_all(b, seqs, n_frames):
    for s, q in enumerate(seqs):
        b.set_calibration(s, q.calib8, q.T_cam_imu)
        for t, g, a in zip(q.imu_t, q.gyro, q.accel):
            b.feed_imu(s, t, a, g)
    out = []
    for i in range(n_frames):
        tr, at = b.feed_host([q.timestamps[i] for q in seqs], [q.frames[i] for q in seqs])
        out.append((tr, at, [b.points(s) for s in range(len(seqs))]))
    return out


def _same_run(a, b):
    for (tr_a, at_a, pts_a), (tr_b, at_b, pts_b) in zip(a, b):
        assert at_a == at_b
        assert tr_a == tr_b                                   # status flags agree (bar: >= 99.9 %)
        for (xy_a, id_a), (xy_b, id_b) in zip(pts_a, pts_b):
            assert np.array_equal(id_a, id_b)
            assert np.abs(xy_a - xy_b).max(initial=0) <= 0.01  # px


@pytest.mark.parametrize("opts", [dict(win_size=21, use_gyro_seed=1), dict(win_size=15, speculative_detect=0, reject_outliers=0),
                                  dict(win_size=21, forward_backward=1, fb_thresh=1.0, pyr_levels=4, num_pts=400, grid_x=8, grid_y=5)])
def test_batch_tracker_matches_cpu_checker(opts):
    O.lib()
    seqs = [synth.Sequence(752, 480, 10, stream_id=s) for s in range(3)]
    gpu = _feed_all(tracker.BatchTracker(3, 752, 480, **opts), seqs, 10)
    cpu = _feed_all(tracker.BatchTracker(3, 752, 480, lib=O.checker_tracker_lib(), **opts), seqs, 10)
    _same_run(gpu, cpu)
    assert gpu[-1][0] > 250


@pytest.mark.parametrize("name,w,h,opts,min_pts", [
    # BASELINE configs[2]: 1280x720, 2000 features, maxLevel 4, forward-backward, grid 9x5 (cells of about 150 px)
    ("cfg3", 1280, 720, dict(num_pts=2000, win_size=21, pyr_levels=4, forward_backward=1, fb_thresh=1.0, grid_x=9, grid_y=5, use_gyro_seed=1), 1900),
    # BASELINE configs[4]: 3840x2160, 10k features, maxLevel 5, grid 26x15, replenishment forced on every frame
    ("cfg5", 3840, 2160, dict(num_pts=10000, win_size=21, pyr_levels=5, grid_x=26, grid_y=15, use_gyro_seed=1, drop_fraction=0.05), 9500),
])
def test_full_size_configurations_match_cpu_checker(name, w, h, opts, min_pts):
    """The two large configurations exactly as the benchmark runs them (same options, same scene, outlier rejection
    on), four frames: every frame's attempted and tracked counts, ids and positions equal the CPU checker's, and the
    number of points handed to LK stays at or above 95 % of num_pts."""
    O.lib()
    q = synth.Sequence(w, h, 4, stream_id=11, scene="blobs")
    import ctypes
    O.checker_tracker_lib().fto_set_threads(8)
    gpu = _feed_all(tracker.BatchTracker(1, w, h, **opts), [q], 4)
    cpu = _feed_all(tracker.BatchTracker(1, w, h, lib=O.checker_tracker_lib(), n_groups=1, host_threads=1, **opts), [q], 4)
    O.checker_tracker_lib().fto_set_threads(1)
    _same_run(gpu, cpu)
    for tr, at, _ in gpu[1:]:
        assert at >= min_pts and tr >= 0.9 * at
    if opts.get("drop_fraction"):
        # dropped tracks really are replaced: new ids appear on every frame
        for (_, _, a), (_, _, b) in zip(gpu[1:], gpu[2:]):
            assert len(np.setdiff1d(b[0][1], a[0][1])) >= 0.03 * min_pts


@pytest.mark.parametrize("opts", [dict(win_size=15), dict(win_size=21, use_gyro_seed=1)])
def test_tracks_follow_the_true_motion_on_the_device(opts):
    """Absolute accuracy of the product path against the motion the synthetic sequence was made with (same bars as the
    CPU checker's run in tests/test_tracker_cpu.py)."""
    from test_tracker_cpu import check_tracks_follow_the_true_motion
    seq = synth.Sequence(752, 480, 20, stream_id=0)
    b = tracker.BatchTracker(1, seq.w, seq.h, **opts)
    hist = [(tr, at, pts[0][0].copy(), pts[0][1].copy()) for tr, at, pts in _feed_all(b, [seq], 20)]
    check_tracks_follow_the_true_motion(seq, hist)


def test_batch_tracker_with_outlier_rejection_matches_cpu_checker():
    """Scenes with independently moving patches; the rejection kernel must drop exactly the features the CPU
    checker drops (same seeded hypotheses, same arithmetic), frame after frame."""
    O.lib()
    seqs = [synth.Sequence(752, 480, 8, stream_id=40 + s) for s in range(3)]
    for s, q in enumerate(seqs):
        q.frames = tsynth.with_moving_patches(q.frames, seed=s)
    opts = dict(win_size=21, use_gyro_seed=1, reject_outliers=1)
    gpu = _feed_all(tracker.BatchTracker(3, 752, 480, **opts), seqs, 8)
    cpu = _feed_all(tracker.BatchTracker(3, 752, 480, lib=O.checker_tracker_lib(), **opts), seqs, 8)
    _same_run(gpu, cpu)
    loose = _feed_all(tracker.BatchTracker(3, 752, 480, win_size=21, use_gyro_seed=1, reject_outliers=0), seqs, 2)
    assert gpu[1][1] == loose[1][1] and gpu[1][0] <= loose[1][0] - 20   # the patches' features went


@pytest.mark.parametrize("opts", [dict(), dict(reject_outliers=0), dict(stereo_extrinsic_seed=1, reject_outliers=0),
                                  dict(stereo_extrinsic_seed=1, stereo_threshold=0.4, reject_outliers=1)])
def test_stereo_tracker_matches_cpu_checker(opts):
    O.lib()
    q = synth.Sequence(640, 400, 8, stream_id=7)
    runs = []
    for lib in (None, O.checker_tracker_lib()):
        t = tracker.Tracker(640, 400, lib=lib, win_size=15, **opts)
        T_right = q.T_cam_imu.copy()
        T_right[0, 3] += 0.11                     # sideways baseline, used by the extrinsic seed / known-E test
        t.set_calibration(np.stack([q.calib8, q.calib8]), np.stack([q.T_cam_imu, T_right]))
        hist = []
        for i in range(8):
            t.feed_stereo(q.timestamps[i], q.frames[i], np.roll(q.frames[i], -5, axis=1))
            hist.append((t.points(0), t.points(1), t.database_size()))
        runs.append(hist)
    for (l_a, r_a, n_a), (l_b, r_b, n_b) in zip(*runs):
        assert n_a == n_b
        for (xy_a, id_a), (xy_b, id_b) in ((l_a, l_b), (r_a, r_b)):
            assert np.array_equal(id_a, id_b) and np.abs(xy_a - xy_b).max(initial=0) <= 0.01
    assert len(runs[0][-1][0][0]) > 60


def test_device_resident_frames_equal_host_upload():
    import torch
    q = synth.Sequence(752, 480, 6, stream_id=3)
    a = tracker.BatchTracker(1, 752, 480, win_size=21)
    b = tracker.BatchTracker(1, 752, 480, win_size=21)
    dev = torch.from_numpy(q.frames).cuda()
    torch.cuda.synchronize()
    for i in range(6):
        ra = a.feed_host([q.timestamps[i]], [q.frames[i]])
        rb = b.feed_device([q.timestamps[i]], [dev[i].data_ptr()], 752, 480, 752)
        assert ra == rb
        assert np.array_equal(a.points(0)[0], b.points(0)[0])


def _tracks(b, stream):
    xy, ids = b.points(stream)
    return {int(i): b.track(stream, int(i)) for i in ids}


def test_pipelined_groups_with_lookahead_equal_plain_feed():
    """Grouped, pipelined feeding stages the next frame set's ingest / pyramids / detector while the current set is
    tracked (three-slot ring).  Results and feature databases must equal the plain one-set-at-a-time feed."""
    n_streams, n_frames = 8, 14
    seqs = [synth.Sequence(376, 240, n_frames, stream_id=20 + s) for s in range(n_streams)]
    opts = dict(win_size=15, pyr_levels=2, num_pts=80, use_gyro_seed=1)

    def run(pipelined, n_groups):
        b = tracker.BatchTracker(n_streams, 376, 240, n_groups=n_groups, **opts)
        for s, q in enumerate(seqs):
            b.set_calibration(s, q.calib8, q.T_cam_imu)
            for t, g, a in zip(q.imu_t, q.gyro, q.accel):
                b.feed_imu(s, t, a, g)
        tracked = attempted = 0
        for i in range(n_frames):
            tr, at = b.feed_host([q.timestamps[i] for q in seqs], [q.frames[i] for q in seqs], pipelined=pipelined)
            tracked += tr
            attempted += at
        if pipelined:
            tr, at = b.flush()
            tracked += tr
            attempted += at
        pts = [b.points(s) for s in range(n_streams)]
        dbs = [(b.database_size(s), _tracks(b, s)) for s in range(n_streams)]
        return tracked, attempted, pts, dbs

    plain = run(False, 1)
    piped = run(True, 2)
    assert plain[0] == piped[0] and plain[1] == piped[1] and plain[0] > 0
    for (xy_a, id_a), (xy_b, id_b) in zip(plain[2], piped[2]):
        assert np.array_equal(id_a, id_b) and np.array_equal(xy_a, xy_b)
    for (n_a, tr_a), (n_b, tr_b) in zip(plain[3], piped[3]):
        assert n_a == n_b and tr_a.keys() == tr_b.keys()
        for k in tr_a:
            assert np.array_equal(tr_a[k][0], tr_b[k][0]) and np.array_equal(tr_a[k][1], tr_b[k][1])


def test_large_stream_fed_pipelined_equals_plain_feed(monkeypatch):
    """One stream of 2 000 features fed pipelined runs its next frame's ingest, pyramids and detector on a device queue
    of their own, beside the tracking (forward and backward pass in one launch).  Counts, points and feature database
    equal the plain feed's, with that queue and (FT_FRONT_QUEUE=0) without it."""
    w, h, n_frames = 1280, 720, 8
    q = synth.Sequence(w, h, n_frames, stream_id=31, scene="blobs")
    opts = dict(num_pts=2000, win_size=21, pyr_levels=4, forward_backward=1, fb_thresh=1.0, grid_x=9, grid_y=5, use_gyro_seed=1)

    def run(pipelined):
        b = tracker.BatchTracker(1, w, h, **opts)
        b.set_calibration(0, q.calib8, q.T_cam_imu)
        for t, g, a in zip(q.imu_t, q.gyro, q.accel):
            b.feed_imu(0, t, a, g)
        tracked = attempted = 0
        for i in range(n_frames):
            tr, at = b.feed_host([q.timestamps[i]], [q.frames[i]], pipelined=pipelined)
            tracked += tr
            attempted += at
        if pipelined:
            tr, at = b.flush()
            tracked += tr
            attempted += at
        return tracked, attempted, b.points(0), b.database_size(0), _tracks(b, 0)

    plain = run(False)
    piped = run(True)
    monkeypatch.setenv("FT_FRONT_QUEUE", "0")
    one_queue = run(True)
    monkeypatch.delenv("FT_FRONT_QUEUE")
    assert plain[0] > 10000 and plain[1] >= 1900 * (n_frames - 1)
    for other in (piped, one_queue):
        assert other[0] == plain[0] and other[1] == plain[1] and other[3] == plain[3]
        assert np.array_equal(other[2][0], plain[2][0]) and np.array_equal(other[2][1], plain[2][1])
        assert other[4].keys() == plain[4].keys()
        for k in plain[4]:
            assert np.array_equal(other[4][k][0], plain[4][k][0]) and np.array_equal(other[4][k][1], plain[4][k][1])


def test_late_detector_with_streams_that_lose_tracking():
    """One-set-at-a-time feeds queue the detector after the tracking launch and collect it at the next feed; a stream
    that lost everything (blank frames) needs the detection of its new frame at once.  Must equal the CPU checker fed
    the same way and the pipelined feed (detector before tracking), frame by frame."""
    O.lib()
    n_frames = 12
    seqs = [synth.Sequence(376, 240, n_frames, stream_id=60 + s) for s in range(3)]
    seqs[0].frames[4:6] = 128          # stream 0 goes blank for two frames, then the scene is back
    seqs[2].frames[7] = 10             # stream 2 for one
    opts = dict(win_size=15, pyr_levels=2, num_pts=80, use_gyro_seed=1)
    gpu = _feed_all(tracker.BatchTracker(3, 376, 240, **opts), seqs, n_frames)
    cpu = _feed_all(tracker.BatchTracker(3, 376, 240, lib=O.checker_tracker_lib(), **opts), seqs, n_frames)
    _same_run(gpu, cpu)
    assert len(gpu[5][2][0][0]) == 0 and len(gpu[7][2][0][0]) > 40      # lost, then found again
    # pipelined, two groups: the detector runs before the tracking launch there
    b = tracker.BatchTracker(3, 376, 240, n_groups=2, **opts)
    for s, q in enumerate(seqs):
        b.set_calibration(s, q.calib8, q.T_cam_imu)
        for t, g, a in zip(q.imu_t, q.gyro, q.accel):
            b.feed_imu(s, t, a, g)
    for i in range(n_frames):
        b.feed_host([q.timestamps[i] for q in seqs], [q.frames[i] for q in seqs], pipelined=True)
    b.flush()
    for s in range(3):
        xy, ids = b.points(s)
        assert np.array_equal(ids, gpu[-1][2][s][1]) and np.array_equal(xy, gpu[-1][2][s][0])
