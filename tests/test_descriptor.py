"""Descriptor tracker (reference: src/track/TrackDescriptor.cpp): the build's BRIEF-style descriptor against an
independent NumPy statement and known answers, the TrackDescriptor orchestration on the CPU checker backend, and on the
GPU both against the checker.  Parity unpinned: cv::ORB's sampling table is not available, the descriptor is build-defined."""
import numpy as np
import pytest

import oracle_lib as O
import synth
from feature_tracker_amd import synth as seq_synth
from feature_tracker_amd import tracker


def _mix32(x):
    x &= 0xFFFFFFFF
    x ^= x >> 16; x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15; x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


PATTERN = np.array([[_mix32(0xB5297A4D + 4 * i + k) % 31 - 15 for k in range(4)] for i in range(256)], dtype=np.int32)   # x1 y1 x2 y2


def np_descriptors(img, xy):
    k = np.array([1, 4, 6, 4, 1], dtype=np.int64)
    p = img.astype(np.int64)
    hp = sum(k[i] * p[:, i:p.shape[1] - 4 + i] for i in range(5))
    sm = (sum(k[j] * hp[j:hp.shape[0] - 4 + j] for j in range(5)) + 128) >> 8          # sm[y, x] = smoothed(x + 2, y + 2)
    h, w = img.shape
    desc, valid = np.zeros((len(xy), 32), dtype=np.uint8), np.zeros(len(xy), dtype=np.uint8)
    for n, (x, y) in enumerate(np.asarray(xy).astype(np.int64)):
        if x < 17 or y < 17 or x >= w - 17 or y >= h - 17:
            continue
        valid[n] = 1
        a = sm[y + PATTERN[:, 1] - 2, x + PATTERN[:, 0] - 2]
        b = sm[y + PATTERN[:, 3] - 2, x + PATTERN[:, 2] - 2]
        desc[n] = np.packbits((a < b).astype(np.uint8), bitorder="little")
    return desc, valid


def test_descriptor_known_answers_and_border_rule():
    w, h = 120, 90
    ramp = np.tile(np.arange(w, dtype=np.uint8) * 2, (h, 1))          # smoothed value = 2 x: bit i = (x1 < x2)
    pts = np.array([[60, 45], [17, 17], [16, 45], [60, 16], [w - 18, h - 18], [w - 17, 45], [60.9, 45.2]], dtype=np.float32)
    desc, valid = O.compute_descriptors(ramp, pts)
    assert valid.tolist() == [1, 1, 0, 0, 1, 0, 1]
    want = np.packbits((PATTERN[:, 0] < PATTERN[:, 2]).astype(np.uint8), bitorder="little")
    for i in (0, 1, 4, 6):
        assert np.array_equal(desc[i], want)
    assert not desc[2].any() and not desc[3].any() and not desc[5].any()
    assert np.array_equal(O.compute_descriptors(ramp.T.copy(), pts[:1, ::-1].copy())[0][0],
                          np.packbits((PATTERN[:, 1] < PATTERN[:, 3]).astype(np.uint8), bitorder="little"))
    # the pattern fills the 31x31 patch and is balanced enough to discriminate
    assert PATTERN.min() == -15 and PATTERN.max() == 15 and 60 < int((PATTERN[:, 0] < PATTERN[:, 2]).sum()) < 196


def test_descriptor_matches_independent_numpy_and_matches_itself_under_noise():
    img = synth.texture(320, 240, seed=5, n_shapes=300)
    kp = O.grid_fast(img, 300, 6, 4, 20)
    pts = np.concatenate([kp[:, :2], [[3, 3], [319, 100], [160, 239]]]).astype(np.float32)
    desc, valid = O.compute_descriptors(img, pts)
    d_np, v_np = np_descriptors(img, pts)
    assert np.array_equal(valid, v_np) and np.array_equal(desc, d_np) and valid.sum() > 100 and valid[-3:].sum() == 0
    noisy = np.clip(img.astype(np.int16) + np.random.default_rng(1).integers(-4, 5, img.shape), 0, 255).astype(np.uint8)
    d2, _ = O.compute_descriptors(noisy, pts)
    ok = valid == 1
    m = O.match_descriptors(desc[ok], d2[ok], knn_ratio=0.85)
    assert len(m) > 0.6 * ok.sum() and (m[:, 0] == m[:, 1]).mean() > 0.97       # the same keypoint is its own best match


def _stereo_frames(n):
    q = seq_synth.Sequence(640, 400, n, stream_id=7)
    return q, [q.frames[i] for i in range(n)], [np.roll(q.frames[i], -4, axis=1) for i in range(n)]


def _run_descriptor_tracker(lib, n=6, **opts):
    q, left, right = _stereo_frames(n)
    t = tracker.Tracker(640, 400, lib=lib, use_klt=0, num_pts=300, fast_threshold=20, grid_x=8, grid_y=5, knn_ratio=0.8, **opts)
    t.set_calibration(np.stack([q.calib8, q.calib8]), np.stack([q.T_cam_imu, q.T_cam_imu]))
    hist = []
    for i in range(n):
        t.feed_stereo(float(q.timestamps[i]), left[i], right[i])
        hist.append((t.points(0), t.points(1), t.database_size()))
    return hist, t


def test_descriptor_tracker_keeps_ids_across_frames_on_the_checker_backend():
    hist, t = _run_descriptor_tracker(O.checker_tracker_lib())
    for (xy_l, id_l), (xy_r, id_r), _ in hist:
        assert len(id_l) == len(id_r) > 60 and np.array_equal(id_l, id_r)                 # stereo pairs share their id
        right_place = (np.abs((xy_l[:, 0] - xy_r[:, 0]) - 4) <= 1.0) & (np.abs(xy_l[:, 1] - xy_r[:, 1]) <= 1.0)   # 4 px disparity
        assert right_place.mean() > 0.95
    kept = [len(np.intersect1d(a[0][1], b[0][1])) / len(b[0][1]) for a, b in zip(hist, hist[1:])]
    assert min(kept) > 0.4, kept                                                          # tracks continue under their old ids
    assert hist[-1][2] > hist[1][2] >= len(hist[1][0][1])                                 # the database grows with new features
    some = int(np.intersect1d(hist[-1][0][1], hist[-2][0][1])[0])
    uv, ts = t.track(some, 0)
    assert len(ts) >= 2 and np.all(np.diff(ts) > 0)
    with pytest.raises(RuntimeError):
        t.feed_monocular(0.0, np.zeros((400, 640), np.uint8))                             # the descriptor tracker is stereo only
    # the rejection on the matched pairs (the evident intent of the reference's RANSAC step) only ever removes matches
    strict, _ = _run_descriptor_tracker(O.checker_tracker_lib(), descriptor_ransac_on_pairs=1)
    assert all(len(s[0][1]) <= len(h[0][1]) for s, h in zip(strict, hist)) and len(strict[-1][0][1]) > 40


@pytest.mark.gpu
def test_gpu_descriptors_equal_oracle():
    from feature_tracker_amd import device as D
    for w, h, seed in ((752, 480, 3), (321, 243, 4), (40, 36, 5)):
        img = synth.texture(w, h, seed=seed, n_shapes=200)
        rng = np.random.default_rng(seed)
        pts = np.stack([rng.integers(-2, w + 2, 500), rng.integers(-2, h + 2, 500)], axis=1).astype(np.float32)
        want_d, want_v = O.compute_descriptors(img, pts)
        with D.DeviceContext(w, h, max_levels=1, n_streams=1, n_slots=1) as ctx:
            ctx.upload(0, 0, img)
            d, v = ctx.compute_descriptors(0, 0, pts)
            assert np.array_equal(v, want_v) and np.array_equal(d, want_d)
            assert len(ctx.compute_descriptors(0, 0, pts[:0])[0]) == 0
        assert want_v.sum() > (0 if w < 50 else 300)


@pytest.mark.gpu
@pytest.mark.parametrize("opts", [dict(), dict(descriptor_ransac_on_pairs=1)])
def test_gpu_descriptor_tracker_equals_checker(opts):
    gpu, _ = _run_descriptor_tracker(None, **opts)
    cpu, _ = _run_descriptor_tracker(O.checker_tracker_lib(), **opts)
    for (gl, gr, gdb), (cl, cr, cdb) in zip(gpu, cpu):
        assert np.array_equal(gl[1], cl[1]) and np.array_equal(gr[1], cr[1]) and gdb == cdb
        assert np.array_equal(gl[0], cl[0]) and np.array_equal(gr[0], cr[0])
    assert len(gpu[-1][0][1]) > 60
