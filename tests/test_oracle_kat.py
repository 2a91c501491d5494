"""Known-answer tests of the CPU oracle (oracle/ft_oracle.cpp).

The reference holds no golden vectors for this path (it has no tests at all) and OpenCV is absent, so the
oracle is pinned against analytic answers and against independent numpy restatements of the published filters.
"""
import numpy as np
import pytest

import oracle_lib as O
import synth


def np_pyr_down(img):
    """Independent restatement: 5x5 [1 4 6 4 1]^2 on a REFLECT_101-padded image, decimate, (s+128)>>8."""
    h, w = img.shape
    k = np.array([1, 4, 6, 4, 1], dtype=np.int64)
    pad = np.pad(img.astype(np.int64), 2, mode="reflect") if min(h, w) > 2 else None
    if pad is None:   # tiny images: explicit index map
        idx = lambda p, n: 0 if n == 1 else (abs(p) if p < n else 2 * n - 2 - p)
        fix = lambda p, n: idx(idx(p, n), n) if n > 1 else 0
        pad = np.array([[img[fix(y, h), fix(x, w)] for x in range(-2, w + 2)] for y in range(-2, h + 2)], dtype=np.int64)
    dh, dw = (h + 1) // 2, (w + 1) // 2
    out = np.zeros((dh, dw), dtype=np.int64)
    for j in range(5):
        for i in range(5):
            out += k[j] * k[i] * pad[j:j + 2 * dh:2, i:i + 2 * dw:2][:dh, :dw]
    return ((out + 128) >> 8).astype(np.uint8)


def np_scharr(img):
    p = np.pad(img.astype(np.int64), 1, mode="reflect")
    t0 = (p[:-2, :] + p[2:, :]) * 3 + p[1:-1, :] * 10     # vertical smooth, full padded width
    t1 = p[2:, :] - p[:-2, :]                             # vertical difference
    ix = t0[:, 2:] - t0[:, :-2]
    iy = (t1[:, 2:] + t1[:, :-2]) * 3 + t1[:, 1:-1] * 10
    return np.stack([ix, iy], axis=-1).astype(np.int16)


@pytest.mark.parametrize("shape", [(480, 752), (33, 47), (60, 94), (7, 9), (135, 240), (5, 64)])
def test_pyr_down_matches_numpy(shape):
    img = np.random.default_rng(1).integers(0, 256, shape, dtype=np.uint8)
    assert np.array_equal(O.pyr_down(img), np_pyr_down(img))


def test_pyr_down_constant_and_impulse():
    c = np.full((40, 56), 77, dtype=np.uint8)
    assert np.all(O.pyr_down(c) == 77)
    imp = np.zeros((40, 56), dtype=np.uint8)
    imp[20, 30] = 255                      # even coordinates: centre of the kernel of output (15, 10)
    d = O.pyr_down(imp)
    k = np.array([1, 6, 1])               # an even source sample meets taps 1, 6, 1 of outputs i-1, i, i+1
    expect = (255 * np.outer(k, k) + 128) >> 8
    assert np.array_equal(d[9:12, 14:17], expect)
    assert d.sum() == expect.sum()        # nothing outside the 3x3 neighbourhood
    imp2 = np.zeros((40, 56), dtype=np.uint8)
    imp2[21, 31] = 255                    # odd coordinates: tap 4 of the two nearest outputs on each axis
    d2 = O.pyr_down(imp2)
    assert np.array_equal(d2[10:12, 15:17], np.full((2, 2), (255 * 16 + 128) >> 8)) and d2.sum() == 4 * ((255 * 16 + 128) >> 8)


def test_pyr_down_one_pixel_high():
    row = np.arange(0, 200, 5, dtype=np.uint8)[None, :]
    assert np.array_equal(O.pyr_down(row), np_pyr_down(row))


def test_pyramid_levels_and_early_stop():
    img = synth.texture(752, 480, seed=3)
    p = O.Pyramid(img, win=21, max_level=3)
    assert p.levels == 4
    assert [p.level(l).shape for l in range(4)] == [(480, 752), (240, 376), (120, 188), (60, 94)]
    assert np.array_equal(p.level(0), img)
    assert np.array_equal(p.level(1), np_pyr_down(img))
    assert np.array_equal(p.level(2), np_pyr_down(np_pyr_down(img)))
    # buildOpticalFlowPyramid stops when the next level would not exceed the window
    small = synth.texture(100, 80, seed=4)
    q = O.Pyramid(small, win=21, max_level=5)      # 100x80 -> 50x40 -> 25x20 (20 <= 21: stop after level 1)
    assert q.levels == 2
    # REFLECT_101 padding of the level itself
    b = p.level(3, with_border=True)
    assert np.array_equal(b, np.pad(p.level(3), 21, mode="reflect"))


@pytest.mark.parametrize("shape", [(60, 94), (480, 752), (2, 2), (1, 9), (9, 1), (31, 17)])
def test_scharr_matches_numpy(shape):
    img = np.random.default_rng(2).integers(0, 256, shape, dtype=np.uint8)
    if min(shape) == 1:
        # numpy's reflect pad of a length-1 axis is undefined; the rule there is "use the only row/column"
        g = O.scharr(img)
        if shape[0] == 1:
            assert np.all(g[..., 1] == 0)
        else:
            assert np.all(g[..., 0] == 0)
        return
    assert np.array_equal(O.scharr(img), np_scharr(img))


def test_scharr_ramp():
    img = np.tile(np.arange(64, dtype=np.uint8), (20, 1))          # I = x
    g = O.scharr(img)
    assert np.all(g[:, 1:-1, 0] == 32) and np.all(g[..., 1] == 0)   # gain 32, no vertical gradient
    assert np.all(g[:, 0, 0] == 0) and np.all(g[:, -1, 0] == 0)     # REFLECT_101 at the edges


def test_lk_integer_shift_converges():
    prev, nxt = synth.shifted_pair(320, 240, dx=3, dy=-2, seed=5)
    pts = synth.random_points(320, 240, 200, seed=6, border=30)
    pp, pn = O.Pyramid(prev, 21, 3), O.Pyramid(nxt, 21, 3)
    out, st, err = O.lk_track(pp, pn, pts, win=21, max_level=3)
    ok = st == 1
    assert ok.mean() > 0.97
    d = out[ok] - pts[ok] - np.array([3, -2], dtype=np.float32)
    assert np.abs(d).max() < 0.05
    assert err[ok].max() < 2.0


def test_lk_subpixel_shift_analytic():
    f = synth.smooth_field(256, 192, seed=7)
    prev = synth.render(f, 256, 192)
    nxt = synth.render(f, 256, 192, dx=1.37, dy=-0.81)
    pts = synth.random_points(256, 192, 100, seed=8, border=30)
    pp, pn = O.Pyramid(prev, 15, 3), O.Pyramid(nxt, 15, 3)
    out, st, _ = O.lk_track(pp, pn, pts, win=15, max_level=3)
    ok = st == 1
    assert ok.mean() > 0.9
    d = out[ok] - pts[ok] - np.array([1.37, -0.81], dtype=np.float32)
    assert np.median(np.abs(d)) < 0.03 and np.abs(d).max() < 0.15   # quantisation-limited sanity bound


def test_lk_flat_patch_and_out_of_image():
    img = np.full((120, 160), 90, dtype=np.uint8)
    img[:, 100:] = synth.texture(60, 120, seed=9)
    p = O.Pyramid(img, 15, 2)
    pts = np.array([[30.0, 60.0], [-40.0, 50.0], [400.0, 50.0]], dtype=np.float32)
    out, st, err = O.lk_track(p, p, pts, win=15, max_level=2)
    assert st.tolist() == [0, 0, 0]                                 # min-eigenvalue fail, outside, outside
    assert err[1] == 0 and err[2] == 0


def test_lk_status_written_only_at_level0_and_guess_propagates():
    # a point whose window is outside the coarse levels' bound but inside level 0 keeps status 1 when level 0 works
    prev, nxt = synth.shifted_pair(200, 150, dx=1, dy=1, seed=10)
    pp, pn = O.Pyramid(prev, 15, 3), O.Pyramid(nxt, 15, 3)
    pts = np.array([[100.0, 75.0]], dtype=np.float32)
    out, st, _, it = O.lk_track(pp, pn, pts, win=15, max_level=3, want_iters=True)
    assert st[0] == 1 and it.shape == (1, 4) and np.all(it[0] >= 1)
    # zero iterations: the guess comes back unchanged at every level (x 2^-L then x2 back)
    g = np.array([[101.5, 76.25]], dtype=np.float32)
    out0, st0, _ = O.lk_track(pp, pn, pts, guess=g, win=15, max_level=3, max_iter=0)
    assert np.allclose(out0, g, atol=1e-4) and st0[0] == 1


def test_lk_forward_backward_rejects_inconsistent():
    prev, nxt = synth.shifted_pair(320, 240, dx=2, dy=1, seed=11)
    nxt = nxt.copy()
    nxt[100:140, 150:200] = np.random.default_rng(0).integers(0, 256, (40, 50), dtype=np.uint8)   # occlusion
    pts = synth.random_points(320, 240, 300, seed=12, border=25)
    pp, pn = O.Pyramid(prev, 21, 3), O.Pyramid(nxt, 21, 3)
    _, st_plain, _ = O.lk_track(pp, pn, pts, win=21)
    out, st_fb, _ = O.lk_track(pp, pn, pts, win=21, flags=O.LK_USE_INITIAL_FLOW | O.LK_FORWARD_BACKWARD, fb_thresh=1.0)
    assert st_fb.sum() <= st_plain.sum() and np.all(st_fb <= st_plain)
    inside = (pts[:, 0] > 160) & (pts[:, 0] < 190) & (pts[:, 1] > 110) & (pts[:, 1] < 130)
    assert inside.any() and st_fb[inside].mean() < 0.5
    good = st_fb == 1
    e = np.abs(out[good] - pts[good] - np.array([2, 1], dtype=np.float32)).max(axis=1)
    assert np.median(e) < 0.01 and (e < 0.3).mean() > 0.97   # survivors sit on the true shift (a few straddle the patch edge)


def _corner_template(bright=True, arc=9, centre=100, delta=60):
    """7x7 patch: `arc` contiguous circle pixels differ from the centre by delta, the rest equal the centre."""
    circle = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
              (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
    p = np.full((7, 7), centre, dtype=np.uint8)
    for k in range(arc):
        dx, dy = circle[k]
        p[3 + dy, 3 + dx] = centre + delta if bright else centre - delta
    return p


def test_fast_templates_and_score():
    for bright in (True, False):
        img = np.full((21, 21), 100, dtype=np.uint8)
        img[7:14, 7:14] = _corner_template(bright, arc=9, delta=60)
        kps = O.fast(img, threshold=20)
        assert kps.shape[0] == 1 and kps[0, 0] == 10 and kps[0, 1] == 10
        assert kps[0, 2] == 59                       # largest threshold that still passes: delta - 1
        assert O.fast(img, threshold=60).shape[0] == 0
    img = np.full((21, 21), 100, dtype=np.uint8)
    img[7:14, 7:14] = _corner_template(True, arc=8)
    assert O.fast(img, threshold=20).shape[0] == 0   # 8 contiguous is not a corner


def test_fast_nms_strict_ties_and_borders():
    img = np.full((30, 40), 100, dtype=np.uint8)
    img[5:12, 5:12] = _corner_template(True, 9, delta=60)
    sm = O.fast_score_map(img, 20)
    assert sm[8, 8] == 59
    # two adjacent equal-score corners kill each other under strict NMS
    sm2 = np.zeros_like(sm)
    img2 = synth.texture(160, 120, seed=13)
    kps = O.fast(img2, 20)
    sm2 = O.fast_score_map(img2, 20)
    for x, y, s in kps:
        x, y = int(x), int(y)
        nb = sm2[y - 1:y + 2, x - 1:x + 2].astype(int).copy()
        nb[1, 1] = -1
        assert sm2[y, x] == s and s > nb.max()
    # every strict local maximum of the score map is reported, in row-major order
    loc = [(x, y) for y in range(1, 119) for x in range(1, 159)
           if sm2[y, x] > 0 and sm2[y, x] > np.delete(sm2[y - 1:y + 2, x - 1:x + 2].ravel(), 4).max()]
    assert [(int(k[0]), int(k[1])) for k in kps] == loc
    assert sm2[:3].max() == 0 and sm2[-3:].max() == 0 and sm2[:, :3].max() == 0 and sm2[:, -3:].max() == 0


def test_grid_geometry_and_selection():
    img = synth.texture(752, 480, seed=14)
    kps = O.grid_fast(img, 150, 5, 5, 20)
    assert kps.shape[0] <= 25 * 7
    assert kps[:, 0].max() < 750                       # columns 750-751 are never searched (752 % 5 = 2)
    cx, cy = (kps[:, 0] // 150).astype(int), (kps[:, 1] // 96).astype(int)
    cell = cy * 5 + cx
    assert np.all(np.diff(cell) >= 0)                  # concatenated in cell order
    lx, ly = kps[:, 0] - cx * 150, kps[:, 1] - cy * 96
    assert lx.min() >= 3 and lx.max() <= 146 and ly.min() >= 3 and ly.max() <= 92   # 3-px dead band in every cell
    for c in np.unique(cell):
        m = cell == c
        assert m.sum() <= 7
        r = kps[m]
        key = list(zip(-r[:, 2], r[:, 1], r[:, 0]))
        assert key == sorted(key)                      # (response desc, y asc, x asc)
        # per-cell result equals FAST on the cell view followed by the same ordering
        x0, y0 = (c % 5) * 150, (c // 5) * 96
        ref = O.fast(img[y0:y0 + 96, x0:x0 + 150], 20)
        ref = sorted(ref.tolist(), key=lambda k: (-k[2], k[1], k[0]))[:7]
        assert [[k[0] + x0, k[1] + y0, k[2]] for k in ref] == r.tolist()
    # degenerate geometry: 59 px wide, 10 columns -> cells of 5 px, 11 of them, all processed (none can hold a corner)
    assert O.grid_fast(synth.texture(59, 40, seed=15), 50, 10, 2, 20).shape[0] == 0
    # std::sort variant agrees wherever scores are distinct within a cell
    k1 = O.grid_fast(img, 150, 5, 5, 20, tie_mode=1)
    assert k1.shape == kps.shape
    assert sorted(map(tuple, k1[:, 2:3].tolist())) == sorted(map(tuple, kps[:, 2:3].tolist()))


def test_gyro_seed_pure_rotation():
    # camera rotating about its optical axis: the predicted flow equals the in-plane rotation about (cx, cy)
    K4 = np.array([458.0, 457.0, 367.0, 248.0], dtype=np.float32)
    rate = 0.2                                           # rad/s about camera z
    t = np.arange(0, 1.0, 0.005) + 0.001                 # off the window edges, so no rounding ambiguity
    gyro = np.tile([0.0, 0.0, rate], (len(t), 1))
    R, used = O.integrate_gyro(t, gyro, 0.50, 0.55, np.eye(3))
    assert used == 13                                    # samples in [0.49, 0.555): 0.491 .. 0.551
    ang = rate * 0.05
    assert np.allclose(R, [[np.cos(ang), np.sin(ang), 0], [-np.sin(ang), np.cos(ang), 0], [0, 0, 1]], atol=1e-6)
    pts = synth.random_points(752, 480, 50, seed=16)
    out = O.predict_flow(K4, R, pts)
    xn, yn = (pts[:, 0] - 367.0) / 458.0, (pts[:, 1] - 248.0) / 457.0
    ex = 458.0 * (np.cos(ang) * xn + np.sin(ang) * yn) + 367.0
    ey = 457.0 * (-np.sin(ang) * xn + np.cos(ang) * yn) + 248.0
    assert np.abs(out[:, 0] - ex).max() < 2e-3 and np.abs(out[:, 1] - ey).max() < 2e-3
    # identity rotation leaves points where they are
    assert np.abs(O.predict_flow(K4, np.eye(3), pts) - pts).max() < 1e-3


# ---- outlier rejection (fundamental-matrix RANSAC, the build's own seeded definition) ----

@pytest.mark.parametrize("calib8,t", [((460.0, 458.0, 376.0, 240.0, 0.0, 0.0, 0.0, 0.0), (0.3, 0.02, 0.05)),
                                      ((460.0, 458.0, 376.0, 240.0, -0.28, 0.07, 2e-4, 2e-5), (0.3, 0.02, 0.05)),
                                      ((460.0, 458.0, 376.0, 240.0, 0.0, 0.0, 0.0, 0.0), (0.25, 0.0, 0.0))])   # pure sideways: F33 = 0
def test_reject_outliers_separates_random_matches(calib8, t):
    import synth
    prev, nxt, is_out = synth.two_view_matches(150, 30, calib8, seed=3, t=t)
    st, h, F = O.reject_outliers(prev, nxt, np.ones(150, np.uint8), calib8, seed=7)
    assert h >= 0
    assert st[~is_out].mean() >= 0.95          # nearly every true match survives
    assert st[is_out].mean() <= 0.15           # nearly every random match is rejected
    # the winning model is a fundamental matrix of the normalized points: x2^T F x1 ~ 0 on the true matches
    def norm(p):
        return np.stack([(p[:, 0] - calib8[2]) / calib8[0], (p[:, 1] - calib8[3]) / calib8[1], np.ones(len(p))], axis=1)
    if not any(calib8[4:]):
        r = np.abs(np.einsum("ij,jk,ik->i", norm(nxt[~is_out]), F, norm(prev[~is_out]))) / np.linalg.norm(F)
        assert np.median(r) < 2e-3


@pytest.mark.parametrize("n,share", sorted(__import__("synth").REJECTION_BARS))
def test_reject_outliers_precision_and_recall_against_ground_truth(n, share):
    """The claim that the build's own RANSAC does what the reference's findFundamentalMat step is for, as numbers: with
    the reference's threshold (1 / max_focal) and OpenCV's iteration limit, precision and recall of the final mask
    against the known random matches, for a dozen, 150 and 2000 points."""
    import synth
    p_bar, r_bar = synth.REJECTION_BARS[(n, share)]
    p, r = synth.rejection_scores(lambda a, b, c, seed: O.reject_outliers(a, b, np.ones(len(a), np.uint8), c, hypotheses=0, seed=seed)[0], n, share)
    assert p >= p_bar and r >= r_bar, (p, r)


def test_reject_outliers_is_deterministic_and_keeps_lost_points_lost():
    import synth
    calib8 = (460.0, 458.0, 376.0, 240.0, -0.1, 0.01, 0.0, 0.0)
    prev, nxt, is_out = synth.two_view_matches(80, 10, calib8, seed=5)
    status = np.ones(80, np.uint8)
    status[::7] = 0                                               # points LK had already lost
    a = O.reject_outliers(prev, nxt, status, calib8, seed=1)
    b = O.reject_outliers(prev, nxt, status, calib8, seed=1)
    c = O.reject_outliers(prev, nxt, status, calib8, seed=2)
    assert np.array_equal(a[0], b[0]) and a[1] == b[1] and np.array_equal(a[2], b[2])
    assert not a[0][::7].any()                                    # status is only ever cleared
    assert a[1] != c[1] or np.array_equal(a[0], c[0])             # another seed draws other samples
    few = O.reject_outliers(prev[:7], nxt[:7], np.ones(7, np.uint8), calib8)
    assert few[1] == -1 and not few[0].any()                      # fewer than 8 pairs: nothing survives
    flat = O.reject_outliers(prev, prev * 0 + 5.0, np.ones(80, np.uint8), calib8)
    assert flat[1] == -1 and not flat[0].any()                    # degenerate data: no model


# ---- stereo matching geometry (seed from the extrinsic rotation, known-E test) ----

CAM0 = (458.654, 457.296, 367.215, 248.375, -0.28340811, 0.07395907, 0.00019359, 1.76187114e-05)
CAM1 = (457.587, 456.134, 379.999, 255.238, -0.28368365, 0.07451284, -0.00010473, -3.55590700e-05)


def test_stereo_seed_is_the_projection_of_the_point_at_infinity():
    import synth
    # very distant points: the translation does not matter, the seed must land on the true cam1 pixel
    a, b, Rt, _ = synth.stereo_views(200, CAM0, CAM1, seed=1, depth=(4000.0, 9000.0))
    seed = O.stereo_seed(a, CAM0, CAM1, Rt)
    assert np.abs(seed - b).max() < 0.05
    # identity rotation, one undistorted camera: the seed is the point itself
    pin = (460.0, 458.0, 376.0, 240.0, 0, 0, 0, 0)
    eye = np.concatenate([np.eye(3).reshape(9), [-0.1, 0, 0]])
    same = O.stereo_seed(a, pin, pin, eye)
    assert np.abs(same - a).max() < 1e-3
    # ... up to the float rounding of the normalized point between undistortPoints and projectPoints (cv::Point2f)
    xn = ((a.astype(np.float64) - [376.0, 240.0]) * [1 / 460.0, 1 / 458.0]).astype(np.float32).astype(np.float64)
    assert np.array_equal(same, (xn * [460.0, 458.0] + [376.0, 240.0]).astype(np.float32))
    # a near point is displaced from its seed along the baseline by about fx * b / z
    a, b, Rt, z = synth.stereo_views(200, pin, pin, rot=(0, 0, 0), t=(-0.11, 0, 0), seed=2)
    seed = O.stereo_seed(a, pin, pin, Rt)
    assert np.abs((b - seed)[:, 0] + 460.0 * 0.11 / z).max() < 0.01 and np.abs((b - seed)[:, 1]).max() < 0.01


def test_stereo_check_keeps_true_matches_and_drops_off_line_and_outside_points():
    import synth
    a, b, Rt, _ = synth.stereo_views(300, CAM0, CAM1, seed=3)
    inside = (b[:, 0] >= 0) & (b[:, 0] <= 751) & (b[:, 1] >= 0) & (b[:, 1] <= 479)
    st = O.stereo_check(a, b, np.ones(300, np.uint8), CAM0, CAM1, Rt, 752, 480, 3.0)
    assert np.array_equal(st == 1, inside)                         # exact geometry: only the frame test removes points
    # 8 px off the (nearly horizontal) epipolar line: dropped; 1 px off: kept
    off = b + np.array([0.0, 8.0], np.float32)
    near = b + np.array([0.0, 1.0], np.float32)
    ok = inside & (off[:, 1] <= 479) & (near[:, 1] <= 479)
    assert not O.stereo_check(a, off, np.ones(300, np.uint8), CAM0, CAM1, Rt, 752, 480, 3.0)[ok].any()
    assert O.stereo_check(a, near, np.ones(300, np.uint8), CAM0, CAM1, Rt, 752, 480, 3.0)[ok].all()
    # sliding along the line never matters; lost points stay lost; threshold <= 0 leaves only the frame test
    along = b + np.array([6.0, 0.0], np.float32)
    ok2 = inside & (along[:, 0] <= 751)
    assert O.stereo_check(a, along, np.ones(300, np.uint8), CAM0, CAM1, Rt, 752, 480, 3.0)[ok2].mean() > 0.97
    lost = np.ones(300, np.uint8)
    lost[::5] = 0
    assert not O.stereo_check(a, b, lost, CAM0, CAM1, Rt, 752, 480, 3.0)[::5].any()
    assert np.array_equal(O.stereo_check(a, off, np.ones(300, np.uint8), CAM0, CAM1, Rt, 752, 480, 0.0) == 1,
                          (off[:, 0] >= 0) & (off[:, 0] <= 751) & (off[:, 1] >= 0) & (off[:, 1] <= 479))


# ---- descriptor matching (Hamming 2-NN both ways, ratio test, symmetry test) ----

def test_match_descriptors_finds_planted_pairs_and_nothing_else():
    import synth
    d0, d1, pairs = synth.descriptor_sets(120, 40, 60, flip_bits=12, seed=1)
    m = O.match_descriptors(d0, d1, knn_ratio=0.7)
    assert np.array_equal(m[:, :2], pairs)                       # every planted pair, in query order, and no random pair
    want = np.array([np.unpackbits(d0[i] ^ d1[j]).sum() for i, j in pairs])
    assert np.array_equal(m[:, 2], want) and want.max() <= 12


def test_match_descriptors_ratio_symmetry_and_degenerate_sets():
    rng = np.random.default_rng(2)
    base = rng.integers(0, 256, (6, 32), dtype=np.uint8)
    # query 0 has two equally good candidates (its copy twice): ratio 0/0 is not "> ratio", the lower index wins,
    # and the symmetry test keeps it only if that candidate points back
    d1 = np.concatenate([base[:1], base[:1], base[1:]])
    m = O.match_descriptors(base, d1, knn_ratio=0.7)
    assert [tuple(r) for r in m[:, :2]] == [(0, 0), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6)]
    # an ambiguous query (two train descriptors at almost the same distance) fails the ratio test
    amb = base[2].copy(); amb[0] ^= 1
    amb2 = base[2].copy(); amb2[1] ^= 1
    q = base[2].copy(); q[2] ^= 1
    assert len(O.match_descriptors(np.stack([q, base[5]]), np.stack([amb, amb2]), knn_ratio=0.7)) == 0
    assert [tuple(r) for r in O.match_descriptors(np.stack([q, base[5]]), np.stack([amb, base[4]]), knn_ratio=0.7)] == [(0, 0, 2)]
    # a train set with fewer than two descriptors has no second neighbour: nothing survives; empty sets
    assert len(O.match_descriptors(base, base[:1], knn_ratio=0.7)) == 0
    assert len(O.match_descriptors(base[:0], base, knn_ratio=0.7)) == 0 and len(O.match_descriptors(base, base[:0], knn_ratio=0.7)) == 0
    # symmetry: two queries competing for one train descriptor, only its own nearest is accepted
    near = base[3].copy(); near[5] ^= 0x0f
    m = O.match_descriptors(np.stack([near, base[3]]), np.stack([base[3], base[0]]), knn_ratio=0.7)
    assert [tuple(r) for r in m] == [(1, 0, 0)]


def test_adaptive_hypothesis_count_shortcut_equals_the_walk(tmp_path):
    """The device decides the adaptive hypothesis count of the outlier rejection by repeated squaring wherever that cannot
    differ from the definition (the smallest N whose running product q * q * ... * q, N factors in that order, is <= 0.001)
    and walks the products only in a narrow band.  The function is host-compilable text: it is cut out of the kernel
    header, built with g++ (the fast logarithm replaced by logf: it only places the first guess) and compared with the
    plain walk for every (inliers, points) pair up to 400 points, a sample of larger sets, and several limits."""
    import ctypes, os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "feature_tracker_amd", "csrc", "ft_kernels_outlier.hpp")).read()
    a = src.index("__device__ inline double ro_pow_int")
    b = src.index("// T threads per workgroup")
    body = src[a:b].replace("__device__ inline", "static inline").replace("__logf", "logf")
    cpp = tmp_path / "count.cpp"
    cpp.write_text("#include <cmath>\n" + body + "\nextern \"C\" int count(int inliers, int n, int max_h) { return ro_hypotheses_needed(inliers, n, max_h); }\n")
    so = tmp_path / "count.so"
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", str(so), str(cpp)])
    fn = ctypes.CDLL(str(so)).count
    fn.argtypes = [ctypes.c_int] * 3
    fn.restype = ctypes.c_int

    def walk(inliers, n, max_h):
        w = inliers / n
        w2 = w * w; w4 = w2 * w2; w8 = w4 * w4
        q = 1.0 - w8
        p, N = 1.0, 0
        while N < max_h:
            p = p * q
            N += 1
            if p <= 0.001:
                break
        return N

    rng = np.random.default_rng(5)
    cases = [(i, n, m) for m in (1, 2, 32, 256, 1000, 1024) for n in range(8, 401, 7) for i in range(1, n + 1)]
    for _ in range(20000):
        n = int(rng.integers(400, 20000)); cases.append((int(rng.integers(1, n + 1)), n, int(rng.choice([256, 1000, 1024]))))
    for i, n, m in cases:
        assert fn(i, n, m) == walk(i, n, m), (i, n, m)
