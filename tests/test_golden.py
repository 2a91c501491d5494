"""Committed golden vectors (tests/golden/frontend_golden.npz, made by tests/golden/make_golden.py from the oracle).

CPU part: the oracle still reproduces them, so an edit to the oracle cannot silently move the target.  GPU part: the
device path matches them without consulting the oracle at all."""
import os

import numpy as np
import pytest

import oracle_lib as O

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frontend_golden.npz"))
WIN, MAX_LEVEL, LEVELS = int(G["win"]), int(G["max_level"]), int(G["levels"])
FAST_KEY = "fast_grid_4x3_need120_thr12"
R = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "outlier_golden.npz"))


def test_oracle_reproduces_golden_vectors():
    pp, pn = O.Pyramid(G["prev"], WIN, MAX_LEVEL), O.Pyramid(G["next"], WIN, MAX_LEVEL)
    assert pp.levels == LEVELS
    for l in range(LEVELS):
        assert np.array_equal(pp.level(l), G[f"prev_level{l}"])
        assert np.array_equal(O.scharr(pp.level(l)), G[f"prev_grad{l}"])
    assert np.array_equal(O.grid_fast(G["prev"], 120, 4, 3, 12), G[FAST_KEY])
    xy, st, er, it = O.lk_track(pp, pn, G["lk_prev_xy"], win=WIN, max_level=MAX_LEVEL, want_iters=True)
    assert np.array_equal(st, G["lk_status"]) and np.array_equal(it, G["lk_iters"])
    ok = st == 1
    assert np.abs(xy[ok] - G["lk_next_xy"][ok]).max() <= 1e-4 and np.abs(er[ok] - G["lk_err"][ok]).max() <= 1e-4
    _, st_fb, _ = O.lk_track(pp, pn, G["lk_prev_xy"], win=WIN, max_level=MAX_LEVEL, flags=O.LK_USE_INITIAL_FLOW | O.LK_FORWARD_BACKWARD,
                             fb_thresh=0.5)
    assert np.array_equal(st_fb, G["lk_fb_status"])
    R, used = O.integrate_gyro(G["gyro_t"], G["gyro_w"], 0.05, 0.1, np.eye(3))
    assert used == int(G["gyro_used"]) and np.allclose(R, G["gyro_R"], atol=1e-7)
    assert np.allclose(O.homography(G["gyro_K4"], R), G["gyro_H"], atol=1e-5)
    assert np.allclose(O.predict_flow(G["gyro_K4"], R, G["lk_prev_xy"]), G["gyro_pred"], atol=1e-3)


def test_oracle_reproduces_outlier_golden_vectors():
    st, h, F = O.reject_outliers(R["prev_xy"], R["next_xy"], R["status_in"], R["calib_prev"], R["calib_next"],
                                 hypotheses=int(R["hypotheses"]), seed=int(R["seed"]))
    assert np.array_equal(st, R["status_out"]) and h == int(R["winner"]) and np.array_equal(F, R["F"])


def test_outlier_golden_vectors_are_sane():
    """Independent of any implementation: most true matches kept, nearly all random ones dropped, lost points stay lost."""
    keep, bad, was = R["status_out"] == 1, R["is_outlier"], R["status_in"] == 1
    assert not keep[~was].any()
    assert keep[~bad & was].mean() >= 0.9 and keep[bad].mean() <= 0.15


def test_golden_vectors_are_sane():
    """Independent of any implementation: the tracked displacement is small and consistent (one camera motion)."""
    ok = G["lk_status"] == 1
    assert ok.sum() >= 0.8 * len(ok) - 4
    flow = (G["lk_next_xy"] - G["lk_prev_xy"])[ok]
    assert np.abs(flow).max() < 8 and flow.std(axis=0).max() < 1.5
    assert not G["lk_status"][-2:].any()          # the two points far outside the image are lost


@pytest.mark.gpu
def test_device_matches_golden_vectors():
    from feature_tracker_amd import device as D
    h, w = G["prev"].shape
    with D.DeviceContext(w, h, max_levels=4, n_streams=1, n_slots=2, max_points=4096, max_grid_cells=16, max_per_cell=16) as ctx:
        ctx.upload(0, 0, G["prev"])
        ctx.upload(0, 1, G["next"])
        ach = ctx.build_pyramid([0, 0], [0, 1], MAX_LEVEL, WIN, with_gradients=True)
        assert ach[0] == LEVELS - 1
        for l in range(LEVELS):
            assert np.array_equal(ctx.download_level(0, 0, l), G[f"prev_level{l}"])
            assert np.array_equal(ctx.download_grad(0, 0, l), G[f"prev_grad{l}"])
        kp, n, _ = ctx.fast_grid([0], [0], 12, 4, 3, 120 // 12 + 1)
        assert np.array_equal(kp[0, :n[0]], G[FAST_KEY])
        job = dict(prev=(0, 0), next=(0, 1), prev_xy=G["lk_prev_xy"])
        xy, st, er, it = ctx.lk_track([job], D.LKParams(win=WIN, max_level=MAX_LEVEL), want_iters=True)[0]
        assert (st == G["lk_status"]).mean() >= 0.999
        ok = (st == 1) & (G["lk_status"] == 1)
        assert np.abs(xy[ok] - G["lk_next_xy"][ok]).max() <= 0.01       # px, the stated tolerance
        assert np.abs(er[ok] - G["lk_err"][ok]).max() <= 1e-3
        assert np.array_equal(it, G["lk_iters"])
        st_fb = ctx.lk_track([job], D.LKParams(win=WIN, max_level=MAX_LEVEL, flags=D.FT_LK_USE_INITIAL_FLOW | D.FT_LK_FORWARD_BACKWARD,
                                                fb_thresh=0.5))[0][1]
        assert np.array_equal(st_fb, G["lk_fb_status"])
        H = ctx.homography(G["gyro_K4"], G["gyro_R"])
        assert np.allclose(H, G["gyro_H"], atol=1e-5)
        assert np.abs(ctx.predict_flow(H, G["lk_prev_xy"]) - G["gyro_pred"]).max() <= 1e-3


@pytest.mark.gpu
def test_device_matches_outlier_golden_vectors():
    from feature_tracker_amd import device as D
    with D.DeviceContext(64, 64, max_levels=1, n_streams=1, n_slots=2, max_points=512, max_grid_cells=4, max_per_cell=4) as ctx:
        st = ctx.reject_outliers([dict(prev_xy=R["prev_xy"], next_xy=R["next_xy"], status=R["status_in"], calib_prev=R["calib_prev"],
                                       calib_next=R["calib_next"])], hypotheses=int(R["hypotheses"]), seed=int(R["seed"]))[0]
    assert np.array_equal(st, R["status_out"])
