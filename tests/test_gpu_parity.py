"""GPU parity: the gfx950 kernels, called through the C ABI (include/ft_device.h), against the CPU oracle.

Bars: pyramids, gradients and FAST/grid selections bit-exact; LK positions within 0.01 px and status flags
agreeing on >= 99.9 % of features (the kernels sum exactly like the oracle, so in practice they are identical).
"""
import numpy as np
import pytest

import oracle_lib as O
import synth

pytestmark = pytest.mark.gpu

POS_TOL = 0.01        # px, tolerance stated by the north star
STATUS_AGREE = 0.999


@pytest.fixture(scope="module")
def dev():
    from feature_tracker_amd import device as D
    return D


@pytest.fixture(params=["one_wave_per_point", "two_waves_per_point"])
def lk_form(request, monkeypatch):
    """The tracking kernel has two forms (one wavefront per point; a tracker / templater pair per point, used for
    launches of up to 2048 points): contexts created inside the test run every launch with the named one."""
    monkeypatch.setenv("FT_LK_PAIR_MAX", "0" if request.param == "one_wave_per_point" else "1000000000")
    return request.param


def _ctx(D, w, h, **kw):
    args = dict(max_levels=6, n_streams=2, n_slots=2, max_points=20000, max_grid_cells=400, max_per_cell=32)
    args.update(kw)
    return D.DeviceContext(w, h, **args)


@pytest.mark.parametrize("shape,win,max_level", [((480, 752), 21, 3), ((480, 752), 15, 3), ((720, 1280), 21, 4),
                                                  ((479, 751), 15, 3), ((33, 47), 5, 2), ((100, 80), 21, 5),
                                                  ((257, 129), 9, 4), ((8, 300), 3, 1)])
def test_pyramid_and_gradients_bit_exact(dev, shape, win, max_level):
    h, w = shape
    img = synth.texture(w, h, seed=21) if min(shape) > 16 else np.random.default_rng(3).integers(0, 256, shape, dtype=np.uint8)
    ref = O.Pyramid(img, win, max_level)
    with _ctx(dev, w, h) as ctx:
        ctx.upload(0, 0, img)
        ach = ctx.build_pyramid([0], [0], max_level, win, with_gradients=True)
        assert ach[0] == ref.levels - 1
        for l in range(ref.levels):
            assert np.array_equal(ctx.download_level(0, 0, l), ref.level(l)), f"level {l}"
            assert np.array_equal(ctx.download_grad(0, 0, l), O.scharr(ref.level(l))), f"gradient {l}"


def test_pyramid_random_noise_and_extremes(dev):
    rng = np.random.default_rng(5)
    for img in (rng.integers(0, 256, (240, 376), dtype=np.uint8), np.zeros((64, 200), np.uint8), np.full((64, 200), 255, np.uint8)):
        h, w = img.shape
        ref = O.Pyramid(img, 15, 2)
        with _ctx(dev, w, h) as ctx:
            ctx.upload(0, 0, img)
            ctx.build_pyramid([0], [0], 2, 15, with_gradients=False)
            ctx.build_gradients([0], [0])
            for l in range(ref.levels):
                assert np.array_equal(ctx.download_level(0, 0, l), ref.level(l))
                assert np.array_equal(ctx.download_grad(0, 0, l), O.scharr(ref.level(l)))


def test_pyramid_batched_streams(dev):
    imgs = [synth.texture(376, 240, seed=30 + i) for i in range(4)]
    with _ctx(dev, 376, 240, n_streams=4) as ctx:
        for i, im in enumerate(imgs):
            ctx.upload(i, 1, im)
        ctx.build_pyramid([0, 1, 2, 3], [1, 1, 1, 1], 3, 15)
        for i, im in enumerate(imgs):
            ref = O.Pyramid(im, 15, 3)
            for l in range(ref.levels):
                assert np.array_equal(ctx.download_level(i, 1, l), ref.level(l))
                assert np.array_equal(ctx.download_grad(i, 1, l), O.scharr(ref.level(l)))


@pytest.mark.parametrize("shape,win,max_level,pad", [((480, 752), 21, 3, 0), ((479, 751), 15, 3, 5), ((33, 47), 5, 2, 1), ((100, 80), 21, 5, 0),
                                                      ((257, 129), 9, 4, 3), ((8, 300), 3, 1, 7), ((720, 1280), 21, 4, 0), ((40, 7), 3, 2, 2)])
def test_ingest_pyramid_single_pass_bit_exact(dev, shape, win, max_level, pad):
    """Frames already in HBM (any row stride and alignment): level 0 and the pyramid come out of one pass."""
    import torch
    h, w = shape
    rng = np.random.default_rng(11)
    imgs = [synth.texture(w, h, seed=61) if min(shape) > 16 else rng.integers(0, 256, shape, dtype=np.uint8),
            rng.integers(0, 256, shape, dtype=np.uint8)]
    # rows with a stride that is not a multiple of anything convenient, starting at an odd byte offset
    stride = w + pad
    flat = torch.zeros(2 * h * stride + 64, dtype=torch.uint8, device="cuda")
    ptrs = []
    for i, im in enumerate(imgs):
        padded = np.zeros((h, stride), np.uint8)
        padded[:, :w] = im
        off = 1 + i * h * stride if pad else i * h * stride
        flat[off:off + h * stride] = torch.from_numpy(padded.reshape(-1)).cuda()
        ptrs.append(flat.data_ptr() + off)
    torch.cuda.synchronize()
    with _ctx(dev, w, h) as ctx:
        ach = ctx.set_frames_device_pyramid([0, 1], [1, 0], ptrs, w, h, stride, max_level, win)
        ctx.sync()
        for i, im in enumerate(imgs):
            ref = O.Pyramid(im, win, max_level)
            assert ach[i] == ref.levels - 1
            for l in range(ref.levels):
                assert np.array_equal(ctx.download_level(i, 1 - i, l), ref.level(l)), f"image {i} level {l}"


@pytest.mark.parametrize("w,h,n,stage_bar", [(751, 479, 3, "1"), (751, 479, 3, "0"), (752, 480, 8, "1"), (752, 480, 2, "1"), (752, 480, 2, "0")])
def test_staged_host_frames_pyramid_bit_exact(dev, w, h, n, stage_bar, monkeypatch):
    """Host frames through the staging buffers.  Pinned staging: a few frames (or a width that is no multiple of 16) are read
    in place by the fused level-0 pass, many frames cross the bus in a linear pass of their own first.  Contexts of up to
    four streams on a device with a large BAR stage in device memory instead (FT_STAGE_BAR=0 keeps them on pinned
    staging).  Later rounds reuse the staging buffers (the staging call waits for the pass that still reads them)."""
    monkeypatch.setenv("FT_STAGE_BAR", stage_bar)
    with _ctx(dev, 752, 480, n_streams=n) as ctx:
        for rnd in range(3):
            frames = [synth.texture(w, h, seed=200 + n * rnd + i) for i in range(n)]
            # round 2 names the streams out of order: their staging buffers are then not at a constant distance, and the
            # linear pass reads them over the bus itself instead of from the landing area of one copy command
            order = list(range(n)) if rnd != 2 else [1, 0] + list(range(2, n))
            ach = ctx.set_frames_host_pyramid(order, [rnd % 2] * n, [frames[i] for i in order], max_level=3, win=21)
            ach = [ach[order.index(i)] for i in range(n)]
            for i, f in enumerate(frames):
                ref = O.Pyramid(f, 21, 3)
                assert ach[i] == ref.levels - 1
                for l in range(ref.levels):
                    assert np.array_equal(ctx.download_level(i, rnd % 2, l), ref.level(l)), f"round {rnd} image {i} level {l}"


def _compare_lk(gpu, ref, min_agree=STATUS_AGREE):
    g_xy, g_st, g_err = gpu[:3]
    r_xy, r_st, r_err = ref[:3]
    agree = (g_st == r_st).mean()
    assert agree >= min_agree, f"status agreement {agree}"
    both = (g_st == 1) & (r_st == 1)
    if both.any():
        d = np.abs(g_xy[both] - r_xy[both]).max()
        assert d <= POS_TOL, f"max position difference {d}"
        assert np.abs(g_err[both] - r_err[both]).max() <= 1e-3
    return agree


@pytest.mark.parametrize("win,max_level,shape,n", [(21, 3, (480, 752), 150), (15, 3, (480, 752), 150), (21, 4, (720, 1280), 2000),
                                                    (9, 2, (240, 320), 300), (31, 2, (480, 752), 200)])
def test_lk_matches_oracle(dev, win, max_level, shape, n, lk_form, monkeypatch):
    """Every window runs the general kernels; 21 x 21 and 15 x 15 also have kernels compiled for that size, which are the
    default for them (FT_LK_FIXED_WINDOWS=0, read when the context is created, keeps them on the general ones)."""
    h, w = shape
    prev, nxt = synth.shifted_pair(w, h, dx=4, dy=-3, seed=40 + win)
    nxt = np.clip(nxt.astype(np.int16) + np.random.default_rng(1).integers(-3, 4, nxt.shape), 0, 255).astype(np.uint8)
    pts = synth.random_points(w, h, n, seed=41, border=-20)   # some points outside the frame on purpose
    rp, rn = O.Pyramid(prev, win, max_level), O.Pyramid(nxt, win, max_level)
    ref = O.lk_track(rp, rn, pts, win=win, max_level=max_level, want_iters=True)
    with _ctx(dev, w, h) as ctx:
        ctx.upload(0, 0, prev)
        ctx.upload(0, 1, nxt)
        ctx.build_pyramid([0, 0], [0, 1], max_level, win, with_gradients=False)   # LK evaluates Scharr itself
        out = ctx.lk_track([dict(prev=(0, 0), next=(0, 1), prev_xy=pts)], dev.LKParams(win=win, max_level=max_level), want_iters=True)[0]
    _compare_lk(out, ref)
    assert np.array_equal(out[3], ref[3])          # identical iteration counts per level
    assert np.array_equal(out[0], ref[0])          # exact sums on both sides: positions are bit-identical
    if win in (21, 15):
        monkeypatch.setenv("FT_LK_FIXED_WINDOWS", "0")
        with _ctx(dev, w, h) as ctx:
            ctx.upload(0, 0, prev)
            ctx.upload(0, 1, nxt)
            ctx.build_pyramid([0, 0], [0, 1], max_level, win, with_gradients=False)
            gen = ctx.lk_track([dict(prev=(0, 0), next=(0, 1), prev_xy=pts)], dev.LKParams(win=win, max_level=max_level), want_iters=True)[0]
        monkeypatch.delenv("FT_LK_FIXED_WINDOWS")
        for a, b in zip(out, gen):
            assert np.array_equal(a, b)
    if win == 21:
        # a context of one stream takes the 21 x 21 kernels that hold their pixel pairs across iterations (FT_LK_HOLD_SINGLE=0: not)
        with _ctx(dev, w, h, n_streams=1) as ctx:
            ctx.upload(0, 0, prev)
            ctx.upload(0, 1, nxt)
            ctx.build_pyramid([0, 0], [0, 1], max_level, win, with_gradients=False)
            one = ctx.lk_track([dict(prev=(0, 0), next=(0, 1), prev_xy=pts)], dev.LKParams(win=win, max_level=max_level), want_iters=True)[0]
            fb = ctx.lk_track([dict(prev=(0, 0), next=(0, 1), prev_xy=pts)], dev.LKParams(win=win, max_level=max_level, flags=dev.FT_LK_USE_INITIAL_FLOW | dev.FT_LK_FORWARD_BACKWARD, fb_thresh=1.0))[0]
        for a, b in zip(out, one):
            assert np.array_equal(a, b)
        with _ctx(dev, w, h) as ctx:
            ctx.upload(0, 0, prev)
            ctx.upload(0, 1, nxt)
            ctx.build_pyramid([0, 0], [0, 1], max_level, win, with_gradients=False)
            fb2 = ctx.lk_track([dict(prev=(0, 0), next=(0, 1), prev_xy=pts)], dev.LKParams(win=win, max_level=max_level, flags=dev.FT_LK_USE_INITIAL_FLOW | dev.FT_LK_FORWARD_BACKWARD, fb_thresh=1.0))[0]
        for a, b in zip(fb, fb2):
            assert np.array_equal(a, b)


def test_lk_initial_guess_no_initial_flow_and_min_eig(dev, lk_form):
    w, h = 376, 240
    prev, nxt = synth.shifted_pair(w, h, dx=6, dy=5, seed=50)
    pts = synth.random_points(w, h, 400, seed=51, border=5)
    guess = pts + np.array([5.0, 4.0], dtype=np.float32)
    rp, rn = O.Pyramid(prev, 15, 3), O.Pyramid(nxt, 15, 3)
    with _ctx(dev, w, h) as ctx:
        ctx.upload(0, 0, prev)
        ctx.upload(0, 1, nxt)
        ctx.build_pyramid([0, 0], [0, 1], 3, 15)
        job = dict(prev=(0, 0), next=(0, 1), prev_xy=pts, guess=guess)
        for flags in (dev.FT_LK_USE_INITIAL_FLOW, 0, dev.FT_LK_USE_INITIAL_FLOW | dev.FT_LK_GET_MIN_EIGENVALS):
            ref = O.lk_track(rp, rn, pts, guess=guess, win=15, max_level=3, flags=flags)
            out = ctx.lk_track([job], dev.LKParams(win=15, max_level=3, flags=flags))[0]
            _compare_lk(out, ref)
            assert np.array_equal(out[2], ref[2])   # err (residual or min eigenvalue) identical
        # without the residual: same positions and status flags as with it, err left at 0
        with_err = ctx.lk_track([job], dev.LKParams(win=15, max_level=3))[0]
        flags = dev.FT_LK_USE_INITIAL_FLOW | dev.FT_LK_NO_ERROR
        ref = O.lk_track(rp, rn, pts, guess=guess, win=15, max_level=3, flags=flags)
        out = ctx.lk_track([job], dev.LKParams(win=15, max_level=3, flags=flags))[0]
        _compare_lk(out, ref)
        assert np.array_equal(out[0], with_err[0]) and np.array_equal(out[1], with_err[1])
        assert not out[2].any() and not ref[2].any() and with_err[2].any()
        # tight termination settings
        ref = O.lk_track(rp, rn, pts, guess=guess, win=15, max_level=3, max_iter=3, eps=0.1, min_eig=5e-3)
        out = ctx.lk_track([job], dev.LKParams(win=15, max_level=3, max_iter=3, eps=0.1, min_eig=5e-3))[0]
        _compare_lk(out, ref)
        assert (ref[1] == 0).any()


def test_lk_forward_backward_and_rect_window(dev, lk_form, monkeypatch):
    """Forward-backward tracking equals the checker's, in one launch (the default: a feature's wavefronts go on with
    its backward pass) and in two launches (FT_LK_FUSED_FB=0, read when the context is created)."""
    w, h = 640, 360
    prev, nxt = synth.shifted_pair(w, h, dx=3, dy=2, seed=60)
    nxt = nxt.copy()
    nxt[100:180, 200:320] = np.random.default_rng(2).integers(0, 256, (80, 120), dtype=np.uint8)
    pts = synth.random_points(w, h, 1000, seed=61, border=10)
    for win in (21, (21, 11), (24, 21), 17):   # (24, 21): three 12-column tasks per gradient patch row, two passes
        rp, rn = O.Pyramid(prev, win, 4), O.Pyramid(nxt, win, 4)
        flags = dev.FT_LK_USE_INITIAL_FLOW | dev.FT_LK_FORWARD_BACKWARD
        ref = O.lk_track(rp, rn, pts, win=win, max_level=4, flags=flags, fb_thresh=1.0)
        with _ctx(dev, w, h) as ctx:
            ctx.upload(0, 0, prev)
            ctx.upload(0, 1, nxt)
            ctx.build_pyramid([0, 0], [0, 1], 4, win)
            out = ctx.lk_track([dict(prev=(0, 0), next=(0, 1), prev_xy=pts)], dev.LKParams(win=win, max_level=4, flags=flags, fb_thresh=1.0))[0]
        _compare_lk(out, ref)
        assert 0 < (ref[1] == 0).sum() < len(pts)
        if win in (21, (21, 11)):
            monkeypatch.setenv("FT_LK_FUSED_FB", "0")
            with _ctx(dev, w, h) as ctx:
                ctx.upload(0, 0, prev)
                ctx.upload(0, 1, nxt)
                ctx.build_pyramid([0, 0], [0, 1], 4, win)
                two = ctx.lk_track([dict(prev=(0, 0), next=(0, 1), prev_xy=pts)], dev.LKParams(win=win, max_level=4, flags=flags, fb_thresh=1.0))[0]
            monkeypatch.delenv("FT_LK_FUSED_FB")
            for a, b in zip(out, two):
                assert np.array_equal(a, b)


@pytest.mark.gpu
def test_front_queue_gives_the_same_results(dev):
    """Pyramids and detector of the next frame on a queue of their own, enqueued while the current frame's tracking is
    still queued (three ring slots, fences): keypoints, positions and status equal the one-queue run's."""
    w, h = 640, 360
    frames = [synth.shifted_pair(w, h, dx=2 * k, dy=k, seed=90)[1] for k in range(5)]
    pts = synth.random_points(w, h, 1500, seed=91, border=12)

    def run(front):
        res = []
        with _ctx(dev, w, h, n_streams=1, n_slots=3) as ctx:
            ctx.set_front_queue(front)

            def stage(k):   # ingest + pyramid of frame k into its ring slot
                ctx.upload(0, k % 3, frames[k])
                ctx.build_pyramid([0], [k % 3], 3, 21, with_gradients=False)

            stage(0)
            stage(1)
            for k in range(1, len(frames)):
                cur, last = k % 3, (k - 1) % 3
                det = ctx.fast_grid([0], [cur], 20, 5, 4, 9, sync=False)
                out = ctx.lk_track([dict(prev=(0, last), next=(0, cur), prev_xy=pts)], dev.LKParams(win=21, max_level=3), sync=False)[0]
                tok = ctx.fence()
                if k + 1 < len(frames):   # the next frame's front work goes in before this frame's tracking is waited for
                    stage(k + 1)
                ctx.wait(tok)
                res.append((det[0].copy(), det[1].copy(), out[0].copy(), out[1].copy(), out[2].copy()))
        return res

    one, two = run(False), run(True)
    assert len(one) == len(two) == len(frames) - 1
    for a, b in zip(one, two):
        assert a[1][0] > 50 and a[3].sum() > 1000
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_lk_batched_jobs_and_gyro_seed(dev, lk_form):
    w, h = 376, 240
    K4 = np.array([300.0, 300.0, 188.0, 120.0], dtype=np.float32)
    R, _ = O.integrate_gyro(np.arange(0, 1, 0.005) + 0.001, np.tile([0.01, -0.02, 0.3], (200, 1)), 0.5, 0.55, np.eye(3))
    jobs, refs = [], []
    with _ctx(dev, w, h, n_streams=3) as ctx:
        for s in range(3):
            prev, nxt = synth.shifted_pair(w, h, dx=2 + s, dy=1 - s, seed=70 + s)
            pts = synth.random_points(w, h, 100 + 50 * s, seed=71 + s, border=8)
            ctx.upload(s, 0, prev)
            ctx.upload(s, 1, nxt)
            rp, rn = O.Pyramid(prev, 21, 3), O.Pyramid(nxt, 21, 3)
            guess = O.predict_flow(K4, R, pts)
            refs.append(O.lk_track(rp, rn, pts, guess=guess, win=21, max_level=3))
            jobs.append(dict(prev=(s, 0), next=(s, 1), prev_xy=pts, H=ctx.homography(K4, R)))
        assert np.array_equal(ctx.homography(K4, R), O.homography(K4, R))
        assert np.abs(ctx.predict_flow(O.homography(K4, R), jobs[0]["prev_xy"]) - O.predict_flow(K4, R, jobs[0]["prev_xy"])).max() <= 1e-4
        ctx.build_pyramid([0, 1, 2, 0, 1, 2], [0, 0, 0, 1, 1, 1], 3, 21)
        outs = ctx.lk_track(jobs, dev.LKParams(win=21, max_level=3))
    for out, ref in zip(outs, refs):
        _compare_lk(out, ref)


@pytest.mark.parametrize("shape,grid,need,thr", [((480, 752), (5, 5), 150, 20), ((720, 1280), (9, 5), 2000, 20),
                                                  ((480, 752), (5, 5), 30, 10), ((240, 376), (7, 3), 400, 35),
                                                  ((40, 59), (10, 2), 50, 20), ((479, 751), (5, 5), 150, 20),
                                                  ((121, 63), (1, 1), 40, 12), ((64, 7), (1, 2), 10, 5)])
def test_fast_grid_bit_exact(dev, shape, grid, need, thr):
    h, w = shape
    img = synth.texture(w, h, seed=80, n_shapes=400)
    gx, gy = grid
    per_cell = need // (gx * gy) + 1
    ref = O.grid_fast(img, need, gx, gy, thr)
    with _ctx(dev, w, h, max_per_cell=64) as ctx:
        ctx.upload(0, 0, img)
        out, n_out, cell_counts = ctx.fast_grid([0], [0], thr, gx, gy, per_cell)
    got = out[0, : n_out[0]]
    assert got.shape == ref.shape
    assert np.array_equal(got, ref)
    # NMS survivor counts per cell equal FAST on each cell view
    sx, sy = w // gx, h // gy
    ncx, ncy = w // sx, h // sy
    for c in range(ncx * ncy):
        x0, y0 = (c % ncx) * sx, (c // ncx) * sy
        assert cell_counts[c] == O.fast(img[y0:y0 + sy, x0:x0 + sx], thr).shape[0]


@pytest.mark.parametrize("shape,max_level,n", [((480, 752), 3, 3), ((479, 751), 2, 2), ((720, 1280), 4, 2), ((240, 376), 1, 2)])
@pytest.mark.parametrize("fuse", ["1", "0"])
def test_pyramid_then_detector_in_shared_launches(dev, shape, max_level, n, fuse, monkeypatch):
    """A detector call right behind the pyramid call of the same frames carries the pyramid's upper levels in its own
    launches (detector + level 1 -> 2, ranking + level 2 -> 3; FT_FRONT_FUSE=0: separate launches).  Every level and the
    ranked keypoints equal the CPU checker's; a second round with another call in between (which makes the library launch
    what it held back) does as well."""
    h, w = shape
    monkeypatch.setenv("FT_FRONT_FUSE", fuse)
    frames = [synth.texture(w, h, seed=300 + i, n_shapes=300) for i in range(2 * n)]
    with _ctx(dev, w, h, n_streams=n, max_per_cell=16) as ctx:
        for rnd in range(2):
            fr = frames[rnd * n:(rnd + 1) * n]
            ach = ctx.set_frames_host_pyramid(list(range(n)), [rnd] * n, fr, max_level=max_level, win=21)
            if rnd == 1:
                ctx.flush()
            out, n_out, _ = ctx.fast_grid(list(range(n)), [rnd] * n, 20, 5, 4, 8, sync=False)
            ctx.sync()
            for i, f in enumerate(fr):
                ref = O.Pyramid(f, 21, max_level)
                assert ach[i] == ref.levels - 1
                for l in range(ref.levels):
                    assert np.array_equal(ctx.download_level(i, rnd, l), ref.level(l)), f"round {rnd} image {i} level {l}"
                want = O.grid_fast(f, 8 * 20 - 1, 5, 4, 20)
                assert np.array_equal(out[i, : n_out[i]], want)


@pytest.mark.parametrize("shape,max_level,n,pad", [((480, 752), 3, 3, 0), ((479, 751), 2, 2, 5), ((720, 1280), 4, 2, 0), ((240, 376), 1, 2, 3), ((120, 188), 0, 2, 1)])
def test_device_frames_then_detector_in_shared_launches(dev, shape, max_level, n, pad):
    """Frames already in device memory (any stride and alignment): a detector call right behind the pyramid call carries the
    pyramid's upper levels in its own launches.  Every level and the ranked keypoints equal the CPU checker's; a second
    round with a flush in between, and a third where another pyramid call comes before the detector call, do as well."""
    import torch
    h, w = shape
    stride = w + pad
    frames = [synth.texture(w, h, seed=400 + i, n_shapes=300) for i in range(3 * n)]
    flat = torch.zeros(3 * n * h * stride + 64, dtype=torch.uint8, device="cuda")
    ptrs = []
    for i, im in enumerate(frames):
        padded = np.zeros((h, stride), np.uint8)
        padded[:, :w] = im
        off = (1 if pad else 0) + i * h * stride
        flat[off:off + h * stride] = torch.from_numpy(padded.reshape(-1)).cuda()
        ptrs.append(flat.data_ptr() + off)
    torch.cuda.synchronize()
    with _ctx(dev, w, h, n_streams=n, n_slots=3, max_per_cell=16) as ctx:
        for rnd in range(3):
            fr, pp = frames[rnd * n:(rnd + 1) * n], ptrs[rnd * n:(rnd + 1) * n]
            if rnd == 2:   # the frames of round 0 again into the slots of round 0, held back and then pushed out by the next call
                ctx.set_frames_device_pyramid(list(range(n)), [0] * n, ptrs[:n], w, h, stride, max_level, 21)
            ach = ctx.set_frames_device_pyramid(list(range(n)), [rnd] * n, pp, w, h, stride, max_level, 21)
            if rnd == 1:
                ctx.flush()
            out, n_out, _ = ctx.fast_grid(list(range(n)), [rnd] * n, 20, 5, 4, 8, sync=False)
            ctx.sync()
            for i, f in enumerate(fr):
                ref = O.Pyramid(f, 21, max_level)
                assert ach[i] == ref.levels - 1
                for l in range(ref.levels):
                    assert np.array_equal(ctx.download_level(i, rnd, l), ref.level(l)), f"round {rnd} image {i} level {l}"
                want = O.grid_fast(f, 8 * 20 - 1, 5, 4, 20)
                assert np.array_equal(out[i, : n_out[i]], want)
            if rnd == 2:
                for i, f in enumerate(frames[:n]):
                    assert np.array_equal(ctx.download_level(i, 0, 0), f)


def test_fast_grid_batched_and_plateaus(dev):
    rng = np.random.default_rng(7)
    imgs = [synth.texture(376, 240, seed=90 + i, n_shapes=300) for i in range(3)]
    imgs[2] = (imgs[2] // 32 * 32).astype(np.uint8)     # coarse quantisation: plateaus and score ties
    with _ctx(dev, 376, 240, n_streams=3) as ctx:
        for i, im in enumerate(imgs):
            ctx.upload(i, 0, im)
        out, n_out, _ = ctx.fast_grid([0, 1, 2], [0, 0, 0], 20, 5, 5, 7)
    for i, im in enumerate(imgs):
        ref = O.grid_fast(im, 150, 5, 5, 20)
        assert np.array_equal(out[i, : n_out[i]], ref)


def test_full_size_4k_frame_10k_features(dev, lk_form):
    """The largest configuration (3840x2160, up to 10 000 features, maxLevel 5, grid 26x15) against the CPU checker."""
    w, h = 3840, 2160
    prev, nxt = synth.shifted_pair(w, h, dx=5, dy=-3, seed=500)
    rp, rn = O.Pyramid(prev, 21, 5), O.Pyramid(nxt, 21, 5)
    with _ctx(dev, w, h, n_streams=1, max_levels=6, max_points=12000, max_grid_cells=26 * 15, max_per_cell=32) as ctx:
        ctx.set_frames_host_pyramid([0], [0], [prev], max_level=5, win=21)
        ctx.set_frames_host_pyramid([0], [1], [nxt], max_level=5, win=21)
        for l in range(rp.levels):
            assert np.array_equal(ctx.download_level(0, 0, l), rp.level(l)), f"level {l}"
        kp, n_kp, _ = ctx.fast_grid([0], [0], 8, 26, 15, 10000 // 390 + 1)
        ref = O.grid_fast(prev, 10000, 26, 15, 8, n_threads=8)
        assert np.array_equal(kp[0, : n_kp[0]], ref) and n_kp[0] > 2000
        pts = np.ascontiguousarray(ref[:10000, :2])
        xy, st, er = ctx.lk_track([dict(prev=(0, 0), next=(0, 1), prev_xy=pts)], dev.LKParams(win=21, max_level=5))[0]
    r_xy, r_st, r_er = O.lk_track(rp, rn, pts, win=21, max_level=5, n_threads=8)
    assert np.array_equal(st, r_st) and st.mean() > 0.95
    ok = st == 1
    assert np.array_equal(xy[ok], r_xy[ok])
    assert np.percentile(np.abs((xy[ok] - pts[ok]) - [5, -3]), 99) < 0.5   # and the flow is the known displacement


def test_properties_at_the_benchmark_batch_size(dev):
    """64 streams of 752x480 in one call, checked through properties that need no reference: detector output ranked and
    inside the cell insets, tracking a frame against itself is the identity with zero residual, and tracking against
    an integer-shifted copy returns exactly that shift for every feature whose window stays inside both frames."""
    S, w, h = 64, 752, 480
    big = [synth.texture(w + 16, h + 16, seed=700 + s) for s in range(S)]
    a = [np.ascontiguousarray(b[8:8 + h, 8:8 + w]) for b in big]
    sh = [np.ascontiguousarray(b[8 - 2:8 - 2 + h, 8 + 3:8 + 3 + w]) for b in big]      # content moves by (-3, +2)
    ids = list(range(S))
    with _ctx(dev, w, h, n_streams=S, max_levels=4, max_points=S * 200, max_grid_cells=25, max_per_cell=8) as ctx:
        ctx.set_frames_host_pyramid(ids, [0] * S, a, max_level=3, win=21)
        ctx.set_frames_host_pyramid(ids, [1] * S, sh, max_level=3, win=21)
        kp, n_kp, _ = ctx.fast_grid(ids, [0] * S, 20, 5, 5, 7)
        sx, sy = w // 5, h // 5
        for s in range(S):
            k = kp[s, : n_kp[s]]
            assert 0 < len(k) <= 25 * 7
            cell = (k[:, 1] // sy).astype(int) * 5 + (k[:, 0] // sx).astype(int)
            assert np.all(np.diff(cell) >= 0)                                            # concatenated by cell index
            same = np.diff(cell) == 0
            assert np.all(np.diff(k[:, 2])[same] <= 0)                                   # ranked inside a cell
            lx, ly = k[:, 0] % sx, k[:, 1] % sy
            assert lx.min() >= 3 and lx.max() < sx - 3 and ly.min() >= 3 and ly.max() < sy - 3 and k[:, 2].min() >= 20
        jobs_id = [dict(prev=(s, 0), next=(s, 0), prev_xy=kp[s, : n_kp[s], :2]) for s in range(S)]
        jobs_sh = [dict(prev=(s, 0), next=(s, 1), prev_xy=kp[s, : n_kp[s], :2]) for s in range(S)]
        out_id = ctx.lk_track(jobs_id, dev.LKParams(win=21, max_level=3))
        out_sh = ctx.lk_track(jobs_sh, dev.LKParams(win=21, max_level=3))
    n_inner = 0
    for s in range(S):
        p = kp[s, : n_kp[s], :2]
        xy, st, er = out_id[s]
        assert st.all() and np.array_equal(xy, p) and not er.any()
        xy, st, er = out_sh[s]
        inner = (p[:, 0] > 40) & (p[:, 0] < w - 40) & (p[:, 1] > 40) & (p[:, 1] < h - 40)
        assert st[inner].all()
        assert np.abs(xy[inner] - p[inner] - [-3, 2]).max() < 0.02 and er[inner].max() < 0.5
        n_inner += int(inner.sum())
    assert n_inner > S * 60


def test_ragged_empty_and_full_batches(dev, lk_form):
    """The shapes a caller can throw at the batched entry points: a full batch of 256 small images, jobs with no
    points, one point, and only lost points next to ordinary ones, a flat image without any corner."""
    n_img, w, h = 256, 96, 64
    rng = np.random.default_rng(11)
    prevs = [synth.texture(w, h, seed=300 + i, n_shapes=60) for i in range(n_img)]
    prevs[7] = np.full((h, w), 90, np.uint8)                                 # no texture at all
    nexts = [np.roll(p, (1, 2), axis=(0, 1)) for p in prevs]
    with _ctx(dev, w, h, n_streams=n_img, max_levels=3, max_points=4000, max_grid_cells=6, max_per_cell=8) as ctx:
        ids = list(range(n_img))
        ctx.set_frames_host_pyramid(ids, [0] * n_img, prevs, max_level=2, win=9)
        ctx.set_frames_host_pyramid(ids, [1] * n_img, nexts, max_level=2, win=9)
        kp, n_kp, _ = ctx.fast_grid(ids, [0] * n_img, 20, 3, 2, 4)
        for i in (0, 7, 100, 255):
            assert np.array_equal(kp[i, : n_kp[i]], O.grid_fast(prevs[i], 6 * 4 - 1, 3, 2, 20))
        assert n_kp[7] == 0
        jobs, counts = [], []
        for i in range(n_img):
            m = [0, 1, 12, 5][i % 4]
            pts = synth.random_points(w, h, m, seed=400 + i, border=6) if i % 8 != 3 else np.full((m, 2), -500.0, np.float32)
            jobs.append(dict(prev=(i, 0), next=(i, 1), prev_xy=pts))
            counts.append(m)
        outs = ctx.lk_track(jobs, dev.LKParams(win=9, max_level=2))
        for i in (0, 1, 2, 3, 7, 11, 254, 255):
            xy, st, er = outs[i]
            assert len(st) == counts[i]
            if counts[i] == 0:
                continue
            r_xy, r_st, r_er = O.lk_track(O.Pyramid(prevs[i], 9, 2), O.Pyramid(nexts[i], 9, 2), jobs[i]["prev_xy"], win=9, max_level=2)
            assert np.array_equal(st, r_st)
            ok = st == 1
            assert np.array_equal(xy[ok], r_xy[ok])
            if i % 8 == 3:
                assert not st.any()                                          # far outside the frame: all lost
        with pytest.raises(dev.FtError):                                     # one image more than a batch may hold
            ctx.fast_grid(ids + [0], [0] * (n_img + 1), 20, 3, 2, 4)


CALIBS = [np.array([458.7, 457.3, 367.2, 248.4, -0.28, 0.074, 2e-4, 2e-5]), np.array([600., 600., 376., 240., 0, 0, 0, 0])]


@pytest.mark.parametrize("n,n_out,seed", [(150, 30, 1), (8, 0, 2), (400, 200, 3), (64, 50, 4), (3000, 900, 5)])   # last: beyond the LDS-resident part
def test_reject_outliers_equals_oracle(dev, n, n_out, seed):
    """Standalone rejection on synthetic two-view matches: the status bytes equal the CPU checker's bit for bit
    (same hash-drawn samples, same double-precision elimination, same float error measure)."""
    ctx = _ctx(dev, 64, 64, max_points=2 * n + 400)
    sets, want = [], []
    rng = np.random.default_rng(seed)
    for k, calib in enumerate(CALIBS):
        prev, nxt, _ = synth.two_view_matches(n, n_out, calib, seed=seed * 10 + k)
        status = (rng.random(n) > 0.1).astype(np.uint8)
        sets.append(dict(prev_xy=prev, next_xy=nxt, status=status, calib_prev=calib, calib_next=CALIBS[1 - k]))
        want.append(O.reject_outliers(prev, nxt, status, calib, CALIBS[1 - k], hypotheses=200, seed=seed)[0])
    with ctx:
        got = ctx.reject_outliers(sets, hypotheses=200, seed=seed)
    for g, w_ in zip(got, want):
        assert np.array_equal(g, w_)
    assert sum(int(w_.sum()) for w_ in want) > 0


@pytest.mark.parametrize("sizes,n_out_frac,hyp", [((513, 150, 1024, 1025, 7, 2600), 0.3, 256), ((9000,), 0.05, 256), ((1500, 1500, 40), 0.75, 256),
                                                  ((17000, 600), 0.2, 64)])
def test_reject_outliers_split_pairs_equal_oracle(dev, sizes, n_out_frac, hyp):
    """Pairs of more than 512 points are spread over several workgroups that meet at a barrier after every group of
    four models: sizes around the chunk boundaries, small pairs in the same launch, many outliers (many rounds before
    the search stops, or no model at all), a pair beyond 32 chunks.  The masks equal the CPU checker's."""
    sets, want = [], []
    for k, n in enumerate(sizes):
        calib = CALIBS[k % 2]
        prev, nxt, _ = synth.two_view_matches(n, int(n * n_out_frac), calib, seed=100 + k)
        status = (np.random.default_rng(k).random(n) > 0.05).astype(np.uint8)
        sets.append(dict(prev_xy=prev, next_xy=nxt, status=status, calib_prev=calib))
        want.append(O.reject_outliers(prev, nxt, status, calib, calib, hypotheses=hyp, seed=7)[0])
    with _ctx(dev, 64, 64, max_points=sum(sizes) + 16) as ctx:
        for _ in range(2):                                                   # the second launch finds the counters as the first left them
            got = ctx.reject_outliers(sets, hypotheses=hyp, seed=7)
            for g, w_ in zip(got, want):
                assert np.array_equal(g, w_)


@pytest.mark.parametrize("n,share", sorted(synth.REJECTION_BARS))
def test_reject_outliers_precision_and_recall_against_ground_truth(dev, n, share):
    """The device's masks against the known random matches, at the same bars as the CPU checker's (tests/synth.py)."""
    p_bar, r_bar = synth.REJECTION_BARS[(n, share)]
    with _ctx(dev, 64, 64, max_points=n + 16) as ctx:
        p, r = synth.rejection_scores(lambda a, b, c, seed: ctx.reject_outliers([dict(prev_xy=a, next_xy=b, calib_prev=c)], hypotheses=0, seed=seed)[0], n, share)
    assert p >= p_bar and r >= r_bar, (p, r)


def test_reject_outliers_more_parts_than_the_device_holds(dev, monkeypatch):
    """A launch whose split pairs would need more than 128 workgroups runs one workgroup per pair instead, and
    FT_RO_SPLIT=0 does so always: same masks either way."""
    calib = CALIBS[1]
    sets, want = [], []
    for k in range(50):
        n = 1100 if k % 2 == 0 else 30
        prev, nxt, _ = synth.two_view_matches(n, n // 5, calib, seed=300 + k)
        sets.append(dict(prev_xy=prev, next_xy=nxt, calib_prev=calib))
    want = [O.reject_outliers(q["prev_xy"], q["next_xy"], np.ones(len(q["prev_xy"]), np.uint8), calib, calib, hypotheses=32, seed=1)[0] for q in sets[:6]]
    with _ctx(dev, 64, 64, max_points=60000) as ctx:
        got = ctx.reject_outliers(sets, hypotheses=32, seed=1)               # 25 * 3 + 25 = 100 workgroups: split
        got_few = ctx.reject_outliers(sets[:6], hypotheses=32, seed=1)
    monkeypatch.setenv("FT_RO_SPLIT", "0")
    with _ctx(dev, 64, 64, max_points=60000) as ctx:
        got_one = ctx.reject_outliers(sets, hypotheses=32, seed=1)
    big = [dict(prev_xy=np.tile(q["prev_xy"], (3, 1)), next_xy=np.tile(q["next_xy"], (3, 1)), calib_prev=calib) for q in sets[:40:2]]
    monkeypatch.delenv("FT_RO_SPLIT")
    with _ctx(dev, 64, 64, max_points=70000) as ctx:
        got_big = ctx.reject_outliers(big, hypotheses=32, seed=1)            # 20 * 7 = 140 workgroups: not split
    for i in range(6):
        assert np.array_equal(got[i], want[i]) and np.array_equal(got_few[i], want[i])
    for a, b in zip(got, got_one):
        assert np.array_equal(a, b)
    w0 = O.reject_outliers(big[0]["prev_xy"], big[0]["next_xy"], np.ones(3300, np.uint8), calib, calib, hypotheses=32, seed=1)[0]
    assert np.array_equal(got_big[0], w0)


def test_reject_outliers_barrier_that_gives_up_is_an_error_not_a_hang(dev, monkeypatch):
    """The workgroups of a split pair wait for each other with a bounded number of polls.  With the bound at one poll
    some workgroup gives up: the call returns an error (no mask is delivered as if it were a result), the queue is not
    hung, and a context with the normal bound then gets the checker's masks again."""
    calib = CALIBS[0]
    sets = []
    for k in range(6):
        prev, nxt, _ = synth.two_view_matches(3000, 900, calib, seed=500 + k)
        sets.append(dict(prev_xy=prev, next_xy=nxt, calib_prev=calib))
    monkeypatch.setenv("FT_RO_SPIN_LIMIT", "1")
    with _ctx(dev, 64, 64, max_points=20000) as ctx:
        failed = 0
        for _ in range(5):
            try:
                ctx.reject_outliers(sets, hypotheses=64, seed=3)
            except dev.FtError as e:
                failed += 1
                assert e.code == -3 and "barrier" in str(e)
        assert failed >= 1
        small = ctx.reject_outliers([dict(prev_xy=sets[0]["prev_xy"][:200], next_xy=sets[0]["next_xy"][:200], calib_prev=calib)], hypotheses=64, seed=3)[0]
        assert np.array_equal(small, O.reject_outliers(sets[0]["prev_xy"][:200], sets[0]["next_xy"][:200], np.ones(200, np.uint8), calib, hypotheses=64, seed=3)[0])
    monkeypatch.delenv("FT_RO_SPIN_LIMIT")
    with _ctx(dev, 64, 64, max_points=20000) as ctx:
        got = ctx.reject_outliers(sets, hypotheses=64, seed=3)
    for q, g in zip(sets, got):
        assert np.array_equal(g, O.reject_outliers(q["prev_xy"], q["next_xy"], np.ones(3000, np.uint8), calib, hypotheses=64, seed=3)[0])


def test_reject_outliers_degenerate_sets(dev):
    calib = CALIBS[0]
    prev, nxt, _ = synth.two_view_matches(100, 0, calib, seed=5)
    with _ctx(dev, 64, 64, max_points=600) as ctx:
        few = ctx.reject_outliers([dict(prev_xy=prev[:7], next_xy=nxt[:7], calib_prev=calib)])[0]
        assert not few.any()                                                  # fewer than 8 points: no model
        flat = ctx.reject_outliers([dict(prev_xy=prev, next_xy=prev * 0 + 5.0, calib_prev=calib)])[0]
        assert np.array_equal(flat, O.reject_outliers(prev, prev * 0 + 5.0, np.ones(100, np.uint8), calib)[0])
        empty = ctx.reject_outliers([dict(prev_xy=prev[:0], next_xy=nxt[:0], calib_prev=calib),
                                     dict(prev_xy=prev, next_xy=nxt, calib_prev=calib)])
        assert len(empty[0]) == 0 and np.array_equal(empty[1], O.reject_outliers(prev, nxt, np.ones(100, np.uint8), calib)[0])
        with pytest.raises(dev.FtError):                                  # more points than the context was sized for
            ctx.reject_outliers([dict(prev_xy=np.zeros((601, 2)), next_xy=np.zeros((601, 2)), calib_prev=calib)])


def test_lk_with_outlier_rejection_equals_lk_then_oracle_rejection(dev, lk_form):
    """FT_LK_REJECT_OUTLIERS = the plain LK result with the CPU checker's rejection applied to it."""
    w, h = 640, 400
    a, b = synth.shifted_pair(w, h, 3, -2, seed=11)
    pts = synth.random_points(w, h, 300, seed=3)
    calib = CALIBS[0]
    job = dict(prev=(0, 0), next=(0, 1), prev_xy=pts, calib_prev=calib)
    with _ctx(dev, w, h, max_points=600) as ctx:
        ctx.upload(0, 0, a)
        ctx.upload(0, 1, b)
        ctx.build_pyramid([0, 0], [0, 1], 3, 21)
        plain = ctx.lk_track([job], dev.LKParams(21, 3))[0]
        strict = ctx.lk_track([job], dev.LKParams(21, 3, flags=dev.FT_LK_USE_INITIAL_FLOW | dev.FT_LK_REJECT_OUTLIERS,
                                                  ransac_hypotheses=128, ransac_seed=9))[0]
        with pytest.raises(dev.FtError):                                  # the flag needs the camera models
            ctx.lk_track([dict(prev=(0, 0), next=(0, 1), prev_xy=pts)],
                         dev.LKParams(21, 3, flags=dev.FT_LK_USE_INITIAL_FLOW | dev.FT_LK_REJECT_OUTLIERS))
    assert np.array_equal(plain[0], strict[0]) and np.array_equal(plain[2], strict[2])      # positions, residuals untouched
    want = O.reject_outliers(pts, plain[0], plain[1], calib, hypotheses=128, seed=9)[0]
    assert np.array_equal(strict[1], want)


@pytest.mark.parametrize("n", [300, 3000])
def test_lk_input_block_paths_agree(dev, n, monkeypatch):
    """A tracking batch's input block reaches the device in one of three ways: written by the host straight into device
    memory (where the device has a large BAR: the default there), read in place from pinned memory by the tracking
    wavefronts, which leave device copies for the rejection kernel (small batches without the BAR), or fetched by a kernel.
    Positions, status after rejection and residuals are the same."""
    w, h = 640, 360
    prev, nxt = synth.shifted_pair(w, h, dx=3, dy=-2, seed=77)
    pts = synth.random_points(w, h, n, seed=78, border=10)
    calib = np.array([420.0, 418.0, 318.0, 182.0, -0.28, 0.07, 2e-4, 1e-5])
    prm = dev.LKParams(21, 3, flags=dev.FT_LK_USE_INITIAL_FLOW | dev.FT_LK_REJECT_OUTLIERS, ransac_hypotheses=64, ransac_seed=3)

    def run():
        with _ctx(dev, w, h) as ctx:
            ctx.upload(0, 0, prev)
            ctx.upload(0, 1, nxt)
            ctx.build_pyramid([0, 0], [0, 1], 3, 21, with_gradients=False)
            return ctx.lk_track([dict(prev=(0, 0), next=(0, 1), prev_xy=pts, calib_prev=calib, calib_next=calib)], prm)[0]

    default = run()
    monkeypatch.setenv("FT_LK_BAR_INPUT", "0")
    in_place_or_fetched = run()
    monkeypatch.setenv("FT_LK_DIRECT_MAX", "0")
    fetched = run()
    monkeypatch.delenv("FT_LK_DIRECT_MAX")
    monkeypatch.delenv("FT_LK_BAR_INPUT")
    assert default[1].sum() > 0.8 * n
    for other in (in_place_or_fetched, fetched):
        for a, b in zip(default[:3], other[:3]):
            assert np.array_equal(a, b)


STEREO_RIGS = [
    # undistorted twin cameras, pure sideways baseline: the epipolar lines are the image rows
    ((460.0, 460.0, 320.0, 200.0, 0, 0, 0, 0), (460.0, 460.0, 320.0, 200.0, 0, 0, 0, 0), (0, 0, 0), (-0.11, 0, 0)),
    # distorted, slightly different cameras, slightly rotated rig
    ((458.654, 457.296, 317.2, 198.4, -0.2834, 0.0740, 1.9e-4, 1.8e-5), (457.587, 456.134, 322.0, 203.2, -0.2837, 0.0745, -1.0e-4, -3.6e-5),
     (0.002, -0.004, 0.003), (-0.11, 0.001, 0.0005)),
]


@pytest.mark.parametrize("c0,c1,rot,t", STEREO_RIGS)
@pytest.mark.parametrize("threshold", [3.0, 0.5, 0.0])
def test_lk_stereo_job_equals_oracle_seed_track_check(dev, c0, c1, rot, t, threshold):
    """A job with stereo_Rt = seed from the extrinsic rotation, LK, frame + epipolar test, all on the device; against the
    CPU checker's three steps.  Right image = left image displaced along the rows (a fronto-parallel plane)."""
    w, h = 640, 400
    left, right = synth.shifted_pair(w, h, dx=-7, dy=0, seed=90)
    pts = synth.random_points(w, h, 400, seed=91, border=-6)
    Rt = np.concatenate([synth.rotation_xyz(rot).reshape(9), np.asarray(t, dtype=np.float64)])
    pl, pr = O.Pyramid(left, 21, 3), O.Pyramid(right, 21, 3)
    guess = O.stereo_seed(pts, c0, c1, Rt)
    r_xy, r_st, r_err = O.lk_track(pl, pr, pts, guess=guess, win=21, max_level=3)
    want = O.stereo_check(pts, r_xy, r_st, c0, c1, Rt, w, h, threshold)
    with _ctx(dev, w, h) as ctx:
        ctx.upload(0, 0, left)
        ctx.upload(0, 1, right)
        ctx.build_pyramid([0, 0], [0, 1], 3, 21, with_gradients=False)
        plain = dict(prev=(0, 0), next=(0, 1), prev_xy=pts, guess=guess)
        stereo = dict(prev=(0, 0), next=(0, 1), prev_xy=pts, calib_prev=c0, calib_next=c1, stereo_Rt=Rt)
        # flags without USE_INITIAL_FLOW for the stereo call: its guesses must be used all the same
        g_plain, g_stereo = ctx.lk_track([plain, stereo], dev.LKParams(win=21, max_level=3, stereo_threshold=threshold))
        only = ctx.lk_track([stereo], dev.LKParams(win=21, max_level=3, flags=0, stereo_threshold=threshold))[0]
    assert np.array_equal(g_plain[0], r_xy) and np.array_equal(g_plain[1], r_st)      # the plain job of the batch is untouched
    assert np.array_equal(g_stereo[0], r_xy) and np.array_equal(g_stereo[2], r_err)   # same seed -> same track
    assert np.array_equal(g_stereo[1], want)
    assert np.array_equal(only[0], r_xy) and np.array_equal(only[1], want)
    if threshold >= 3.0 and not any(rot):
        assert want.sum() >= 0.9 * r_st.sum()        # a row-wise displacement fits a sideways baseline
    if threshold > 0:
        assert want.sum() <= r_st.sum()


def test_lk_stereo_job_with_outlier_rejection_and_missing_calibration(dev):
    w, h = 640, 400
    c0, c1, rot, t = STEREO_RIGS[1]
    left, right = synth.shifted_pair(w, h, dx=-5, dy=0, seed=92)
    pts = synth.random_points(w, h, 300, seed=93, border=4)
    Rt = np.concatenate([synth.rotation_xyz(rot).reshape(9), np.asarray(t, dtype=np.float64)])
    pl, pr = O.Pyramid(left, 15, 2), O.Pyramid(right, 15, 2)
    r_xy, r_st, _ = O.lk_track(pl, pr, pts, guess=O.stereo_seed(pts, c0, c1, Rt), win=15, max_level=2)
    want = O.reject_outliers(pts, r_xy, O.stereo_check(pts, r_xy, r_st, c0, c1, Rt, w, h, 3.0), c0, c1, hypotheses=64, seed=5)[0]
    with _ctx(dev, w, h) as ctx:
        ctx.upload(0, 0, left)
        ctx.upload(0, 1, right)
        ctx.build_pyramid([0, 0], [0, 1], 2, 15, with_gradients=False)
        prm = dev.LKParams(win=15, max_level=2, flags=dev.FT_LK_USE_INITIAL_FLOW | dev.FT_LK_REJECT_OUTLIERS, ransac_hypotheses=64, ransac_seed=5)
        got = ctx.lk_track([dict(prev=(0, 0), next=(0, 1), prev_xy=pts, calib_prev=c0, calib_next=c1, stereo_Rt=Rt)], prm)[0]
        assert np.array_equal(got[0], r_xy) and np.array_equal(got[1], want)
        with pytest.raises(dev.FtError):
            ctx.lk_track([dict(prev=(0, 0), next=(0, 1), prev_xy=pts, stereo_Rt=Rt)], dev.LKParams(win=15, max_level=2))


@pytest.mark.parametrize("n_common,n0,n1,flips,ratio,seed", [(120, 40, 60, 12, 0.7, 1), (700, 300, 500, 40, 0.85, 2), (5, 0, 0, 3, 0.75, 3),
                                                             (300, 0, 0, 90, 0.95, 4)])
def test_match_descriptors_equals_oracle(dev, n_common, n0, n1, flips, ratio, seed):
    """Hamming 2-NN on the device + ratio/symmetry tests = the CPU checker, match for match (integers: exact)."""
    d0, d1, pairs = synth.descriptor_sets(n_common, n0, n1, flip_bits=flips, seed=seed)
    # duplicates make equal distances, where the lower index must win on both sides
    d1[len(d1) // 2] = d1[0]
    want = O.match_descriptors(d0, d1, knn_ratio=ratio)
    with _ctx(dev, 64, 64) as ctx:
        got = ctx.match_descriptors(d0, d1, knn_ratio=ratio)
        assert np.array_equal(got, want)
        assert len(ctx.match_descriptors(d0[:0], d1)) == 0 and len(ctx.match_descriptors(d0, d1[:1])) == 0
        again = ctx.match_descriptors(d1, d0, knn_ratio=ratio)          # a second call reuses the buffers
        assert np.array_equal(again, O.match_descriptors(d1, d0, knn_ratio=ratio))
    if flips <= 12:
        assert len(want) >= n_common - 2


def test_error_codes(dev):
    with _ctx(dev, 128, 96, n_streams=1) as ctx:
        with pytest.raises(dev.FtError) as e:
            ctx.build_pyramid([0], [0], 3, 15)          # no frame yet
        assert e.value.code == -5
        with pytest.raises(dev.FtError) as e:
            ctx.upload(0, 0, np.zeros((200, 200), np.uint8))
        assert e.value.code == -2
        with pytest.raises(dev.FtError) as e:
            ctx.upload(5, 0, np.zeros((96, 128), np.uint8))
        assert e.value.code == -1
        ctx.upload(0, 0, np.zeros((96, 128), np.uint8))
        with pytest.raises(dev.FtError) as e:            # gradient images were never built
            ctx.download_grad(0, 0, 0)
        assert e.value.code == -5
        with pytest.raises(dev.FtError) as e:            # window larger than the kernel supports
            ctx.lk_track([dict(prev=(0, 0), next=(0, 0), prev_xy=np.zeros((4, 2), np.float32))], dev.LKParams(win=41, max_level=2))
        assert e.value.code == -1


def test_fence_wait_delivers_only_earlier_async_results(dev):
    """ft_fence / ft_wait: outputs of async calls made before the fence are delivered by the wait, outputs of calls
    made after it only by a later wait or sync; a stale or unknown token is refused."""
    imgs = [synth.texture(376, 240, seed=40 + i, n_shapes=300) for i in range(3)]
    with _ctx(dev, 376, 240, n_streams=3) as ctx:
        for i, im in enumerate(imgs):
            ctx.upload(i, 0, im)
        ref = [ctx.fast_grid([i], [0], 20, 5, 5, 7) for i in range(3)]
        out0, n0, _ = ctx.fast_grid([0], [0], 20, 5, 5, 7, sync=False)
        tok = ctx.fence()
        out1, n1, _ = ctx.fast_grid([1], [0], 20, 5, 5, 7, sync=False)
        ctx.build_pyramid([2], [0], 2, 15)
        ctx.wait(tok)
        assert n0[0] == ref[0][1][0] and np.array_equal(out0, ref[0][0])
        tok2 = ctx.fence()
        ctx.wait(tok2)
        assert n1[0] == ref[1][1][0] and np.array_equal(out1, ref[1][0])
        # a third undelivered detector result forces a full queue wait and still delivers everything in order
        outs = [ctx.fast_grid([i], [0], 20, 5, 5, 7, sync=False) for i in range(3)]
        ctx.sync()
        for i in range(3):
            assert np.array_equal(outs[i][0], ref[i][0]) and outs[i][1][0] == ref[i][1][0]
        with pytest.raises(dev.FtError) as e:
            ctx.wait(12345)
        assert e.value.code == -1
        for _ in range(20):
            ctx.fence()
        with pytest.raises(dev.FtError) as e:   # the ring entry of the first token has been reused
            ctx.wait(tok)
        assert e.value.code == -1
