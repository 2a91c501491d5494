"""Seeded synthetic images for the parity tests (numpy only)."""
import numpy as np


def texture(w, h, seed=0, n_waves=48, n_shapes=200):
    """Band-limited sinusoid mix plus random filled rectangles/discs (gives FAST corners), quantised to u8."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.zeros((h, w), dtype=np.float64)
    for _ in range(n_waves):
        period = rng.uniform(8, 128)
        ang = rng.uniform(0, 2 * np.pi)
        ph = rng.uniform(0, 2 * np.pi)
        amp = rng.uniform(0.3, 1.0)
        img += amp * np.sin(2 * np.pi * (xx * np.cos(ang) + yy * np.sin(ang)) / period + ph)
    img = 128 + 40 * img / np.sqrt(n_waves / 2)
    for _ in range(n_shapes):
        cx, cy = rng.uniform(0, w), rng.uniform(0, h)
        s = rng.uniform(4, 24)
        val = rng.uniform(-70, 70)
        if rng.random() < 0.5:
            ang = rng.uniform(0, np.pi / 2)
            u = (xx - cx) * np.cos(ang) + (yy - cy) * np.sin(ang)
            v = -(xx - cx) * np.sin(ang) + (yy - cy) * np.cos(ang)
            m = (np.abs(u) < s) & (np.abs(v) < s * rng.uniform(0.4, 1.0))
        else:
            m = (xx - cx) ** 2 + (yy - cy) ** 2 < s * s
        img[m] += val
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def shifted_pair(w, h, dx, dy, seed=0, margin=16):
    """Two crops of one larger texture, the second displaced by integer (dx, dy): prev(x,y) == next(x+dx, y+dy)."""
    big = texture(w + 2 * margin, h + 2 * margin, seed)
    prev = big[margin:margin + h, margin:margin + w]
    nxt = big[margin - dy:margin - dy + h, margin - dx:margin - dx + w]
    return np.ascontiguousarray(prev), np.ascontiguousarray(nxt)


def smooth_field(w, h, seed=0):
    """Analytic band-limited image function f(x, y) -> float, for sub-pixel shift tests."""
    rng = np.random.default_rng(seed)
    comps = [(rng.uniform(24, 96), rng.uniform(0, 2 * np.pi), rng.uniform(0, 2 * np.pi), rng.uniform(0.3, 1.0)) for _ in range(32)]

    def f(x, y):
        v = np.zeros(np.broadcast(x, y).shape, dtype=np.float64)
        for period, ang, ph, amp in comps:
            v += amp * np.sin(2 * np.pi * (x * np.cos(ang) + y * np.sin(ang)) / period + ph)
        return 128 + 90 * v / np.sqrt(len(comps) / 2) / 2

    return f


def render(f, w, h, dx=0.0, dy=0.0):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    return np.clip(np.rint(f(xx - dx, yy - dy)), 0, 255).astype(np.uint8)


def random_points(w, h, n, seed=0, border=0.0):
    rng = np.random.default_rng(seed)
    return np.stack([rng.uniform(border, w - 1 - border, n), rng.uniform(border, h - 1 - border, n)], axis=1).astype(np.float32)
