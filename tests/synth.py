"""Seeded synthetic ground-texture frames (SURVEY.md 8(d) "Synthetic inputs").

base = gaussian_blur(N(0,1) noise, sigma=1.5 px) on a 2H x 2W canvas, min-max normalised, quantised
to u8; a frame is the H x W window at integer offset (dy, dx) from the canvas centre, optionally
after rotating the canvas by theta degrees about its centre (bilinear).  Used by tests/ and bench.py.
"""
import numpy as np
from scipy import ndimage


def canvas(seed, H, W, sigma=1.5):
    rng = np.random.default_rng(seed)
    c = ndimage.gaussian_filter(rng.standard_normal((2 * H, 2 * W)).astype(np.float32), sigma, mode="wrap")
    c = (c - c.min()) / (c.max() - c.min())
    return np.round(c * 255).astype(np.uint8)


def window(cv, H, W, dy=0, dx=0, theta=0.0):
    """camera window moved by (dy, dx) px and rotated by theta degrees over the static canvas."""
    src = cv
    if theta != 0.0:
        src = ndimage.rotate(cv.astype(np.float32), theta, reshape=False, order=1, mode="wrap")
        src = np.clip(np.round(src), 0, 255).astype(np.uint8)
    y0, x0 = (cv.shape[0] - H) // 2 + dy, (cv.shape[1] - W) // 2 + dx
    return np.ascontiguousarray(src[y0:y0 + H, x0:x0 + W])


def make_pair(seed, H, W, dy, dx, theta=0.0):
    cv = canvas(seed, H, W)
    return window(cv, H, W), window(cv, H, W, dy, dx, theta)


def make_batch(n, H, W, seed0=0, max_shift=None, max_theta=10.0, half_degree=False):
    """n pairs with dy,dx ~ U{-max_shift..max_shift}, theta ~ U(-max_theta, max_theta)."""
    if max_shift is None:
        max_shift = max(1, min(H, W) // 10)
    rng = np.random.default_rng(10_000 + seed0)
    keys = np.empty((n, H, W), np.uint8)
    curs = np.empty((n, H, W), np.uint8)
    motions = []
    for i in range(n):
        dy, dx = (int(v) for v in rng.integers(-max_shift, max_shift + 1, 2))
        th = float(rng.uniform(-max_theta, max_theta)) if max_theta > 0 else 0.0
        if half_degree:
            th = round(th * 2) / 2
        keys[i], curs[i] = make_pair(seed0 + i, H, W, dy, dx, th)
        motions.append((dy, dx, th))
    return keys, curs, motions


def make_unique_batch(n, H, W, seed0=0, max_theta=10.0, ncanvas=32, max_shift=48, base_shift=60):
    """n pairs with pairwise DIFFERENT images: pair i uses canvas i % ncanvas and its own key window on it; the current
    frame is the key window moved by (dy, dx) and rotated by theta about the key window's centre."""
    rng = np.random.default_rng(seed0)
    cvs = [canvas(seed0 + c, H, W) for c in range(min(ncanvas, n))]
    keys = np.empty((n, H, W), np.uint8)
    curs = np.empty((n, H, W), np.uint8)
    motions = []
    for i in range(n):
        cv = cvs[i % len(cvs)]
        by, bx = (int(v) for v in rng.integers(-base_shift, base_shift + 1, 2))
        dy, dx = (int(v) for v in rng.integers(-max_shift, max_shift + 1, 2))
        th = float(rng.uniform(-max_theta, max_theta)) if max_theta > 0 else 0.0
        keys[i] = window(cv, H, W, by, bx)
        if th != 0.0:
            curs[i] = window(np.roll(cv, (-by, -bx), axis=(0, 1)), H, W, dy, dx, th)    # key centre moved to the canvas centre first
        else:
            curs[i] = window(cv, H, W, by + dy, bx + dx)
        motions.append((dy, dx, th))
    return keys, curs, motions
