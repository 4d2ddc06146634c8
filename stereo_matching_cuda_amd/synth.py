"""Seeded synthetic stereo pairs: blurred random texture, per-row disparity ramp.

Data generation only (numpy/scipy on the host); nothing here is on the measured path.
"""
import numpy as np

SEEDS = {"kitti": 20150101, "motorcycle": 20140101, "4k": 38402160}
SHAPES = {"tsukuba": (384, 288, 16), "kitti": (1242, 375, 192), "motorcycle": (2964, 2000, 280),
          "4k": (3840, 2160, 512)}


def _blur(a, sigma):
    from scipy.ndimage import gaussian_filter
    return gaussian_filter(a.astype(np.float32), sigma=(sigma, sigma, 0) if a.ndim == 3 else sigma,
                           mode="nearest")


def row_disparity(h, size_d):
    y = np.arange(h)
    if size_d >= 48:
        return 5 + (size_d - 47) * y // h  # ramp inside [5, D-42]
    return np.minimum(size_d - 1, 1 + (max(size_d - 2, 0)) * y // h)


def gen_pair(w, h, size_d, seed):
    """Returns (left_gray, right_gray) uint8 (h, w).  left[y, x] == right[y, x - disp(y)]."""
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, size=(h, w + size_d, 3), dtype=np.uint8)
    base = np.clip(np.rint(_blur(base, 1.5)), 0, 255).astype(np.uint8)
    # gray weights of the reference (SystemIncludes.h:7-9), double precision, truncation
    g = (0.299 * base[..., 0].astype(np.float64) + 0.587 * base[..., 1] + 0.0721 * base[..., 2])
    gray = g.astype(np.uint8)
    disp = row_disparity(h, size_d)
    left = np.ascontiguousarray(gray[:, :w])
    cols = np.arange(w)[None, :] + disp[:, None]
    right = np.take_along_axis(gray, cols, axis=1)
    return left, np.ascontiguousarray(right)
