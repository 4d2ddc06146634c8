"""numpy front end of the host-pointer stage API -- same names, argument meaning and in/out
behaviour as the reference's per-stage host functions (host arrays in, host arrays out,
synchronous).  Every function is a thin call into the C-ABI (include/smx.h); all arithmetic
happens in the HIP kernels.

Reference signatures (stereo_matching_cuda/*.cuh):
  rgb_to_grayscale(h_rgb, n, channels, compare)                       rgb_to_grayscale.cuh:7
  compute_cost(i1, i2, cost, w1, w2, h1, h2, dmin, compare)           costVolume.cuh:7
  compute_guided_filter(i, cost, filter_cost, disp_map, mean, w, h, size_d, dmin, compare)
                                                                       guidedFilter.cuh:7
  integral(image, integral, width, height)                            integral.cuh:3
  detect_occlusion(dL, dR, dOcclusion, dmapl, dmapr, w, h)            occlusion.cuh:8
  fill_occlusion(disparity, w, h, vMin)                               occlusion.cuh:14
"""
import ctypes as C

import numpy as np

from . import _lib

WTA_INIT_BITS = 0x7F7F7F7F  # main.cu:112: memset(best_cost, 9999999.0f, ...) sets every byte to 0x7F


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    a = np.ascontiguousarray(a, dtype=dt)
    return a


def _params(p):
    return p if p is not None else _lib.default_params()


def rgb_to_grayscale(h_rgb, params=None):
    """(h, w, ch>=3) u8 -> (h, w) u8.  rgb_to_grayscale.cu:25-73."""
    rgb = _c(h_rgb, np.uint8)
    if rgb.ndim != 3 or rgb.shape[2] < 3:
        raise ValueError("rgb_to_grayscale expects an (h, w, channels>=3) uint8 array")
    h, w, ch = rgb.shape
    gray = np.empty((h, w), np.uint8)
    _lib.check(_lib.lib().smx_rgb_to_grayscale(C.byref(_params(params)), _ptr(rgb), h * w, ch, _ptr(gray)))
    return gray


def compute_cost(i1, i2, size_d, dmin, params=None):
    """Cost volume [z][y][x], slice z has label dmin + z.  costVolume.cu:4-84."""
    i1, i2 = _c(i1, np.uint8), _c(i2, np.uint8)
    if i1.ndim != 2 or i2.ndim != 2:
        raise ValueError("compute_cost expects two (h, w) uint8 images")
    h1, w1 = i1.shape
    h2, w2 = i2.shape
    cost = np.empty((size_d, h1, w1), np.float32)
    _lib.check(_lib.lib().smx_compute_cost(C.byref(_params(params)), _ptr(i1), _ptr(i2), _ptr(cost),
                                           w1, w2, h1, h2, size_d, dmin))
    return cost


def integral(image):
    """Integral image with the reference's sequential f32 addition order.  integral.cu:3-51."""
    image = _c(image, np.float32)
    if image.ndim != 2:
        raise ValueError("integral expects an (h, w) float32 image")
    h, w = image.shape
    out = np.empty((h, w), np.float32)
    _lib.check(_lib.lib().smx_integral(_ptr(image), _ptr(out), w, h))
    return out


def filter(image, params=None):
    """The reference's dead `filter()` (filter.cu:117-207): direct zero-padded box filter.
    Returns (mean u8, var f32) like its two output arguments."""
    image = _c(image, np.uint8)
    if image.ndim != 2:
        raise ValueError("filter expects an (h, w) uint8 image")
    h, w = image.shape
    mean = np.empty((h, w), np.uint8)
    var = np.empty((h, w), np.float32)
    _lib.check(_lib.lib().smx_filter(C.byref(_params(params)), _ptr(image), w, h, _ptr(mean), _ptr(var)))
    return mean, var


def init_wta(h, w):
    """best/dmap presets of main.cu:112-118."""
    best = np.full((h, w), WTA_INIT_BITS, np.uint32).view(np.float32)
    dmap = np.zeros((h, w), np.float32)
    return best, dmap


def compute_guided_filter(i, cost, filter_cost, disp_map, dmin, want_agg=False, params=None):
    """Guided-filter aggregation + WTA.  filter_cost/disp_map are updated IN PLACE exactly like the
    reference's in/out arguments; returns (mean_u8, agg or None).  guidedFilter.cu:4-295."""
    i, cost = _c(i, np.uint8), _c(cost, np.float32)
    h, w = i.shape
    size_d = cost.shape[0]
    if cost.shape != (size_d, h, w):
        raise ValueError("cost must be (size_d, h, w)")
    for a in (filter_cost, disp_map):
        if a.dtype != np.float32 or a.shape != (h, w) or not a.flags.c_contiguous:
            raise ValueError("filter_cost/disp_map must be C-contiguous float32 (h, w) arrays")
    mean = np.empty((h, w), np.uint8)
    agg = np.empty((size_d, h, w), np.float32) if want_agg else None
    _lib.check(_lib.lib().smx_compute_guided_filter(
        C.byref(_params(params)), _ptr(i), _ptr(cost), _ptr(filter_cost), _ptr(disp_map), _ptr(mean),
        _ptr(agg), w, h, size_d, dmin))
    return mean, agg


def detect_occlusion(disparity_left, disparity_right, d_occlusion, params=None):
    """LR consistency check; returns the updated copy of disparity_left.  occlusion.cu:17-85."""
    dl = _c(disparity_left, np.float32).copy()
    dr = _c(disparity_right, np.float32)
    h, w = dl.shape
    _lib.check(_lib.lib().smx_detect_occlusion(C.byref(_params(params)), _ptr(dl), _ptr(dr),
                                               int(d_occlusion), w, h))
    return dl


def fill_occlusion(disparity, v_min):
    """Scan-line filling; returns the filled copy.  occlusion.cu:111-132."""
    d = _c(disparity, np.float32).copy()
    h, w = d.shape
    _lib.check(_lib.lib().smx_fill_occlusion(_ptr(d), w, h, float(v_min)))
    return d


def stereo_pair(gray_l, gray_r, size_d, dminl=None, dminr=0, want_cost=False, want_agg=False,
                params=None):
    """main.cu:65-155 on two gray images, device-resident between the stages."""
    gl, gr = _c(gray_l, np.uint8), _c(gray_r, np.uint8)
    h, w = gl.shape
    if gr.shape != (h, w):
        raise ValueError("both views must have the same shape")
    if dminl is None:
        dminl = -(size_d - 1)
    vol = (size_d, h, w)
    r = {
        "bestl": np.empty((h, w), np.float32), "bestr": np.empty((h, w), np.float32),
        "dmapl": np.empty((h, w), np.float32), "dmapr": np.empty((h, w), np.float32),
        "meanl": np.empty((h, w), np.uint8), "meanr": np.empty((h, w), np.uint8),
        "occlusion": np.empty((h, w), np.float32), "filled": np.empty((h, w), np.float32),
        "costl": np.empty(vol, np.float32) if want_cost else None,
        "costr": np.empty(vol, np.float32) if want_cost else None,
        "aggl": np.empty(vol, np.float32) if want_agg else None,
        "aggr": np.empty(vol, np.float32) if want_agg else None,
    }
    out = _lib.PairOut()
    for f, k in (("best_l", "bestl"), ("best_r", "bestr"), ("dmap_l", "dmapl"), ("dmap_r", "dmapr"),
                 ("mean_l", "meanl"), ("mean_r", "meanr"), ("occlusion", "occlusion"),
                 ("filled", "filled"), ("cost_l", "costl"), ("cost_r", "costr"),
                 ("agg_l", "aggl"), ("agg_r", "aggr")):
        setattr(out, f, None if r[k] is None else r[k].ctypes.data)
    _lib.check(_lib.lib().smx_stereo_pair(C.byref(_params(params)), _ptr(gl), _ptr(gr), w, h, size_d,
                                          dminl, dminr, C.byref(out)))
    return r


def write_mat(mat):
    """Host-side float -> u8 normaliser of main.cu:13-35 (PNG writer input).  The reference's
    loop only lowers `min` on elements that did not raise `max` (`else if`, main.cu:22)."""
    m = np.ascontiguousarray(mat, dtype=np.float32).ravel()
    prev = np.concatenate(([np.float32(-150000000.0)], np.maximum.accumulate(m)[:-1]))
    prev = np.maximum(prev, np.float32(-150000000.0))
    record = m > prev
    mx = max(np.float32(-150000000.0), m.max())
    rest = m[~record]
    mn = np.float32(150000000.0)
    if rest.size:
        mn = min(mn, rest.min())
    c = ((m - np.float32(mn)) * np.float32(255.0) / np.float32(mx - mn)).astype(np.float32)
    return c.astype(np.int32).astype(np.uint8).reshape(np.shape(mat))
