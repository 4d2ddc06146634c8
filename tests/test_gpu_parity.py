"""Parity of the HIP path (through the C-ABI) against the CPU oracle and the reference's committed
Tsukuba outputs.  Bit-exact everywhere: labels, u8 images and every f32 array (the aggregated
volume included -- the 1e-4 relative tolerance of the north star is met with margin 0).

Run on the GPU box:  python -m pytest tests -m gpu -x -q
"""
import ctypes as C

import numpy as np
import pytest

import stereo_matching_cuda_amd as smx
from stereo_matching_cuda_amd import synth

pytestmark = pytest.mark.gpu


def _eq(a, b, name=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, (name, a.shape, b.shape, a.dtype, b.dtype)
    if a.dtype == np.float32:
        both_nan = np.isnan(a) & np.isnan(b)       # NaN payload / sign is not part of the contract
        a, b = a.view(np.uint32), b.view(np.uint32)
        a = np.where(both_nan, 0, a)
        b = np.where(both_nan, 0, b)
    bad = np.flatnonzero(a.ravel() != b.ravel())
    assert bad.size == 0, f"{name}: {bad.size} of {a.size} elements differ, first at {bad[:5]}"


# ---------------------------------------------------------------------------------------------
# stage by stage, host-pointer API (the reference's per-stage wrappers)
# ---------------------------------------------------------------------------------------------
def test_gray_matches_committed_images_and_oracle(golden, orc):
    for src, png in (("tsukuba0", "image_left"), ("tsukuba1", "image_right")):
        g = smx.rgb_to_grayscale(golden[src])
        _eq(g, golden[png], png)
    rng = np.random.default_rng(5)
    for shape in ((1, 1, 3), (7, 13, 3), (33, 65, 4)):
        rgb = rng.integers(0, 256, size=shape, dtype=np.uint8)
        _eq(smx.rgb_to_grayscale(rgb), orc.gray(rgb), f"gray{shape}")
    # every (r, g, b) on a coarse lattice incl. the sums that land on integers
    v = np.arange(0, 256, 5, dtype=np.uint8)
    rgb = np.stack(np.meshgrid(v, v, v, indexing="ij"), -1).reshape(1, -1, 3)
    _eq(smx.rgb_to_grayscale(rgb), orc.gray(rgb), "gray lattice")


@pytest.mark.parametrize("w,h,size_d,dmin", [
    (384, 288, 16, -15), (384, 288, 16, 0), (33, 7, 5, -4), (2, 1, 3, -1), (70, 3, 80, -79),
    (65, 9, 40, -10),
])
def test_cost_volume(tsukuba_gray, orc, w, h, size_d, dmin):
    if (w, h) == (384, 288):
        i1, i2 = tsukuba_gray if dmin < 0 else tsukuba_gray[::-1]
    else:
        rng = np.random.default_rng(w * 1000 + h)
        i1 = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
        i2 = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
    _eq(smx.compute_cost(i1, i2, size_d, dmin), orc.cost_volume(i1, i2, size_d, dmin), "cost")


@pytest.mark.parametrize("w,h", [(1, 1), (1, 70), (70, 1), (31, 63), (32, 64), (33, 65), (384, 288),
                                 (1242, 375), (100, 130)])
def test_integral(orc, w, h):
    rng = np.random.default_rng(w + 7 * h)
    img = (rng.random((h, w), dtype=np.float32) * 3).astype(np.float32)
    _eq(smx.integral(img), orc.integral(img), "integral")
    img = rng.normal(size=(h, w)).astype(np.float32)  # signed, cancellation
    _eq(smx.integral(img), orc.integral(img), "integral signed")


def test_integral_keeps_negative_zero(orc):
    img = np.zeros((5, 70), np.float32)
    img[0, 0] = -0.0
    img[2, :] = -0.0
    _eq(smx.integral(img), orc.integral(img), "integral -0")


def test_guided_filter_tsukuba(tsukuba_gray, tsukuba_oracle, golden, orc):
    Il, Ir = tsukuba_gray
    for side, I, dmin in (("l", Il, -15), ("r", Ir, 0)):
        best, dmap = smx.init_wta(*I.shape)
        mean, agg = smx.compute_guided_filter(I, tsukuba_oracle["cost" + side], best, dmap, dmin,
                                              want_agg=True)
        _eq(mean, tsukuba_oracle["mean" + side], "mean" + side)
        _eq(agg, tsukuba_oracle["agg" + side], "agg" + side)
        _eq(best, tsukuba_oracle["best" + side], "best" + side)
        _eq(dmap, tsukuba_oracle["dmap" + side], "dmap" + side)
        # straight against the reference's committed PNGs
        _eq(smx.write_mat(dmap), golden["disparity_map" + side], "disparity png")
        _eq(smx.write_mat(best), golden["best_cost" + side], "best png")
        _eq(mean, golden["image_mean_left" if side == "l" else "image_mean_right"], "mean png")


def test_guided_filter_inout_semantics(orc):
    """filter_cost/disp_map are in/out (guidedFilter.cu:403-411): presets lower than some q survive."""
    rng = np.random.default_rng(11)
    h, w, D = 37, 45, 6
    I = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
    cost = (rng.random((D, h, w), dtype=np.float32) * 2.5).astype(np.float32)
    _, _, _, agg = orc.guided_filter(I, cost, -5, want_agg=True)
    preset = np.median(agg.min(0)).astype(np.float32)
    best0 = np.full((h, w), preset, np.float32)
    best0[::2] = agg.min(0)[::2]  # exact ties with the preset on half the rows
    dmap0 = np.full((h, w), 77, np.float32)
    b1, d1 = best0.copy(), dmap0.copy()
    orc.guided_filter(I, cost, -5, best=b1, dmap=d1)
    b2, d2 = best0.copy(), dmap0.copy()
    smx.compute_guided_filter(I, cost, b2, d2, -5)
    _eq(b2, b1, "best inout")
    _eq(d2, d1, "dmap inout")
    assert (d1 == 77).any() and (d1 != 77).any()


def test_wta_ties_go_to_the_larger_slice(orc):
    """Duplicate slices give exact ties in q; `>=` keeps the later one (guidedFilter.cu:406)."""
    rng = np.random.default_rng(13)
    h, w = 21, 40
    I = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
    one = (rng.random((1, h, w), dtype=np.float32) * 2.5).astype(np.float32)
    cost = np.concatenate([one, one + 1, one, one + 2, one], 0)
    b1, d1 = orc.init_wta(h, w)
    orc.guided_filter(I, cost, 0, best=b1, dmap=d1)
    b2, d2 = smx.init_wta(h, w)
    smx.compute_guided_filter(I, cost, b2, d2, 0)
    _eq(d2, d1, "tie labels")
    assert (d1 == 4).all()


@pytest.mark.parametrize("w,h,D", [(20, 20, 3), (19, 40, 4), (64, 9, 2), (129, 70, 5)])
def test_guided_filter_small_and_ragged(orc, w, h, D):
    """Images smaller than / not a multiple of the 19x19 window, the 64-row bands and 32-col tiles."""
    rng = np.random.default_rng(w * h)
    I = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
    cost = (rng.random((D, h, w), dtype=np.float32) * 2.5).astype(np.float32)
    b1, d1, m1, a1 = orc.guided_filter(I, cost, -2, want_agg=True)
    b2, d2 = smx.init_wta(h, w)
    m2, a2 = smx.compute_guided_filter(I, cost, b2, d2, -2, want_agg=True)
    _eq(m2, m1, "mean")
    _eq(a2, a1, "agg")
    _eq(b2, b1, "best")
    _eq(d2, d1, "dmap")


def test_occlusion_tsukuba(tsukuba_oracle, golden):
    occ = smx.detect_occlusion(tsukuba_oracle["dmapl"], tsukuba_oracle["dmapr"], -115)
    _eq(occ, tsukuba_oracle["occlusion"], "occlusion")
    fil = smx.fill_occlusion(occ, -15.0)
    _eq(fil, tsukuba_oracle["filled"], "filled")
    _eq(smx.write_mat(occ), golden["occlu_mapl"], "occlu png")
    _eq(smx.write_mat(fil), golden["occlu_mapl_filled"], "filled png")


@pytest.mark.parametrize("w,h", [(1, 3), (63, 4), (64, 4), (65, 4), (200, 5), (1242, 3), (8193, 3), (16384, 3)])
def test_fill_occlusion_adversarial(orc, w, h):
    """(rows of more than 8192 pixels need more than 64 KB of LDS: the launch raises the kernel's limit)"""
    rng = np.random.default_rng(w)
    vmin = -20.0
    d = rng.integers(-20, 1, size=(h, w)).astype(np.float32)
    occl = rng.random((h, w)) < 0.6
    d[occl] = -120
    d[0, :] = -120                    # a fully occluded row
    if h > 1:
        d[1, : w // 2] = -120          # run touching the left border
    if h > 2:
        d[2, w // 2:] = -120           # run touching the right border
    _eq(smx.fill_occlusion(d, vmin), orc.fill_occlusion(d, vmin), "fill")
    # non-integer values: the reference tests (int)v >= vMin for "occluded" but v >= vMin for "valid"
    d2 = d + rng.random((h, w)).astype(np.float32) * np.float32(0.9)
    _eq(smx.fill_occlusion(d2, vmin), orc.fill_occlusion(d2, vmin), "fill non-integer")


def test_detect_occlusion_random(orc):
    rng = np.random.default_rng(17)
    h, w, D = 11, 90, 30
    dl = -rng.integers(0, D, size=(h, w)).astype(np.float32)
    dr = rng.integers(0, D, size=(h, w)).astype(np.float32)
    _eq(smx.detect_occlusion(dl, dr, -D - 99), orc.detect_occlusion(dl, dr, -D - 99), "detect")
    p = smx.default_params()
    p.d_lr = 2
    po = orc.Params.from_buffer_copy(bytes(p))
    _eq(smx.detect_occlusion(dl, dr, -200, params=p), orc.detect_occlusion(dl, dr, -200, params=po),
        "detect d_lr=2")


# ---------------------------------------------------------------------------------------------
# whole pair (main.cu:65-155), host-pointer entry and device-resident pipeline
@pytest.mark.parametrize("w,h,radius", [(384, 288, 9), (1, 1, 9), (40, 7, 9), (63, 65, 4), (20, 20, 0)])
def test_filter_dead_code_of_the_reference(tsukuba_gray, orc, w, h, radius):
    """filter() (filter.cu:117-207) is never called by main.cu; kept as a standalone op."""
    if (w, h) == (384, 288):
        I = tsukuba_gray[0]
    else:
        I = np.random.default_rng(w + h).integers(0, 256, size=(h, w), dtype=np.uint8)
    p = smx.default_params()
    p.radius = radius
    po = orc.Params.from_buffer_copy(bytes(p))
    mean, var = smx.filter(I, params=p)
    wm, wv = orc.filter(I, params=po)
    _eq(mean, wm, "filter mean")
    _eq(var, wv, "filter var")


# ---------------------------------------------------------------------------------------------
KEYS = ("meanl", "meanr", "bestl", "bestr", "dmapl", "dmapr", "occlusion", "filled")


def test_pair_tsukuba_against_committed_outputs(tsukuba_gray, tsukuba_oracle, golden):
    Il, Ir = tsukuba_gray
    r = smx.stereo_pair(Il, Ir, 16, dminl=-15, dminr=0, want_cost=True, want_agg=True)
    for k in KEYS + ("costl", "costr", "aggl", "aggr"):
        _eq(r[k], tsukuba_oracle[k], k)
    for k, png in (("bestl", "best_costl"), ("bestr", "best_costr"), ("dmapl", "disparity_mapl"),
                   ("dmapr", "disparity_mapr"), ("occlusion", "occlu_mapl"),
                   ("filled", "occlu_mapl_filled")):
        _eq(smx.write_mat(r[k]), golden[png], png)
    _eq(smx.write_mat(r["costl"][0]), golden["cost_lminus15"], "cost png l")
    _eq(smx.write_mat(r["costr"][0]), golden["cost_rminus15"], "cost png r")
    _eq(r["meanl"], golden["image_mean_left"], "mean png")


# what smx_last_agg_path() reports for a forced path: 2 = fused with the walker the call picks (comb walker 5 where
# it applies -- radius 9, costs built from the images, default-like cost parameters -- else the ring walker 2),
# 3 = ring walker forced, 5 = comb walker forced, 1 = multi-kernel, 4 = FAST
_RAN = {0: (1, 2, 5), 1: (1,), 2: (2, 5), 3: (2,), 4: (4,), 5: (5,)}


def _device_pair(Il, Ir, D, path=2, **kw):
    """path: 2 = fused single-kernel aggregation (the default product path), 1 = multi-kernel path,
    3 / 5 = fused with the ring / comb walker forced."""
    import torch
    from stereo_matching_cuda_amd.device import PairPipeline
    h, w = Il.shape
    dl = torch.from_numpy(Il).cuda()
    dr = torch.from_numpy(Ir).cuda()
    pipe = PairPipeline(w, h, D, multi_kernel=(path == 1), **kw)     # (the multi-kernel path needs five planes per slice)
    smx.lib().smx_set_agg_path(path)
    try:
        pipe.run(dl, dr)
        assert smx.lib().smx_last_agg_path() in _RAN[path]
    finally:
        smx.lib().smx_set_agg_path(0)
    return pipe.results()


@pytest.mark.parametrize("path", [2, 1, 3, 5])
def test_device_pipeline_tsukuba_fused_cost(tsukuba_gray, tsukuba_oracle, path):
    Il, Ir = tsukuba_gray
    r = _device_pair(Il, Ir, 16, path=path, dminl=-15, dminr=0, want_agg=True)
    for k in KEYS + ("aggl", "aggr"):
        _eq(r[k], tsukuba_oracle[k], k)


@pytest.mark.parametrize("radius,alpha,thc,thg,eps,dminl,dminr", [
    (9, 0.9, 7, 2, 6.5025, None, 0),       # reference defaults
    (3, 0.9, 7, 2, 6.5025, None, 0),       # smaller window: narrower tiles, other halo width
    (0, 0.5, 3, 1, 1.0, None, 0),          # degenerate 1x1 window; var + eps hits 0 -> NaN slices
    (5, 0.25, 20, 5, 0.01, -30, 7),        # other thresholds, ranges that do not start at 0
    (9, 0.9, 7, 2, 6.5025, -400, 380),     # every disparity points outside the image
])
def test_fused_path_parameter_variations(orc, radius, alpha, thc, thg, eps, dminl, dminr):
    """smx_params are runtime values (SystemIncludes.h:7-24 are macros in the reference)."""
    w, h, D = 210, 150, 24
    rng = np.random.default_rng(radius * 31 + thc)
    base = rng.integers(0, 256, size=(h, w + D), dtype=np.uint8)
    Il = np.ascontiguousarray(base[:, :w])
    Ir = np.ascontiguousarray(base[:, 9:9 + w])
    p = smx.default_params()
    p.radius, p.alpha, p.th_color, p.th_grad, p.eps = radius, alpha, thc, thg, eps
    po = orc.Params.from_buffer_copy(bytes(p))
    want = orc.stereo_pair(Il, Ir, D, dminl=dminl, dminr=dminr, want_agg=True, params=po)
    kw = dict(dminr=dminr, want_agg=True, params=p)
    if dminl is not None:
        kw["dminl"] = dminl
    r = _device_pair(Il, Ir, D, **kw)
    for k in KEYS + ("aggl", "aggr"):
        _eq(r[k], want[k], k)


@pytest.mark.parametrize("seed", range(12))
def test_fused_path_fuzz(orc, seed):
    """Seeded random shapes, disparity ranges and parameters through the fused path vs the oracle."""
    rng = np.random.default_rng(1000 + seed)
    w = int(rng.integers(2, 420))
    h = int(rng.integers(1, 260))
    D = int(rng.integers(1, 70))
    p = smx.default_params()
    p.radius = int(rng.integers(0, 10))
    p.alpha = float(rng.choice([0.9, 0.5, 0.1]))
    p.th_color = int(rng.integers(1, 30))
    p.th_grad = int(rng.integers(1, 8))
    p.eps = float(rng.choice([6.5025, 100.0, 0.5]))
    p.d_lr = int(rng.integers(0, 3))
    dminl = int(-rng.integers(0, 2 * D + 3))
    dminr = int(rng.integers(-3, D + 3))
    shift = int(rng.integers(0, max(1, min(D, w // 3))))
    base = rng.integers(0, 256, size=(h, w + D + 8), dtype=np.uint8)
    if seed % 3 == 0:                         # flat regions: zero costs, exact ties, tiny sums
        base = (base // 64 * 64).astype(np.uint8)
    Il = np.ascontiguousarray(base[:, :w])
    Ir = np.ascontiguousarray(base[:, shift:shift + w])
    po = orc.Params.from_buffer_copy(bytes(p))
    want = orc.stereo_pair(Il, Ir, D, dminl=dminl, dminr=dminr, want_agg=True, params=po)
    r = _device_pair(Il, Ir, D, dminl=dminl, dminr=dminr, want_agg=True, params=p)
    for k in KEYS + ("aggl", "aggr"):
        _eq(r[k], want[k], f"seed {seed} w={w} h={h} D={D} R={p.radius} {k}")


def test_radius_above_nine_uses_the_multi_kernel_path(orc):
    w, h, D = 90, 70, 5
    rng = np.random.default_rng(77)
    Il = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
    Ir = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
    p = smx.default_params()
    p.radius = 12
    po = orc.Params.from_buffer_copy(bytes(p))
    want = orc.stereo_pair(Il, Ir, D, params=po)
    r = _device_pair(Il, Ir, D, path=0, params=p)
    for k in KEYS:
        _eq(r[k], want[k], k)
    assert smx.lib().smx_last_agg_path() == 1


def test_default_path_is_the_fused_one(tsukuba_gray):
    import torch
    from stereo_matching_cuda_amd.device import PairPipeline
    Il, Ir = tsukuba_gray
    pipe = PairPipeline(384, 288, 16)
    pipe.run(torch.from_numpy(Il).cuda(), torch.from_numpy(Ir).cuda())
    assert smx.lib().smx_last_agg_path() == 5      # the comb walker (smx_agg_v5.hip) at the reference's parameters


def _geometry(radius):
    ow, bh, tw = C.c_int(), C.c_int(), C.c_int()
    smx.check(smx.lib().smx_agg_geometry(radius, C.byref(ow), C.byref(bh), C.byref(tw)))
    return ow.value, bh.value, tw.value


def _ragged_shapes(OW, BH, dense):
    """Shapes aimed at the tile boundaries of a fused walker, derived from its geometry constants
    (smx_agg_geometry): widths around multiples of the strip width OW (w % OW in {0, 1, OW-1}, and
    w + R crossing a strip count), heights around multiples of the band height BH (exact multi-band
    heights, one row more / less, heights whose lagged stage-2 / q rows need an extra band)."""
    ws = [2, OW - 1, OW, OW + 1, 2 * OW - 9, 2 * OW - 8, 2 * OW, 2 * OW + 1, 3 * OW - 1]
    hs = [1, BH - 1, BH, BH + 1, 2 * BH, 2 * BH + 1, 3 * BH - 18, 3 * BH - 9, 3 * BH, 4 * BH + 7]
    shapes = [(2, 1, 2), (20, 20, 3), (19, 40, 4), (300, 200, 70 if dense else 9)]
    for i, w in enumerate(ws):
        shapes.append((w, hs[i % len(hs)], 3 + i % 5))
    for i, h in enumerate(hs):
        shapes.append((ws[(i + 3) % len(ws)], h, 2 + i % 4))
    return sorted(set(shapes))


# literal copies of the library constants; test_ragged_shapes_match_the_library_geometry pins them
_RING = (64, 16)       # smx_agg_v4.hip: output columns per strip, rows per band
_COMB = (152, 10)      # smx_agg_v5.hip (comb length 9)


def test_ragged_shapes_match_the_library_geometry():
    assert _geometry(9) == (_COMB[0], _COMB[1], _COMB[0] + 19)
    assert _geometry(0) == (_RING[0], _RING[1], _RING[0] + 1)        # the comb walker serves radius 9 only
    smx.lib().smx_set_agg_path(3)
    try:
        assert _geometry(9) == (_RING[0], _RING[1], _RING[0] + 19)
    finally:
        smx.lib().smx_set_agg_path(0)


@pytest.mark.parametrize("path,w,h,D", [(3,) + s for s in _ragged_shapes(*_RING, True)] +
                         [(5,) + s for s in _ragged_shapes(*_COMB, False)])
def test_fused_path_small_and_ragged(orc, path, w, h, D):
    """Strip / band boundaries of both fused walkers (ring walker: 64 output columns per strip, 16-row bands,
    36-row rings; comb walker: 152 columns, 10-row bands, 20-slot register rings -- the shapes follow
    smx_agg_geometry), images smaller than one tile, disparity ranges wider than the image."""
    rng = np.random.default_rng(w * 7 + h * 3 + D)
    base = rng.integers(0, 256, size=(h, w + D), dtype=np.uint8)
    Il = np.ascontiguousarray(base[:, :w])
    Ir = np.ascontiguousarray(base[:, D // 2: D // 2 + w])
    want = orc.stereo_pair(Il, Ir, D, want_agg=True)
    r = _device_pair(Il, Ir, D, path=path, want_agg=True)
    for k in KEYS + ("aggl", "aggr"):
        _eq(r[k], want[k], k)
    if path == 5:
        # ... and the product's default layout: the comb-ordered q scratch (column order in the last strip), un-permuted by
        # its own WTA pass -- what runs when the caller does not ask for the aggregated volume
        r = _device_pair(Il, Ir, D, path=path, want_agg=False)
        for k in KEYS:
            _eq(r[k], want[k], "scratch layout " + k)


@pytest.mark.parametrize("seed", range(10))
def test_comb_walker_fuzz(orc, seed):
    """Seeded random shapes up to five strips wide through the comb walker (radius 9, reference parameters,
    random d_lr and disparity ranges incl. ones that point outside the image) vs the oracle; every third
    seed has flat regions: zero costs, exact ties, zero window sums."""
    rng = np.random.default_rng(5000 + seed)
    w = int(rng.integers(2, 1300))
    h = int(rng.integers(1, 90))
    D = int(rng.integers(1, 12))
    p = smx.default_params()
    p.d_lr = int(rng.integers(0, 3))
    dminl = int(-rng.integers(0, 2 * D + 3))
    dminr = int(rng.integers(-3, D + 3))
    shift = int(rng.integers(0, max(1, min(D, w // 3))))
    base = rng.integers(0, 256, size=(h, w + D + 8), dtype=np.uint8)
    if seed % 3 == 0:
        base = (base // 64 * 64).astype(np.uint8)
    if seed % 5 == 4:
        base[:] = 255                               # saturated: every cost is zero
    Il = np.ascontiguousarray(base[:, :w])
    Ir = np.ascontiguousarray(base[:, shift:shift + w])
    po = orc.Params.from_buffer_copy(bytes(p))
    want = orc.stereo_pair(Il, Ir, D, dminl=dminl, dminr=dminr, want_agg=True, params=po)
    r = _device_pair(Il, Ir, D, path=5, dminl=dminl, dminr=dminr, want_agg=True, params=p)
    for k in KEYS + ("aggl", "aggr"):
        _eq(r[k], want[k], f"seed {seed} w={w} h={h} D={D} {k}")
    r = _device_pair(Il, Ir, D, path=5, dminl=dminl, dminr=dminr, want_agg=False, params=p)     # the comb-ordered scratch + k_v5_wta
    for k in KEYS:
        _eq(r[k], want[k], f"seed {seed} w={w} h={h} D={D} scratch layout {k}")
    # the reference's calling convention: materialised cost volumes through the same kernel (guidedFilter.cu:198)
    import torch
    from stereo_matching_cuda_amd.device import PairPipeline
    wc = orc.stereo_pair(Il, Ir, D, dminl=dminl, dminr=dminr, want_cost=True, params=po)
    pipe = PairPipeline(w, h, D, dminl=dminl, dminr=dminr, params=p)
    pipe.aggregate(torch.from_numpy(Il).cuda(), torch.from_numpy(Ir).cuda(), torch.from_numpy(wc["costl"]).cuda(),
                   torch.from_numpy(wc["costr"]).cuda())
    assert smx.lib().smx_last_agg_path() == (5 if w * h >= 4 else 2)
    pipe.finish()
    r = pipe.results()
    for k in KEYS:
        _eq(r[k], want[k], f"seed {seed} w={w} h={h} D={D} cost volumes {k}")


@pytest.mark.parametrize("h", [1, 9, 12, 22, 23, 24, 32, 33, 41, 52, 60, 95, 130])
def test_items_pipelined_across_a_workgroups_tickets(orc, h):
    """More work items than workgroup slots (3 strips x 2 views x 100 slices = 600 > 512; then 5 strips x 2 x 150 = 1 500): some
    workgroups take a second and third ticket and start it while stage 2 still finishes the one before (smx_agg_v5.hip,
    period(h)).  Heights on both sides of where the period is rounded up (h = 12, 32, 52: the last a/b row falls into a band of
    its own), and images too short for their strip count that must NOT overlap (period < 2 K + 2, smx_agg_v5.h period(): two
    workgroups can wait for each other's last records -- this test caught that, through the bounded hand-off wait, on
    h = 9; and with a period of the whole item the completion flag must not wait for the NEXT item's first slots: h = 1, 5 strips).
    2 strips (300 wide) pipeline from h = 22 with the shortest period there is (6), 3 strips from h = 32, 5 strips from h = 95.
    Three runs each: the failures were a matter of timing."""
    for w, D in ((310, 100), (620, 150), (300, 140)):
        rng = np.random.default_rng(900 + h + w)
        base = rng.integers(0, 256, size=(h, w + D), dtype=np.uint8)
        base = (base // 8 * 8).astype(np.uint8)
        Il = np.ascontiguousarray(base[:, :w])
        Ir = np.ascontiguousarray(base[:, 7:7 + w])
        want = orc.stereo_pair(Il, Ir, D)
        for rep in range(3):
            r = _device_pair(Il, Ir, D, path=5)
            for k in KEYS:
                _eq(r[k], want[k], f"h={h} w={w} run {rep} {k}")


def test_items_pipelined_nine_strips(orc):
    """KITTI's strip count (9) at a height whose period (24 slots) is just above the deadlock-freedom bound 2 K + 2 = 20:
    720 items on 512 workgroup slots, three runs."""
    w, h, D = 1242, 200, 40
    Il, Ir = synth.gen_pair(w, h, D, 77)
    want = orc.stereo_pair(Il, Ir, D)
    for rep in range(3):
        r = _device_pair(Il, Ir, D, path=5)
        for k in KEYS:
            _eq(r[k], want[k], f"run {rep} {k}")


@pytest.mark.parametrize("radius,w,h", [(0, 128, 52), (0, 129, 53), (4, 64, 26), (4, 65, 78), (4, 192, 27), (4, 64, 16), (9, 130, 33), (1, 70, 48)])
def test_fused_path_small_radius_at_tile_edges(orc, radius, w, h):
    """Radii other than 9 change the tile width (OW + 2R + 1), the halo width and the row lags."""
    D = 5
    rng = np.random.default_rng(radius * 100 + w + h)
    base = rng.integers(0, 256, size=(h, w + D), dtype=np.uint8)
    Il = np.ascontiguousarray(base[:, :w])
    Ir = np.ascontiguousarray(base[:, 2:2 + w])
    p = smx.default_params()
    p.radius = radius
    po = orc.Params.from_buffer_copy(bytes(p))
    want = orc.stereo_pair(Il, Ir, D, want_agg=True, params=po)
    r = _device_pair(Il, Ir, D, want_agg=True, params=p)
    for k in KEYS + ("aggl", "aggr"):
        _eq(r[k], want[k], k)


def test_materialised_cost_volume_takes_the_fused_path(tsukuba_gray, tsukuba_oracle):
    """The reference's calling convention (compute_guided_filter with a cost volume from compute_cost,
    guidedFilter.cu:4-295) runs the same fused kernel, reading p instead of building it."""
    import torch
    from stereo_matching_cuda_amd.device import PairPipeline
    Il, Ir = tsukuba_gray
    dl, dr = torch.from_numpy(Il).cuda(), torch.from_numpy(Ir).cuda()
    cl = torch.from_numpy(tsukuba_oracle["costl"]).cuda()
    cr = torch.from_numpy(tsukuba_oracle["costr"]).cuda()
    for want_agg in (True, False):
        pipe = PairPipeline(384, 288, 16, dminl=-15, dminr=0, want_agg=want_agg)
        pipe.aggregate(dl, dr, cl, cr)
        assert smx.lib().smx_last_agg_path() == 5, "radius 9: the comb walker, loading p instead of building it"
        pipe.finish()
        r = pipe.results()
        assert not _fallback_ran(pipe), "costs from costVolume.cu:187 are +0 or normal: the queued ring walker must not run"
        for k in KEYS + (("aggl", "aggr") if want_agg else ()):
            _eq(r[k], tsukuba_oracle[k], k)
    smx.lib().smx_set_agg_path(3)
    try:
        pipe = PairPipeline(384, 288, 16, dminl=-15, dminr=0, want_agg=True)
        pipe.aggregate(dl, dr, cl, cr)
        assert smx.lib().smx_last_agg_path() == 2
        pipe.finish()
        r = pipe.results()
        for k in KEYS + ("aggl", "aggr"):
            _eq(r[k], tsukuba_oracle[k], "ring walker " + k)
    finally:
        smx.lib().smx_set_agg_path(0)


def _fallback_ran(pipe):
    import ctypes as C
    import torch
    torch.cuda.synchronize()
    f = C.c_int(-1)
    smx.check(smx.lib().smx_dev_agg_fallback(C.c_void_p(pipe.ws.data_ptr()), C.byref(f)))
    return bool(f.value)


@pytest.mark.parametrize("w,h,D", _ragged_shapes(*_COMB, False))
def test_cost_volume_convention_on_the_comb_walker(orc, w, h, D):
    """guidedFilter.cu:198-200 (`copyFromBigToLittleOnGPU`: the slice loop reads a materialised volume) on the comb walker at
    every strip / band boundary shape: quads that start at column -1 (strip 0), quads that run over the end of a row, of the
    plane, of the volume (clamped and shifted back), bands below the image; both q layouts; one and several slices per launch."""
    import torch
    from stereo_matching_cuda_amd.device import PairPipeline
    rng = np.random.default_rng(w * 11 + h * 5 + D)
    base = rng.integers(0, 256, size=(h, w + D), dtype=np.uint8)
    Il = np.ascontiguousarray(base[:, :w])
    Ir = np.ascontiguousarray(base[:, D // 2: D // 2 + w])
    want = orc.stereo_pair(Il, Ir, D, want_cost=True, want_agg=True)
    dl, dr = torch.from_numpy(Il).cuda(), torch.from_numpy(Ir).cuda()
    cl, cr = torch.from_numpy(want["costl"]).cuda(), torch.from_numpy(want["costr"]).cuda()
    for want_agg, sif in ((True, None), (False, None), (False, 1)):
        pipe = PairPipeline(w, h, D, want_agg=want_agg, slices_in_flight=sif)
        pipe.aggregate(dl, dr, cl, cr)
        # (a plane of fewer than four costs cannot hold a 16-byte quad: the ring walker takes those)
        assert smx.lib().smx_last_agg_path() == (5 if w * h >= 4 else 2)
        pipe.finish()
        r = pipe.results()
        assert not _fallback_ran(pipe)
        for k in KEYS + (("aggl", "aggr") if want_agg else ()):
            _eq(r[k], want[k], f"agg={want_agg} sif={sif} {k}")


@pytest.mark.parametrize("kind", ["negative", "minus_zero", "denormal", "tiny", "huge", "inf", "nan"])
def test_cost_values_outside_the_comb_walkers_argument_fall_back_on_the_device(orc, kind):
    """The comb walker's exactness argument needs every cost to be +0 or a normal number in [2^-60, 2^60].  A volume with ONE
    value outside that set (in an interior strip, at an image corner) must still give the oracle's result bit for bit: the
    cost wave raises the second status word and the queued ring walker redoes the chunk -- no host round trip."""
    import torch
    from stereo_matching_cuda_amd.device import PairPipeline
    rng = np.random.default_rng(77)
    w, h, D = 330, 47, 3
    I = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
    cost = (rng.random((D, h, w), dtype=np.float32) * 2.5).astype(np.float32)
    bad = {"negative": -0.75, "minus_zero": -0.0, "denormal": 1e-41, "tiny": 2.0 ** -70, "huge": 2.0 ** 70, "inf": np.inf,
           "nan": np.nan}[kind]
    for (z, y, x) in ((1, 20, 200), (D - 1, h - 1, w - 1), (0, 0, 0)):
        c = cost.copy()
        c[z, y, x] = np.float32(bad)
        b1, d1, m1, a1 = orc.guided_filter(I, c, -2, want_agg=True)
        b2, d2 = smx.init_wta(h, w)
        m2, a2 = smx.compute_guided_filter(I, c, b2, d2, -2, want_agg=True)
        assert smx.lib().smx_last_agg_path() == 5
        _eq(m2, m1, f"{kind} at {(z, y, x)} mean")
        _eq(a2, a1, f"{kind} at {(z, y, x)} agg")
        _eq(b2, b1, f"{kind} at {(z, y, x)} best")
        _eq(d2, d1, f"{kind} at {(z, y, x)} dmap")
    # ... and the device-pointer call reports that the fall-back ran (and does not for the clean volume)
    dI = torch.from_numpy(I).cuda()
    for vol, expect in ((c, True), (cost, False)):
        pipe = PairPipeline(w, h, D, dminl=-2, dminr=-2)
        pipe.init_keys()
        pipe.aggregate_view(0, dI, dI, torch.from_numpy(vol).cuda())
        assert _fallback_ran(pipe) == expect, (kind, expect)


def test_pair_step_is_capturable_in_a_hip_graph(tsukuba_gray, tsukuba_oracle):
    """include/smx.h: smx_dev_* do not allocate or synchronise -> one pair step can be captured
    into a hipGraph and replayed; the replay must give the oracle's result."""
    import torch
    from stereo_matching_cuda_amd.device import PairPipeline
    Il, Ir = tsukuba_gray
    dl, dr = torch.from_numpy(Il).cuda(), torch.from_numpy(Ir).cuda()
    pipe = PairPipeline(384, 288, 16, dminl=-15, dminr=0)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        pipe.run(dl, dr)                      # warm-up outside capture
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        pipe.run(dl, dr)
    pipe.keys.zero_()
    pipe.dmap.zero_()
    pipe.filled.zero_()
    for _ in range(2):
        g.replay()
    r = pipe.results()
    for k in KEYS:
        _eq(r[k], tsukuba_oracle[k], k)


def test_finish_pair_equals_the_per_call_sequence(tsukuba_gray, tsukuba_oracle):
    """smx_dev_finish_pair (one launch) against smx_dev_init_wta + 2 x smx_dev_apply_keys + copy +
    smx_dev_detect_occlusion + copy + smx_dev_fill_occlusion, and both against the oracle."""
    import torch
    from stereo_matching_cuda_amd.device import PairPipeline
    Il, Ir = tsukuba_gray
    dl, dr = torch.from_numpy(Il).cuda(), torch.from_numpy(Ir).cuda()
    pipe = PairPipeline(384, 288, 16, dminl=-15, dminr=0)
    pipe.aggregate(dl, dr)
    pipe.finish()
    fused = pipe.results()
    for t in (pipe.best, pipe.dmap, pipe.occlusion, pipe.filled):
        t.fill_(-7.0)
    pipe.finish_per_call()
    percall = pipe.results()
    for k in ("dmapl", "dmapr", "bestl", "bestr", "occlusion", "filled"):
        _eq(fused[k], percall[k], k)
        _eq(fused[k], tsukuba_oracle[k], k)


def test_aggregated_volume_out_at_a_4_byte_aligned_address(orc):
    """d_agg needs no more than float alignment: with an even plane size the WTA pass reads q with 8-byte loads
    only if the address allows it (include/smx.h)."""
    import torch
    from stereo_matching_cuda_amd.device import PairPipeline
    w, h, D = 150, 100, 4
    Il, Ir = synth.gen_pair(w, h, D, 5)
    want = orc.stereo_pair(Il, Ir, D, want_agg=True)
    pipe = PairPipeline(w, h, D, want_agg=True)
    buf = torch.empty(2 * D * h * w + 1, dtype=torch.float32, device="cuda")
    pipe.agg = buf[1:].view(2, D, h, w)
    assert pipe.agg.data_ptr() % 8 == 4
    pipe.run(torch.from_numpy(Il).cuda(), torch.from_numpy(Ir).cuda())
    got = pipe.results()
    for k in ("aggl", "aggr", "dmapl", "dmapr", "bestl", "bestr", "filled"):
        _eq(got[k], want[k], k)


@pytest.mark.parametrize("w,h,D", [(1242, 5, 9), (300, 7, 3), (8192, 3, 5), (9000, 3, 5)])
def test_finish_pair_on_random_keys(orc, w, h, D):
    """smx_dev_finish_pair on keys that no aggregation produced (random costs and slices, some pixels without any
    candidate): the one-launch form (rows <= 8192 pixels) and the three-kernel form behind it, against the oracle's
    presets / dispSelect rule / LR check / filling."""
    import torch
    from stereo_matching_cuda_amd.device import PairPipeline
    rng = np.random.default_rng(w + h)
    dminl, dminr = -(D - 1), 0
    cost = (rng.random((2, h, w), dtype=np.float32) * 3).astype(np.float32)
    cost[rng.random((2, h, w)) < 0.05] = np.nan           # -> identity keys: the presets survive
    slices = rng.integers(0, D, size=(2, h, w))
    keys = orc.pack_keys(cost, slices)
    pipe = PairPipeline(w, h, D, dminl=dminl, dminr=dminr)
    pipe.keys.copy_(torch.from_numpy(keys.reshape(2, h, w)))
    pipe.finish()
    torch.cuda.synchronize()
    got = {k: getattr(pipe, k).cpu().numpy() for k in ("best", "dmap", "occlusion", "filled")}
    has = ~np.isnan(cost)
    best = np.where(has, cost, np.float32(np.frombuffer(b"\x7f\x7f\x7f\x7f", np.float32)[0])).astype(np.float32)
    dmin = np.array([dminl, dminr]).reshape(2, 1, 1)
    dmap = np.where(has, (dmin + slices).astype(np.float32), np.float32(0)).astype(np.float32)
    _eq(got["best"], best, "best")
    _eq(got["dmap"], dmap, "dmap")
    occ = orc.detect_occlusion(dmap[0], dmap[1], dminl - 100)
    _eq(got["occlusion"], occ, "occlusion")
    _eq(got["filled"], orc.fill_occlusion(occ, float(dminl)), "filled")
    # and the per-call sequence on the same keys
    for t in (pipe.best, pipe.dmap, pipe.occlusion, pipe.filled):
        t.fill_(-7.0)
    pipe.finish_per_call()
    torch.cuda.synchronize()
    for k in got:
        _eq(getattr(pipe, k).cpu().numpy(), got[k], "per-call " + k)


def test_two_pipelines_on_two_devices(tsukuba_gray, tsukuba_oracle):
    """Each PairPipeline launches on its OWN device's current stream whatever device is current in the
    calling thread (device.py::_on_device); two shards on two GPUs merged by hand must give the oracle's
    result.  Needs two GPUs: skipped on the one-GPU test box."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from stereo_matching_cuda_amd.device import PairPipeline
    Il, Ir = tsukuba_gray
    pipes = []
    for g, dev in enumerate(("cuda:0", "cuda:1")):
        pipe = PairPipeline(384, 288, 16, dminl=-15, dminr=0, s_begin=8 * g, s_end=8 * (g + 1), device=dev)
        dl, dr = torch.from_numpy(Il).to(dev), torch.from_numpy(Ir).to(dev)
        pipe.aggregate(dl, dr)                 # current device stays cuda:0 for both
        pipes.append(pipe)
    for pipe in pipes:
        torch.cuda.synchronize(pipe.device)
        pipe.check_status()
    merged = torch.minimum(pipes[0].keys, pipes[1].keys.to("cuda:0"))
    pipes[0].keys.copy_(merged)
    pipes[0].finish()
    torch.cuda.synchronize()
    r = pipes[0].results()
    for k in ("dmapl", "dmapr", "bestl", "bestr", "occlusion", "filled"):
        _eq(r[k], tsukuba_oracle[k], k)


@pytest.mark.parametrize("path", [2, 1])
def test_device_pipeline_chunked_equals_unchunked(tsukuba_gray, tsukuba_oracle, path):
    Il, Ir = tsukuba_gray
    r = _device_pair(Il, Ir, 16, path=path, dminl=-15, dminr=0, slices_in_flight=3)
    for k in KEYS:
        _eq(r[k], tsukuba_oracle[k], k)
    if path == 2:
        import ctypes as C
        c, n = C.c_int(), C.c_int()
        smx.check(smx.lib().smx_last_agg_chunk(C.byref(c), C.byref(n)))
        assert (c.value, n.value) == (3, 6), "16 slices, three per launch: six walker launches"


def test_workspace_too_small_is_an_error(tsukuba_gray):
    import torch
    from stereo_matching_cuda_amd.device import PairPipeline
    Il, Ir = tsukuba_gray
    pipe = PairPipeline(384, 288, 16)
    pipe.ws_bytes = 1024
    with pytest.raises(smx.SmxError) as e:
        pipe.aggregate(torch.from_numpy(Il).cuda(), torch.from_numpy(Ir).cuda())
    assert e.value.code == -3


def test_virtual_shards_merge_to_the_unsharded_result(tsukuba_gray, tsukuba_oracle):
    """G virtual shards on one device + u64 min merge == unsharded, bit for bit."""
    import torch
    from stereo_matching_cuda_amd.device import PairPipeline
    from stereo_matching_cuda_amd.sharded import shard_range
    Il, Ir = tsukuba_gray
    dl, dr = torch.from_numpy(Il).cuda(), torch.from_numpy(Ir).cuda()
    for G in (2, 3, 5, 16, 20):  # 20 > D: some shards are empty
        merged = None
        for g in range(G):
            s0, s1 = shard_range(16, g, G)
            pipe = PairPipeline(384, 288, 16, dminl=-15, dminr=0, s_begin=s0, s_end=s1)
            pipe.aggregate(dl, dr)
            k = pipe.keys.clone()
            merged = k if merged is None else torch.minimum(merged, k)
        pipe.keys.copy_(merged)
        pipe.finish()
        r = pipe.results()
        for k in ("bestl", "bestr", "dmapl", "dmapr", "occlusion", "filled"):
            _eq(r[k], tsukuba_oracle[k], f"G={G} {k}")


def test_pair_kitti_shape_synthetic(orc):
    """BASELINE config 4 shape (1242x375, D=192), seeded synthetic pair, vs the oracle."""
    w, h, D = synth.SHAPES["kitti"]
    Il, Ir = synth.gen_pair(w, h, D, synth.SEEDS["kitti"])
    want = orc.stereo_pair(Il, Ir, D, want_agg=True)
    r = _device_pair(Il, Ir, D, want_agg=True)
    for k in KEYS + ("aggl", "aggr"):      # the aggregated volumes too: 2 x 89.4 M cells, bit for bit
        _eq(r[k], want[k], k)


def test_pair_motorcycle_shape_properties():
    """BASELINE config 3 shape (2964x2000, D=280) is too slow for the oracle: check size-independent
    properties instead: 2 virtual shards == unsharded; chunked == unchunked; filling is idempotent;
    every label lies in its range; a key-merge with itself is the identity."""
    import torch
    from stereo_matching_cuda_amd.device import PairPipeline
    from stereo_matching_cuda_amd.sharded import shard_range
    w, h, D = synth.SHAPES["motorcycle"]
    Il, Ir = synth.gen_pair(w, h, D, synth.SEEDS["motorcycle"])
    dl, dr = torch.from_numpy(Il).cuda(), torch.from_numpy(Ir).cuda()
    full = PairPipeline(w, h, D, slices_in_flight=70)
    full.run(dl, dr)
    assert full.last_chunk() == (70, 4), "280 slices, 70 per launch: the running WTA crosses four walker launches"
    ref = full.results()
    keys_full = full.keys.clone()
    assert ref["dmapl"].min() >= -(D - 1) and ref["dmapl"].max() <= 0
    assert ref["dmapr"].min() >= 0 and ref["dmapr"].max() <= D - 1
    again = smx.fill_occlusion(ref["filled"], float(-(D - 1)))
    _eq(again, ref["filled"], "fill idempotent")
    del full
    torch.cuda.empty_cache()
    merged = None
    for g in range(2):
        s0, s1 = shard_range(D, g, 2)
        pipe = PairPipeline(w, h, D, s_begin=s0, s_end=s1, slices_in_flight=35)
        pipe.aggregate(dl, dr)
        assert pipe.last_chunk() == (35, 4)
        k = pipe.keys.clone()
        merged = k if merged is None else torch.minimum(merged, k)
    pipe.keys.copy_(merged)
    assert torch.equal(pipe.keys, keys_full)
    pipe.finish()
    r = pipe.results()
    for k in KEYS[2:]:
        _eq(r[k], ref[k], k)


def test_kitti_shape_virtual_shards_equal_the_oracle(orc):
    """BASELINE config 4 (1242x375 D=192 disparity-sharded over 2/4/8 GPUs) on one device: G virtual
    shards merged with the u64 min == the oracle's unsharded result, bit for bit."""
    import torch
    from stereo_matching_cuda_amd.device import PairPipeline
    from stereo_matching_cuda_amd.sharded import shard_range
    w, h, D = synth.SHAPES["kitti"]
    Il, Ir = synth.gen_pair(w, h, D, synth.SEEDS["kitti"])
    want = orc.stereo_pair(Il, Ir, D)
    dl, dr = torch.from_numpy(Il).cuda(), torch.from_numpy(Ir).cuda()
    for G in (2, 4, 8):
        merged = None
        for g in range(G):
            s0, s1 = shard_range(D, g, G)
            pipe = PairPipeline(w, h, D, s_begin=s0, s_end=s1)
            pipe.aggregate(dl, dr)
            k = pipe.keys.clone()
            merged = k if merged is None else torch.minimum(merged, k)
        pipe.keys.copy_(merged)
        pipe.finish()
        r = pipe.results()
        for k in KEYS:
            _eq(r[k], want[k], f"G={G} {k}")


def test_pair_4k_shape_properties():
    """BASELINE config 5 shape (3840x2160, D=512; 8 GPUs there, one GPU with a chunked workspace here)
    through size-independent properties: 8 virtual shards == unsharded keys; a differently chunked
    run == the same keys; labels inside their ranges; filling idempotent."""
    import torch
    from stereo_matching_cuda_amd.device import PairPipeline
    from stereo_matching_cuda_amd.sharded import shard_range
    w, h, D = synth.SHAPES["4k"]
    Il, Ir = synth.gen_pair(w, h, D, synth.SEEDS["4k"])
    dl, dr = torch.from_numpy(Il).cuda(), torch.from_numpy(Ir).cuda()
    full = PairPipeline(w, h, D, slices_in_flight=128)
    full.run(dl, dr)
    assert full.last_chunk() == (128, 4), "512 slices, 128 per launch"
    ref = full.results()
    keys_full = full.keys.clone()
    assert ref["dmapl"].min() >= -(D - 1) and ref["dmapl"].max() <= 0
    assert ref["dmapr"].min() >= 0 and ref["dmapr"].max() <= D - 1
    again = smx.fill_occlusion(ref["filled"], float(-(D - 1)))
    _eq(again, ref["filled"], "fill idempotent")
    del full
    torch.cuda.empty_cache()
    merged = None
    for g in range(8):
        s0, s1 = shard_range(D, g, 8)
        pipe = PairPipeline(w, h, D, s_begin=s0, s_end=s1, slices_in_flight=24)   # 64 slices, 3 chunks
        pipe.aggregate(dl, dr)
        pipe.check_status()
        assert pipe.last_chunk() == (24, 3), "a 64-slice shard in launches of 24 + 24 + 16"
        k = pipe.keys.clone()
        merged = k if merged is None else torch.minimum(merged, k)
        if g < 7:
            del pipe
            torch.cuda.empty_cache()
    pipe.keys.copy_(merged)
    assert torch.equal(pipe.keys, keys_full)
    pipe.finish()
    r = pipe.results()
    for k in KEYS[2:]:
        _eq(r[k], ref[k], k)


# ---------------------------------------------------------------------------------------------
# N > 1 end to end on one GPU: two processes, each aggregating its disparity shard with the HIP
# kernels on cuda:0, merged by the product's all-reduce (gloo here; RCCL on a multi-GPU node).
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape,ranges,sif", [
    ("motorcycle", [(0, 3), (137, 140), (100, 103), (277, 280)], 2),
    ("4k", [(0, 3), (255, 258), (509, 512)], 2),
])
def test_full_geometry_slices_equal_the_oracle(orc, shape, ranges, sif):
    """BASELINE configs 3 and 5 pinned to the ORACLE at full geometry (2964x2000 D=280, 3840x2160 D=512) on sampled
    slice ranges -- first, last, inside -- each of three or more slices run with at most two slices per walker launch, so that
    every range straddles a launch boundary (asserted through smx_last_agg_chunk): aggregated
    planes and the packed keys of the range, bit for bit, through both q layouts of the comb walker (the caller's
    [z][y][x] volume / the comb-ordered scratch + its WTA pass).  The reference overflows its 32-bit sizes at these
    shapes (guidedFilter.cu:6-7,26, costVolume.cu:6,178); a 32-bit offset, band / strip count or chunking defect
    here would show."""
    import torch
    from stereo_matching_cuda_amd.device import PairPipeline
    w, h, D = synth.SHAPES[shape]
    Il, Ir = synth.gen_pair(w, h, D, synth.SEEDS[shape])
    dl, dr = torch.from_numpy(Il).cuda(), torch.from_numpy(Ir).cuda()
    dminl, dminr = -(D - 1), 0
    for s0, s1 in ranges:
        aggl, keysl = orc.guided_filter_slices(Il, Ir, dminl, s0, s1)
        aggr, keysr = orc.guided_filter_slices(Ir, Il, dminr, s0, s1)
        for want_agg in (True, False):
            pipe = PairPipeline(w, h, D, s_begin=s0, s_end=s1, slices_in_flight=sif, want_agg=want_agg)
            pipe.aggregate(dl, dr)
            pipe.check_status()
            # the bound on the slices per launch is what chunks (smx_set_max_slices_per_launch), whatever the workspace holds:
            # every range here crosses a launch boundary (re-zeroed tickets and flags, reused q scratch and records, the WTA
            # minimum accumulated over launches, slice * q_plane offsets of the second launch)
            assert pipe.last_chunk() == (sif, -(-(s1 - s0) // sif)) and pipe.last_chunk()[1] >= 2
            keys = pipe.keys.cpu().numpy()
            _eq(keys[0], keysl, f"{shape} [{s0},{s1}) agg={want_agg} keys l")
            _eq(keys[1], keysr, f"{shape} [{s0},{s1}) agg={want_agg} keys r")
            if want_agg:
                _eq(pipe.agg[0].cpu().numpy(), aggl, f"{shape} [{s0},{s1}) agg l")
                _eq(pipe.agg[1].cpu().numpy(), aggr, f"{shape} [{s0},{s1}) agg r")
            del pipe
        torch.cuda.empty_cache()


def _rank_worker(rank, world, port, Il, Ir, D, out_dir):
    import os
    import torch
    import torch.distributed as dist
    from stereo_matching_cuda_amd.sharded import ShardedPair
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        h, w = Il.shape
        sp = ShardedPair(w, h, D, rank=rank, world=world, device="cuda:0")
        pipe = sp.run(torch.from_numpy(Il).cuda(), torch.from_numpy(Ir).cuda())
        r = pipe.results()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **{k: r[k] for k in KEYS[2:]})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_pair_two_processes_on_one_gpu(orc, tmp_path, world):
    import socket
    import torch.multiprocessing as mp
    w, h, D = 200, 130, 37
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, size=(h, w + D), dtype=np.uint8)
    Il = np.ascontiguousarray(base[:, :w])
    Ir = np.ascontiguousarray(base[:, 11:11 + w])
    want = orc.stereo_pair(Il, Ir, D)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_rank_worker, args=(world, port, Il, Ir, D, str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        got = np.load(tmp_path / f"rank{rank}.npz")
        for k in KEYS[2:]:
            _eq(got[k], want[k], f"rank{rank} {k}")


@pytest.mark.gpu
def test_persistent_context_reuses_its_buffers(orc):
    """smx_create / smx_ctx_stereo_pair / smx_destroy (the SURVEY 8b context: device buffers, workspace and
    stream created once): three different pairs through ONE context, each equal to the oracle bit for bit,
    and equal to the one-shot smx_stereo_pair."""
    L = smx.lib()
    w, h, D = 150, 100, 7
    params = smx.default_params()
    ctx = C.c_void_p()
    smx.check(L.smx_create(C.byref(params), w, h, D, C.byref(ctx)))
    try:
        for seed in (11, 12, 13):
            Il, Ir = synth.gen_pair(w, h, D, seed)
            want = orc.stereo_pair(Il, Ir, D)
            n = w * h
            bufs = {k: np.empty(n, np.float32) for k in ("best_l", "best_r", "dmap_l", "dmap_r", "occlusion", "filled")}
            means = {k: np.empty(n, np.uint8) for k in ("mean_l", "mean_r")}
            from stereo_matching_cuda_amd._lib import PairOut
            out = PairOut()
            for k, a in {**bufs, **means}.items():
                setattr(out, k, a.ctypes.data)
            smx.check(L.smx_ctx_stereo_pair(ctx, Il.ctypes.data, Ir.ctypes.data, -(D - 1), 0, C.byref(out)))
            for got, key in ((bufs["dmap_l"], "dmapl"), (bufs["dmap_r"], "dmapr"), (bufs["best_l"], "bestl"),
                             (bufs["best_r"], "bestr"), (bufs["occlusion"], "occlusion"), (bufs["filled"], "filled"),
                             (means["mean_l"], "meanl"), (means["mean_r"], "meanr")):
                ref = np.asarray(want[key]).reshape(-1)
                if got.dtype == np.float32:
                    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (seed, key)
                else:
                    assert np.array_equal(got, ref), (seed, key)
    finally:
        smx.check(L.smx_destroy(ctx))


@pytest.mark.gpu
def test_pipelined_context_equals_the_synchronous_entry(orc):
    """smx_ctx_stereo_pair_async / smx_ctx_wait (pinned staging, upload of pair k+1 and download of pair k-1 under the
    aggregation of pair k): five different pairs pipelined two deep through one context; every result -- read through the
    staged pointers AND through the copy-out -- equals the synchronous smx_ctx_stereo_pair and the oracle bit for bit.
    Also the rules of the entry: a third pair in flight and a wait without a pair are SMX_E_ARG, the synchronous entry
    refuses to run while pairs are in flight."""
    from stereo_matching_cuda_amd._lib import PairOut
    L = smx.lib()
    w, h, D = 210, 100, 9
    n = w * h
    params = smx.default_params()
    ctx = C.c_void_p()
    smx.check(L.smx_create(C.byref(params), w, h, D, C.byref(ctx)))
    fkeys = ("best_l", "best_r", "dmap_l", "dmap_r", "occlusion", "filled")
    mkeys = ("mean_l", "mean_r")
    okey = {"best_l": "bestl", "best_r": "bestr", "dmap_l": "dmapl", "dmap_r": "dmapr", "occlusion": "occlusion",
            "filled": "filled", "mean_l": "meanl", "mean_r": "meanr"}

    def outbufs():
        bufs = {k: np.empty(n, np.float32) for k in fkeys}
        bufs.update({k: np.empty(n, np.uint8) for k in mkeys})
        out = PairOut()
        for k, a in bufs.items():
            setattr(out, k, a.ctypes.data)
        return bufs, out

    try:
        assert L.smx_ctx_wait(ctx, None, None) == -1
        pairs = [synth.gen_pair(w, h, D, seed) for seed in (21, 22, 23, 24, 25)]
        sync = []
        for Il, Ir in pairs:
            bufs, out = outbufs()
            smx.check(L.smx_ctx_stereo_pair(ctx, Il.ctypes.data, Ir.ctypes.data, -(D - 1), 0, C.byref(out)))
            sync.append(bufs)
        got = []

        def take():
            bufs, out = outbufs()
            staged = PairOut()
            smx.check(L.smx_ctx_wait(ctx, C.byref(staged), C.byref(out)))
            st = {}
            for k in fkeys:
                st[k] = np.ctypeslib.as_array(C.cast(getattr(staged, k), C.POINTER(C.c_float)), (n,)).copy()
            for k in mkeys:
                st[k] = np.ctypeslib.as_array(C.cast(getattr(staged, k), C.POINTER(C.c_uint8)), (n,)).copy()
            got.append((bufs, st))

        for i, (Il, Ir) in enumerate(pairs):
            Il2, Ir2 = Il.copy(), Ir.copy()
            smx.check(L.smx_ctx_stereo_pair_async(ctx, Il2.ctypes.data, Ir2.ctypes.data, -(D - 1), 0))
            Il2[:] = 0; Ir2[:] = 0            # the caller's buffers are free once the call has returned
            if i == 1:
                # two in flight: a third is refused, and so is the synchronous entry
                assert L.smx_ctx_stereo_pair_async(ctx, Il.ctypes.data, Ir.ctypes.data, -(D - 1), 0) == -1
                _, out = outbufs()
                assert L.smx_ctx_stereo_pair(ctx, Il.ctypes.data, Ir.ctypes.data, -(D - 1), 0, C.byref(out)) == -1
            if i >= 1:
                take()
        take()
        assert L.smx_ctx_wait(ctx, None, None) == -1
        assert len(got) == len(pairs)
        for i, (Il, Ir) in enumerate(pairs):
            want = orc.stereo_pair(Il, Ir, D)
            for k in fkeys + mkeys:
                ref = np.asarray(want[okey[k]]).reshape(-1)
                for name, a in (("sync", sync[i][k]), ("copied", got[i][0][k]), ("staged", got[i][1][k])):
                    if a.dtype == np.float32:
                        assert np.array_equal(a.view(np.uint32), ref.view(np.uint32)), (i, k, name)
                    else:
                        assert np.array_equal(a, ref), (i, k, name)
    finally:
        smx.check(L.smx_destroy(ctx))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["tsukuba", "synthetic"])
def test_fast_mode_is_close_but_not_bit_exact(tsukuba_gray, tsukuba_oracle, orc, shape):
    """the FAST aggregation (smx_set_agg_path(4)) is reported separately and never the default.
    On the comb walker it keeps every sum and its order and only replaces the two correctly rounded divisions of a cell by
    multiplications with the rounded reciprocal of the window area (and drops the exactness vote): a window mean moves by
    an ulp.  Even that is NOT within the 1e-4 `north_star` hoped for: a_k = (mean_Ip - mean_I * mean_p) / (var + eps)
    subtracts two numbers that agree in their first three or four digits, so one ulp in a mean is ~1e-4 .. 1e-3 of the
    covariance and shows up as 2.4e-4 relative in the aggregated volume (measured and printed; bound asserted: 1e-3).  No
    re-ordered or approximately divided f32 implementation can promise 1e-4 on this filter; the exact mode is 2 % slower
    than this one.  Labels flip only where the two best costs of a pixel are that close (count printed)."""
    if shape == "tsukuba":
        Il, Ir = tsukuba_gray
        D, want, kw = 16, tsukuba_oracle, dict(dminl=-15, dminr=0)
    else:
        D = 40
        Il, Ir = synth.gen_pair(330, 190, D, 77)
        want, kw = orc.stereo_pair(Il, Ir, D, want_agg=True), {}
    r = _device_pair(Il, Ir, D, path=4, want_agg=True, **kw)
    flips = {}
    for v in "lr":
        got, ref = np.asarray(r["agg" + v], np.float64), np.asarray(want["agg" + v], np.float64).reshape(r["agg" + v].shape)
        rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3)
        assert rel.max() <= 1e-3, (v, rel.max())
        worst = max(locals().get("worst", 0.0), float(rel.max()))
        flips[v] = int((np.asarray(r["dmap" + v]) != np.asarray(want["dmap" + v]).reshape(r["dmap" + v].shape)).sum())
        assert np.array_equal(r["mean" + v], np.asarray(want["mean" + v]).reshape(r["mean" + v].shape))
    n = Il.size
    print(f"FAST mode on {shape}: largest relative deviation of the aggregated volume {worst:.2e}; "
          f"label flips left {flips['l']} / right {flips['r']} of {n} pixels")
    assert max(flips.values()) <= n // 500      # far fewer in practice
