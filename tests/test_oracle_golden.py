"""Pins the CPU oracle (oracle/smx_oracle.c) to the reference.

1. The 12 output images committed by the reference's authors for the Tsukuba pair
   (reference stereo_matching_cuda/data/*.png -> tests/golden/tsukuba/): the oracle's outputs,
   pushed through the reference's own float->u8 normaliser (write_mat, main.cu:13-35), must equal
   every one of them pixel for pixel.
2. The sha256 manifest of the raw f32/u8 arrays recorded in SURVEY.md Appendix C.  Those dumps came from a
   host build of the reference's kernels behind stand-in CUDA headers, which the build rules do not accept
   as a reference run: the manifest is a regression anchor for the oracle (it agrees), it PINS NOTHING.
   What pins the oracle is (1): exact labels / u8 images, and best cost / first cost slice at the 8-bit
   quantisation of the committed PNGs; the unquantised floats are pinned only indirectly (the labels are exact
   although the smallest best / second-best margin on Tsukuba is 1.9e-6, and a one-ulp change of the
   aggregation order flips labels).
"""
import hashlib
import os

import numpy as np
import pytest

SURVEY_SHA256 = {
    "I_l": "841c87c51e1a57cfb75202b90b44f230879bd84b93a1ad541afc8c673a0e3215",
    "I_r": "f5118ecebd5fbd5ae341821ddaf1bded6276edd8a8880287f3e59e955decb902",
    "bestl": "9789ca055bd1950eca4cc0ca95c9b45e127a12831da71ded3da8aee56a2ec612",
    "bestr": "0f1b9d7baf5ada179883e96c2876e157a621f6998a46f221967abbf0b625ef5c",
    "costl": "25cd25f656511f9842745246339423fe2783e707b605856345431358bdb70ece",
    "costr": "7cafeefa87286cf0dabe796d26288860435abdda5b36cc65e1d3b5e6f94e9f5c",
    "dmapl": "ccd042dc5fb04f9d42d81dba24d874ae682bdcb6f21cb96d3731448c3a1e1c8a",
    "dmapr": "86bc0fd3de3f560d4754d321eeb03e978fe954a104c201dc52a34eae08d445bb",
    "meanl": "4018f13b365921866e90004d4907fddf1e199703b5b44a6ad5d152427214dbf3",
    "meanr": "6198ec4c1402aaaf8f0bb95e8ae7b746d152050379dd17d66b1ccb4e597567bb",
    "occlusion": "4f98010110b78ad2fb0968ec259a90c6497d521629edcc2ad82e43e22940453a",
    "filled": "e02097c358836843e453aaa366b011ef904900705f90ac6c4cbb1f12353f48fc",
    "aggl": "2b696bee75dc560c77e5418f1e25de1ed419c5501da11ed437b438a133debfa3",
    "aggr": "1189ae08672a748f7cf1bdbfb6af53f7cb6f2df3bcbbd6adc8ea4741788c959e",
}

PNG_SHA256 = {
    "best_costl": "bd05afa6ce2b076f875c887b0e625034288ba940a4632d9bfd06515d660064ca",
    "disparity_mapl": "fbf16d7496510e30e39da345b65c9ba2c129772d007cafdd78813dde654da693",
    "disparity_mapr": "4be919a0946c5ccaa1f31b189fd391952c5f3be6bc090e29cac313701b8cd3ed",
    "occlu_mapl": "0651af786845508eec5fdac0e2aba31df1f5b4a7319f66ab734b89ebad5c88a5",
    "occlu_mapl_filled": "e4aaf235f7927c2cd41b488560c8566bcc82c8cbc544c6e4a67ff84db3b585a4",
    "tsukuba0": "282056144ff4cdd5820db5f4a1be17c2d6807bff7810782fd953fd95df3dfb7b",
    "tsukuba1": "de433954a354e4229c638e8eb5f91d787f6aba151a167d732b9ec70089c81248",
}


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_fixture_files_are_the_reference_files():
    root = os.path.join(os.path.dirname(__file__), "golden", "tsukuba")
    for name, want in PNG_SHA256.items():
        with open(os.path.join(root, name + ".png"), "rb") as f:
            assert hashlib.sha256(f.read()).hexdigest() == want, name


def test_npz_matches_pngs(golden):
    PIL = pytest.importorskip("PIL.Image")
    root = os.path.join(os.path.dirname(__file__), "golden", "tsukuba")
    for name, arr in golden.items():
        assert np.array_equal(np.asarray(PIL.open(os.path.join(root, name + ".png"))), arr), name


def test_gray_matches_committed_images(golden, tsukuba_gray):
    Il, Ir = tsukuba_gray
    assert np.array_equal(Il, golden["image_left"])
    assert np.array_equal(Ir, golden["image_right"])


def test_mean_images(golden, tsukuba_oracle):
    assert np.array_equal(tsukuba_oracle["meanl"], golden["image_mean_left"])
    assert np.array_equal(tsukuba_oracle["meanr"], golden["image_mean_right"])


@pytest.mark.parametrize("key,png", [
    ("bestl", "best_costl"), ("bestr", "best_costr"),
    ("dmapl", "disparity_mapl"), ("dmapr", "disparity_mapr"),
    ("occlusion", "occlu_mapl"), ("filled", "occlu_mapl_filled"),
])
def test_maps_match_committed_pngs(golden, tsukuba_oracle, orc, key, png):
    assert np.array_equal(orc.write_mat_u8(tsukuba_oracle[key]), golden[png])


def test_first_cost_slice_matches_committed_pngs(golden, tsukuba_oracle, orc):
    # main.cu:172-175 writes slice 0 (d = dmin) of each cost volume
    assert np.array_equal(orc.write_mat_u8(tsukuba_oracle["costl"][0]), golden["cost_lminus15"])
    assert np.array_equal(orc.write_mat_u8(tsukuba_oracle["costr"][0]), golden["cost_rminus15"])


@pytest.mark.parametrize("key", sorted(SURVEY_SHA256))
def test_raw_dumps_match_survey_manifest(tsukuba_oracle, tsukuba_gray, key):
    arrs = dict(tsukuba_oracle)
    arrs["I_l"], arrs["I_r"] = tsukuba_gray
    assert _sha(arrs[key]) == SURVEY_SHA256[key]


def test_label_histogram(tsukuba_oracle):
    # SURVEY.md Appendix C sanity numbers
    lab, cnt = np.unique(tsukuba_oracle["dmapl"], return_counts=True)
    h = dict(zip(lab.astype(int).tolist(), cnt.tolist()))
    assert h[-5] == 57399 and h[-14] == 6527 and -13 not in h
    assert int((tsukuba_oracle["occlusion"] == -115).sum()) == 10605


def test_wta_replay_over_agg_volume(tsukuba_oracle):
    # replaying the `>=` rule over the aggregated volume reproduces the label map
    for side, dmin in (("l", -15), ("r", 0)):
        agg = tsukuba_oracle["agg" + side]
        best = np.full(agg.shape[1:], 0x7F7F7F7F, np.uint32).view(np.float32).copy()
        dmap = np.zeros(agg.shape[1:], np.float32)
        for s in range(agg.shape[0]):
            m = best >= agg[s]
            dmap[m] = dmin + s
            best[m] = agg[s][m]
        assert np.array_equal(dmap, tsukuba_oracle["dmap" + side])
        assert np.array_equal(best, tsukuba_oracle["best" + side])


def test_pack_keys_match_c(orc):
    rng = np.random.default_rng(0)
    vals = np.concatenate([rng.normal(size=200).astype(np.float32),
                           np.array([0.0, -0.0, 3.3961514e38, 1e-45, -1e-45], np.float32)])
    sl = rng.integers(0, 512, size=vals.size)
    keys = orc.pack_keys(vals, sl)
    for v, s, k in zip(vals, sl, keys):
        assert int(orc.lib().orc_pack_key(float(v), int(s))) == int(k)
    b, s2 = orc.unpack_keys(keys)
    assert np.array_equal(s2, sl)
    assert np.array_equal(b, np.where(vals == 0, np.float32(0), vals))
    # signed order: smaller cost first; equal cost -> larger slice first
    o = np.lexsort((-sl, vals))
    assert keys.dtype == np.int64 and np.all(keys[o][:-1] <= keys[o][1:])
