"""The C++ drop-in host layer (stereo_matching_cuda_amd/host/: main.cpp + the reference's per-stage
headers) built against libsmx_hip.so.  CPU: it builds, keeps the reference's function names, and
fails loudly without a GPU.  GPU: running the drop-in main on the reference's Tsukuba pair writes
12 PNGs whose pixels equal the 12 PNGs the reference's authors committed."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "stereo_matching_cuda_amd", "host")
BIN = os.path.join(ROOT, "stereo_matching_cuda_amd", "_build", "smx_main")

REFERENCE_SIGNATURES = {   # reference header -> function names it must still declare
    "rgb_to_grayscale.cuh": ["rgb_to_grayscale"],
    "costVolume.cuh": ["compute_cost"],
    "guidedFilter.cuh": ["compute_guided_filter"],
    "integral.cuh": ["integral"],
    "occlusion.cuh": ["detect_occlusion", "fill_occlusion"],
    "filter.cuh": ["filter"],
    "helpers.cuh": ["check_errors"],
    "winner_take_all.cuh": ["wta_pack"],
}

OUTPUTS = ["image_left", "image_right", "image_mean_left", "image_mean_right", "best_costl",
           "best_costr", "cost_lminus15", "cost_rminus15", "occlu_mapl", "disparity_mapl",
           "disparity_mapr", "occlu_mapl_filled"]


@pytest.fixture(scope="module")
def binary():
    subprocess.check_call(["make", "-s", "-C", HOST])
    assert os.path.exists(BIN)
    return BIN


def test_headers_keep_the_reference_names():
    for header, names in REFERENCE_SIGNATURES.items():
        src = open(os.path.join(HOST, header)).read()
        for n in names:
            assert re.search(r"\b%s\s*\(" % n, src), (header, n)


def test_host_layer_builds_and_links_the_c_abi(binary):
    out = subprocess.run(["ldd", binary], capture_output=True, text=True).stdout
    assert "libsmx_hip.so" in out


def test_main_fails_loudly_without_gpu(binary, tmp_path):
    import stereo_matching_cuda_amd as smx
    if smx.lib().smx_device_count() > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([binary], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode != 0 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_drop_in_main_reproduces_the_committed_images(binary, golden, tmp_path):
    PIL = pytest.importorskip("PIL.Image")
    data = tmp_path / "data"
    data.mkdir()
    for n in ("tsukuba0", "tsukuba1"):
        src = os.path.join(ROOT, "tests", "golden", "tsukuba", n + ".png")
        (data / (n + ".png")).write_bytes(open(src, "rb").read())
    r = subprocess.run([binary], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    for line in ("Starting...", "Resolution : 384x288", "RGB to grayscale ...", "Cost Volume ...",
                 "guided filter ...", "guided filter ok", "writing images ...", "duration:",
                 "Free the memory ..."):
        assert line in r.stdout, line                      # main.cu:41-186 progress lines
    for name in OUTPUTS:
        got = np.asarray(PIL.open(data / (name + ".png")))
        assert np.array_equal(got, golden[name]), name
