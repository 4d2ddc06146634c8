#!/usr/bin/env python3
"""Decode the Tsukuba PNG fixtures (data files committed by the reference's authors in
stereo_matching_cuda/data/, copied verbatim to tests/golden/tsukuba/) into one .npz so the
tests need numpy only.  Re-run after touching the PNGs:  python tools/make_golden.py
"""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "golden", "tsukuba")
NAMES = ["tsukuba0", "tsukuba1", "image_left", "image_right", "image_mean_left",
         "image_mean_right", "cost_lminus15", "cost_rminus15", "best_costl", "best_costr",
         "disparity_mapl", "disparity_mapr", "occlu_mapl", "occlu_mapl_filled"]


def main():
    out = {}
    for n in NAMES:
        a = np.asarray(Image.open(os.path.join(SRC, n + ".png")))
        out[n] = np.ascontiguousarray(a)
        print(n, a.shape, a.dtype, file=sys.stderr)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "tsukuba_golden.npz"), **out)


if __name__ == "__main__":
    main()
