#!/usr/bin/env python3
"""Writes the progressive-JPEG fixtures of tests/test_jpeg_io.py (tests/golden/jpeg/).  Needs Pillow (libjpeg); the tests do not.

Per case one picture is saved twice with the SAME quantisation — as a progressive file (libjpeg's default scan script: a DC scan with
point transform, spectral bands of the luma AC coefficients, full-band chroma scans, then refinement scans for every band: spectral
selection, successive approximation, DC and AC refinement and end-of-band runs all occur) and as a baseline file.  Both hold identical
quantised coefficients, so a decoder must give identical pixels for the two; the baseline path is pinned independently by the encoder
and the expectation in tests/test_jpeg_io.py.  libjpeg's own decode of the progressive file is stored beside them (jpeg_expected.npz) as
a second, looser anchor: its integer IDCT and its colour conversion round differently from stb_image's (which the library follows), by
a code value or two."""
import os

import numpy as np
from PIL import Image

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "jpeg")
CASES = [  # name, (h, w), grey?, Pillow subsampling (0 = 4:4:4, 1 = 4:2:2, 2 = 4:2:0), quality, restart interval in MCUs (0 = none)
    ("rgb444", (40, 56), False, 0, 90, 0),
    ("rgb420", (50, 70), False, 2, 75, 0),
    ("rgb422_restart", (33, 47), False, 1, 60, 3),
    ("grey", (37, 29), True, 0, 85, 0),
    ("rgb420_tiny", (9, 11), False, 2, 95, 0),
]


def picture(h, w, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([128 + 100 * np.sin(xx / 9.0 + seed), 128 + 90 * np.cos(yy / 7.0), 60 + 1.5 * xx + yy], axis=-1)
    return np.clip(base + rng.normal(0, 6, size=base.shape), 0, 255).astype(np.uint8)


def main():
    os.makedirs(HERE, exist_ok=True)
    expected = {}
    for k, (name, (h, w), grey, sub, quality, restart) in enumerate(CASES):
        arr = picture(h, w, 11 + k)
        img = Image.fromarray(arr[..., 1] if grey else arr)
        extra = {"restart_marker_blocks": restart} if restart else {}
        kw = {} if grey else {"subsampling": sub}
        img.save(os.path.join(HERE, name + "_progressive.jpg"), "JPEG", quality=quality, progressive=True, optimize=True, **kw, **extra)
        img.save(os.path.join(HERE, name + "_baseline.jpg"), "JPEG", quality=quality, progressive=False, optimize=True, **kw, **extra)
        back = np.asarray(Image.open(os.path.join(HERE, name + "_progressive.jpg")))
        expected[name] = back if back.ndim == 3 else back[..., None]
        expected[name + "_source"] = (arr[..., 1:2] if grey else arr)
    np.savez_compressed(os.path.join(HERE, "jpeg_expected.npz"), **expected)
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
