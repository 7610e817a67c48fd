#!/usr/bin/env python3
"""Generates tests/golden/scans/*.npz: the reference's own test scans at sizes that reach the interior strips.

Run in the BUILD container (it reads /root/reference/test_data/binarize; nothing at test time does).  The small fixtures of
make_golden.py are at most 262 pixels wide, so on the device they only ever run edge strips; these are whole scans of
>= 700 x 1200 pixels, 1024 x 1536 crops of the four largest scans and one 5312-column band, which run the float32 interior
pipeline, the tiers, the refine queue and Wolf-Jolion's candidate list (VERDICT r3 "next" 1).

The reference ships inputs only (no expected outputs, no tests) and cannot be built here, so the expected masks come from
the CPU oracle ("parity unpinned") and a fixture is only written when the independent whole-plane numpy model
(oracle/numpy_model.py) produces the same mask bit for bit.

A fixture holds DATA only: the gray plane cv::imread + cvtColor(BGR2GRAY) would hand the binarizer
(binarizeSauvola_sample.cpp:48-53 reads the file, binarizeSauvola.cpp:51 converts it) and, per configuration, the
bit-packed output mask.  Configurations: the five header-default calls (binarizeSauvola.h:43-47, binarizeNiblack.h:43-47,
binarizeWolfJolion.h:43-47, binarizeNICK.h:43-47, binarizeFeng.h:46-53) and the headline parameters of BASELINE.json
(Sauvola w=31, k=0.34, no morphology).
"""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import capi, numpy_model as nm  # noqa: E402

REF = "/root/reference/test_data/binarize"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scans")

WHOLE = ["0004", "0008", "0010", "0020", "0034", "0035", "0077", "0083", "0096", "0110", "0120", "0126", "0138", "0150",
         "0158", "0176", "0195", "0196", "0197", "0198"]
# (name, x0, y0, width, height): 1024 x 1536 crops of the four largest scans, and a full-width band of the widest one
CROPS = [("0018", 1000, 400, 1536, 1024), ("0037", 700, 900, 1024, 1536), ("0064", 1800, 700, 1536, 1024),
         ("0105", 900, 500, 1536, 1024), ("0064", 0, 1300, 5312, 384)]

# key -> (method, window, k, morph); None = the header default
CONFIGS = {
    "sauvola_default": (capi.SAUVOLA, None, None, None),
    "niblack_default": (capi.NIBLACK, None, None, None),
    "wolfjolion_default": (capi.WOLFJOLION, None, None, None),
    "nick_default": (capi.NICK, None, None, None),
    "feng_default": (capi.FENG, None, None, None),
    "sauvola_headline": (capi.SAUVOLA, 31, 0.34, 0),
}


def imread_gray(path):
    im = Image.open(path)
    if im.mode == "L":
        return np.asarray(im, dtype=np.uint8)
    rgb = np.asarray(im.convert("RGB"), dtype=np.uint8)  # cv::imread(IMREAD_COLOR) drops alpha
    return capi.bgr2gray(np.ascontiguousarray(rgb[:, :, ::-1]))


def emit(tag, gray):
    rec = {"gray": gray}
    for key, (m, w, k, mo) in CONFIGS.items():
        p = capi.make_params(m, w, k, mo)
        out = capi.binarize(gray, p)
        ref2 = nm.binarize(gray, m, p.window_size, p.k, p.morph_iterations)
        assert np.array_equal(out, ref2), (tag, key)
        rec["mask_" + key] = np.packbits(out > 0, axis=1)
        rec["shape_" + key] = np.array(out.shape, np.int32)
        rec["params_" + key] = np.array([m, p.window_size, p.k, p.morph_iterations], np.float64)
    path = os.path.join(OUT, tag + ".npz")
    np.savez_compressed(path, **rec)
    print(tag, gray.shape, os.path.getsize(path) // 1024, "KiB", flush=True)


def main():
    os.makedirs(OUT, exist_ok=True)
    for name in WHOLE:
        emit(name, imread_gray(os.path.join(REF, name + ".png")))
    for name, x0, y0, cw, ch in CROPS:
        g = imread_gray(os.path.join(REF, name + ".png"))
        assert y0 + ch <= g.shape[0] and x0 + cw <= g.shape[1], (name, g.shape)
        emit(f"{name}_x{x0}_y{y0}_{cw}x{ch}", np.ascontiguousarray(g[y0:y0 + ch, x0:x0 + cw]))
    print("done")


if __name__ == "__main__":
    main()
