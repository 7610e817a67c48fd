#!/usr/bin/env python3
"""Generates tests/golden/*.npz.  Run in the BUILD container (it reads the reference's test images).

The reference ships inputs only (test_data/binarize/*.png, no expected outputs, no tests), and cannot be
built here (OpenCV/Leptonica absent), so the expected outputs are produced by the CPU oracle
(oracle/prl_oracle.c, "parity unpinned") and cross-checked, at generation time, against the independent
whole-plane numpy model (oracle/numpy_model.py): a fixture is only written if both agree bit for bit.

Fixtures hold DATA only: decoded input pixels (imread semantics: BGR, alpha dropped), the gray plane
cvtColor would produce, and the output mask per configuration.
"""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import capi, numpy_model as nm  # noqa: E402

REF = "/root/reference/test_data/binarize"
OUT = os.path.dirname(os.path.abspath(__file__))

# (method, window, k, morph) — header defaults where the image is large enough, plus small windows
CONFIGS = [
    (capi.SAUVOLA, 31, 0.34, 0), (capi.SAUVOLA, 15, 0.34, 2), (capi.SAUVOLA, 101, 0.01, 2),
    (capi.NIBLACK, 31, 0.01, 2), (capi.NIBLACK, 15, -0.2, 0),
    (capi.WOLFJOLION, 31, 0.3, 0), (capi.WOLFJOLION, 15, 0.01, 2),
    (capi.NICK, 21, -0.01, 0), (capi.NICK, 15, -0.1, -1),
    (capi.FENG, 21, 0.0, 2), (capi.FENG, 15, 0.0, 0),
]


def imread_bgr(path):
    im = Image.open(path)
    if im.mode == "L":
        return np.asarray(im, dtype=np.uint8)
    rgb = np.asarray(im.convert("RGB"), dtype=np.uint8)  # cv::imread(IMREAD_COLOR) drops alpha
    return np.ascontiguousarray(rgb[:, :, ::-1])


def main():
    for name in ["0041.png", "0047.png", "0050.png", "0081.png", "0085.png", "0200.png"]:
        src = imread_bgr(os.path.join(REF, name))
        gray = src if src.ndim == 2 else capi.bgr2gray(src)
        rec = {"source": src, "gray": gray}
        keys = []
        for (m, w, k, mo) in CONFIGS:
            p = capi.make_params(m, w, k, mo)
            try:
                out = capi.binarize(gray, p)
            except capi.OracleError as e:
                assert e.status == capi.PRL_ERR_EMPTY_RECT
                rec[f"status_{m}_{w}_{k}_{mo}"] = np.int32(e.status)
                continue
            ref2 = nm.binarize(gray, m, w, k, mo)
            assert np.array_equal(out, ref2), (name, m, w, k, mo)
            rec[f"mask_{m}_{w}_{k}_{mo}"] = np.packbits(out > 0, axis=1)
            rec[f"shape_{m}_{w}_{k}_{mo}"] = np.array(out.shape, np.int32)
            keys.append((m, w, k, mo))
        np.savez_compressed(os.path.join(OUT, name.replace(".png", ".npz")), **rec)
        print(name, src.shape, len(keys), "masks")

    # synthetic known-answer pages (hand-derivable)
    flat = np.full((64, 64), 255, np.uint8)
    t = capi.threshold_plane(flat, capi.make_params(capi.SAUVOLA, 31, 0.34, 0))
    np.savez_compressed(os.path.join(OUT, "kat_flat255.npz"), page=flat, threshold_00=t[0, 0],
                        mask=capi.binarize(flat, capi.make_params(capi.SAUVOLA, 31, 0.34, 0)))

    # NL-means: one small noisy colour crop with the planes and the final image
    rng = np.random.default_rng(5)
    crop = imread_bgr(os.path.join(REF, "0050.png"))[:48, :64].astype(np.float64)
    noisy = np.clip(np.rint(crop + rng.normal(0, 15, crop.shape)), 0, 255).astype(np.uint8)
    lab = capi.lbgr2lab(noisy)
    np.savez_compressed(os.path.join(OUT, "nlm_0050_crop.npz"), noisy=noisy, lab=lab,
                        l_h10=capi.nlm_planes(np.ascontiguousarray(lab[:, :, 0]), 10.0),
                        ab_h3=capi.nlm_planes(np.ascontiguousarray(lab[:, :, 1:]), 3.0),
                        denoised_s10=capi.denoise(noisy, 10.0))
    print("done")


if __name__ == "__main__":
    main()
