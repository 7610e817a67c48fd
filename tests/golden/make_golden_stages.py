#!/usr/bin/env python3
"""Generates tests/golden/stages/*.npz: the reference's own inputs for the stages around the binarizers.

Run in the BUILD container (it reads /root/reference/test_data; nothing at test time does).  make_golden_scans.py put the
reference's scans through the five binarizers; this script does the same for the rest of the path (VERDICT r4 "next" 2):

  denoise_*.npz   the six images the reference ships for prl::denoise (test_data/denoise/salt_pepper/), whole, as the 8UC3
                  BGR page cv::imread(path) hands the function (as samples/denoise/denoiseSaltPepper_sample.cpp reads its input; a gray or RGBA file
                  becomes 3-channel BGR under IMREAD_COLOR), plus the one RGBA file as 8UC4 (IMREAD_UNCHANGED): the oracle's
                  prl::denoise(strength) for the header default 5.5 (denoiseNLM.h:32) and BASELINE config 4's 10 -
                  CRC-32 of every output, the full output for the images below 200 000 pixels.
  chain_*.npz     ten of the colour originals of test_data/binarize as >= 700 x 1200 BGR pages (whole scans or crops), each
                  with the oracle's prl::deskew (Otsu threshold, HoughLinesP segment count, angle, result size, CRC),
                  prl::backgroundNormalization of the page (CRC), and BASELINE config 5 composed from the oracle's stages -
                  deskew -> denoise(10) -> backgroundNormalization -> BGR2GRAY -> Sauvola(31, 0.34, 0) -> bitwise_not ->
                  thinZhangSuen - with the CRC after every stage and the final skeleton bit-packed.

A fixture holds DATA only (pixels in, numbers / checksums / packed masks out).  The reference ships no expected outputs and
cannot be built here, so the outputs are the CPU oracle's ("parity unpinned"); tests/test_real_stages_gpu.py recomputes them
with the oracle on the CPU (`-m "not gpu"`) and compares the HIP path with them on the GPU.
"""
import os
import sys
import zlib

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import capi  # noqa: E402

REF = "/root/reference/test_data"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stages")

DENOISE = ["105053.png", "Figure-14_lightbox.png", "WithSaltAndPepper.jpg", "butterfly_sp.png", "lone-banana-noise.jpg", "zJRl7.png"]
STRENGTHS = (5.5, 10.0)
# (name, x0, y0, width, height); None = the whole scan.  Chosen for the spread of findAngle results on real paper: no rotation,
# ~ +-1 degree, exactly -45 (diagonal rulings dominate the vote), exactly -90, and 1400 - 1750 Hough segments per page.
CHAIN = [("0004", 90, 150, 900, 1300), ("0008", 300, 0, 1240, 1313), ("0034", 90, 300, 900, 1300), ("0096", 100, 200, 900, 1400),
         ("0110", None, None, None, None), ("0120", 150, 100, 1000, 1300), ("0138", 60, 150, 900, 1300),
         ("0150", 100, 300, 940, 1400), ("0158", 75, 150, 900, 1300), ("0195", None, None, None, None)]
CHAIN_PARAMS = dict(window=31, k=0.34, morph=0, strength=10.0, thin=0)


def crc(a: np.ndarray) -> int:
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def imread_color(path):
    """cv::imread(path) (IMREAD_COLOR): 8UC3 BGR whatever the file holds."""
    return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8)[:, :, ::-1])


def imread_unchanged(path):
    """cv::imread(path, IMREAD_UNCHANGED) of an RGBA file: 8UC4 BGRA."""
    a = np.asarray(Image.open(path), dtype=np.uint8)
    assert a.ndim == 3 and a.shape[2] == 4
    return np.ascontiguousarray(a[:, :, [2, 1, 0, 3]])


def emit_denoise(tag, img):
    rec = {"bgr": img, "strengths": np.array(STRENGTHS)}
    for s in STRENGTHS:
        out = capi.denoise(img, s, threads=8)
        key = ("%g" % s).replace(".", "p")
        rec["crc_" + key] = np.array([crc(out)], np.uint32)
        if img.shape[0] * img.shape[1] < 200_000:
            rec["out_" + key] = out
    path = os.path.join(OUT, tag + ".npz")
    np.savez_compressed(path, **rec)
    print(tag, img.shape, os.path.getsize(path) // 1024, "KiB", flush=True)


def chain_stages(bgr):
    """BASELINE config 5 from the oracle's stages, the intermediate after every stage (what a PRLib user's loop holds)."""
    p = CHAIN_PARAMS
    rot, info = capi.deskew(bgr)
    rot = np.ascontiguousarray(rot)
    den = capi.denoise(rot, p["strength"], threads=8)
    bg = capi.bgnorm(np.ascontiguousarray(den))
    gray = capi.bgr2gray(np.ascontiguousarray(bg))
    mask = capi.binarize(np.ascontiguousarray(gray), capi.make_params(capi.SAUVOLA, p["window"], p["k"], p["morph"]))
    skel = capi.thin(255 - mask, p["thin"])
    return info, rot, den, bg, gray, mask, skel


def emit_chain(tag, bgr):
    info, rot, den, bg, gray, mask, skel = chain_stages(bgr)
    _, binary = capi.otsu(capi.bgr2gray(bgr))
    ang2, nseg = capi.find_angle(binary)
    assert ang2 == info["angle"] and nseg == info["n_lines"]
    rec = {"bgr": bgr,
           "deskew_angle": np.array([info["angle"]], np.float64), "deskew_otsu": np.array([info["otsu"]], np.int32),
           "deskew_segments": np.array([info["n_lines"]], np.int32), "deskew_shape": np.array(rot.shape, np.int32),
           "crc_deskew": np.array([crc(rot)], np.uint32),
           "crc_bgnorm_of_input": np.array([crc(capi.bgnorm(bgr))], np.uint32),
           "crc_chain_denoise": np.array([crc(den)], np.uint32), "crc_chain_bgnorm": np.array([crc(bg)], np.uint32),
           "crc_chain_gray": np.array([crc(gray)], np.uint32), "crc_chain_mask": np.array([crc(mask)], np.uint32),
           "crc_chain_skeleton": np.array([crc(skel)], np.uint32),
           "chain_skeleton": np.packbits(skel > 0, axis=1), "chain_shape": np.array(skel.shape, np.int32),
           "chain_params": np.array([CHAIN_PARAMS[k] for k in ("window", "k", "morph", "strength", "thin")], np.float64)}
    path = os.path.join(OUT, tag + ".npz")
    np.savez_compressed(path, **rec)
    print(tag, bgr.shape, "angle", info["angle"], "segments", info["n_lines"], "->", rot.shape[:2], "skeleton px", int((skel > 0).sum()),
          os.path.getsize(path) // 1024, "KiB", flush=True)


def main():
    os.makedirs(OUT, exist_ok=True)
    for f in DENOISE:
        emit_denoise("denoise_" + os.path.splitext(f)[0].replace("-", "_"), imread_color(os.path.join(REF, "denoise", "salt_pepper", f)))
    emit_denoise("denoise_lone_banana_noise_bgra", imread_unchanged(os.path.join(REF, "denoise", "salt_pepper", "lone-banana-noise.jpg")))
    for name, x0, y0, w, h in CHAIN:
        a = imread_color(os.path.join(REF, "binarize", name + ".png"))
        if x0 is None:
            tag = f"chain_{name}"
        else:
            assert y0 + h <= a.shape[0] and x0 + w <= a.shape[1], (name, a.shape)
            a = np.ascontiguousarray(a[y0:y0 + h, x0:x0 + w])
            tag = f"chain_{name}_x{x0}_y{y0}_{w}x{h}"
        assert min(a.shape[:2]) >= 700 and max(a.shape[:2]) >= 1200
        emit_chain(tag, a)
    total = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("done:", total // 1024, "KiB")


if __name__ == "__main__":
    main()
