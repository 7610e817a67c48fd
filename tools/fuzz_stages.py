#!/usr/bin/env python3
"""Randomised differential run of the round-2 stages against the CPU oracle (confidence run, not part of the test suite):
rotate, deskew (HoughLinesP + angle + warp), backgroundNormalization, thinning, NL-means (small crops: the oracle is a scalar
CPU loop), binarizeByLocalVariancesWithoutFilters.

    python tools/fuzz_stages.py --seconds 150 [--seed 1]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import prlib_amd
from prlib_amd import synth
from oracle import capi as oc

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=150.0)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--only", default="", help="comma-separated stages: rotate,deskew,houghp,bgnorm,thin,denoise,lv (default: all)")
ap.add_argument("--real", type=float, default=0.0, help="fraction of the deskew / houghp pages cut from the reference's scans (tests/golden)")
ap.add_argument("--max-side", type=int, default=520, help="largest page side of the deskew / houghp cases")
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
dev = torch.device("cuda:0")
counts, bad = {}, {}


def note(name, ok, detail):
    counts[name] = counts.get(name, 0) + 1
    if not ok and name not in bad:
        bad[name] = detail


def colour(g, ch):
    if ch == 1:
        return g
    return np.clip(g[..., None].astype(np.int32) + rng.integers(-6, 7, g.shape + (ch,)), 0, 255).astype(np.uint8)


STAGES = ["rotate", "deskew", "bgnorm", "thin", "denoise", "lv", "houghp"]
only = [STAGES.index(x) for x in a.only.split(",") if x] or list(range(len(STAGES)))
_real = []


def real_gray(h, w):
    """An h x w crop (random position, flip, transposition, gain) of one of the reference's scans committed as fixtures."""
    import glob
    if not _real:
        root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
        for pth in sorted(glob.glob(os.path.join(root, "stages", "chain_*.npz"))):
            b = np.load(pth)["bgr"].astype(np.int32)
            _real.append(((b[..., 0] * 1868 + b[..., 1] * 9617 + b[..., 2] * 4899 + 8192) >> 14).astype(np.uint8))
        for pth in sorted(glob.glob(os.path.join(root, "scans", "*.npz")))[:12]:
            z = np.load(pth)
            key = "gray" if "gray" in z.files else z.files[0]
            if z[key].ndim == 2:
                _real.append(z[key].astype(np.uint8))
    g = _real[int(rng.integers(0, len(_real)))]
    if rng.random() < 0.5:
        g = g.T
    if rng.random() < 0.5:
        g = g[:, ::-1]
    reps = (-(-h // g.shape[0]), -(-w // g.shape[1]))
    g = np.tile(g, reps)
    y0, x0 = int(rng.integers(0, g.shape[0] - h + 1)), int(rng.integers(0, g.shape[1] - w + 1))
    g = g[y0:y0 + h, x0:x0 + w].astype(np.float64) * rng.uniform(0.7, 1.2) + rng.uniform(-20, 20)
    return np.ascontiguousarray(np.clip(g, 0, 255).astype(np.uint8))


def text_or_real(h, w):
    if a.real > 0 and rng.random() < a.real:
        return real_gray(h, w)
    g = synth.text_page_numpy(h, w, int(rng.integers(0, 1 << 20)), skew_deg=float(rng.uniform(-5, 5)), shading=float(rng.uniform(0, 0.5)))
    if rng.random() < 0.3:   # clutter
        g = np.where(rng.random((h, w)) < 0.02, rng.integers(0, 120, (h, w)), g).astype(np.uint8)
    return g


t_end = time.time() + a.seconds
while time.time() < t_end:
    which = int(rng.choice(only))
    ch = int(rng.choice([1, 3, 4]))
    if which == 0:    # rotate
        h, w = int(rng.integers(1, 300)), int(rng.integers(1, 400))
        img = colour(rng.integers(0, 256, (h, w), dtype=np.uint8), ch)
        ang = float(rng.choice([90.0, 180.0, 270.0, -90.0, 0.0, 450.0])) if rng.random() < 0.25 else float(rng.uniform(-200, 200))
        got = prlib_amd.rotate(torch.from_numpy(img).to(dev)[None], [ang])[0].cpu().numpy()
        want = oc.rotate(img, ang)
        note("rotate", got.shape == want.shape and np.array_equal(got, want), {"shape": [h, w, ch], "angle": ang})
    elif which == 1:  # deskew: one call of 1..5 pages (a group of workgroups per page, the queue of pages)
        h, w = int(rng.integers(120, max(121, a.max_side - 100))), int(rng.integers(160, max(161, a.max_side)))
        ch = int(rng.choice([1, 3, 4]))
        n = int(rng.choice([1, 1, 2, 5]))
        imgs = np.stack([colour(text_or_real(h, w), ch) for _ in range(n)])
        outs, angs = prlib_amd.deskew(torch.from_numpy(imgs).to(dev))
        for i in range(n):
            want, info = oc.deskew(imgs[i])
            got = outs[i].cpu().numpy()
            note("deskew", angs[i] == info["angle"] and got.shape == want.shape and np.array_equal(got, want),
                 {"shape": [h, w, ch], "pages": n, "page": i, "angle_gpu": float(angs[i]), "angle_cpu": info["angle"]})
    elif which == 6:  # cv::HoughLinesP with its own parameters
        h, w = int(rng.integers(40, max(41, a.max_side - 100))), int(rng.integers(40, max(41, a.max_side)))
        g = text_or_real(h, w)
        thr_img = int(rng.integers(60, 200))
        pts = (g < thr_img).astype(np.uint8) * int(rng.integers(1, 256))
        thr, ll, gap = int(rng.choice([16, 30, 60, 100, 150, 12])), int(rng.integers(5, max(6, w // 3))), int(rng.choice([0, 1, 2, 5, 20, 63, 64, 70]))
        got = prlib_amd.houghp(torch.from_numpy(np.ascontiguousarray(pts)).to(dev), thr, ll, gap)
        want = oc.houghp(np.ascontiguousarray(pts), thr, ll, gap)
        note("houghp", np.array_equal(got, want), {"shape": [h, w], "threshold": thr, "line_length": ll, "line_gap": gap, "n_gpu": len(got), "n_cpu": len(want)})
    elif which == 2:  # backgroundNormalization
        h, w = int(rng.integers(1, 500)), int(rng.integers(1, 700))
        g = synth.text_page_numpy(max(h, 20), max(w, 20), int(rng.integers(0, 1 << 20)), shading=float(rng.uniform(0, 0.6)))[:h, :w]
        if rng.random() < 0.2:
            g = rng.integers(0, 256, (h, w), dtype=np.uint8)
        img = colour(np.ascontiguousarray(g), ch)
        got = prlib_amd.backgroundNormalization(torch.from_numpy(img).to(dev)).cpu().numpy()
        want = oc.bgnorm(img)
        note("bgnorm", got.shape == want.shape and np.array_equal(got, want), {"shape": [h, w, ch]})
    elif which == 3:  # thinning
        h, w = int(rng.integers(1, 300)), int(rng.integers(1, 500))
        m = (synth.page_numpy(max(h, 8), max(w, 8), index=int(rng.integers(0, 1 << 20)))[:h, :w] < 128).astype(np.uint8) * 255
        if rng.random() < 0.3:
            m = (rng.random((h, w)) < rng.uniform(0.05, 0.9)).astype(np.uint8) * 255
        method = int(rng.integers(0, 2))
        fn = prlib_amd.thinZhangSuen if method == 0 else prlib_amd.thinGuoHall
        got = fn(torch.from_numpy(np.ascontiguousarray(m)).to(dev)).cpu().numpy()
        want = oc.thin(np.ascontiguousarray(m), method)
        note("thin", np.array_equal(got, want), {"shape": [h, w], "method": method})
    elif which == 4:  # NL-means, small
        h, w = int(rng.integers(1, 90)), int(rng.integers(1, 110))
        ch = int(rng.choice([3, 4]))
        g = synth.page_numpy(max(h, 8), max(w, 8), index=int(rng.integers(0, 1 << 20)))[:h, :w]
        img = np.clip(colour(np.ascontiguousarray(g), ch).astype(np.int32) + rng.normal(0, 12, (h, w, ch)), 0, 255).astype(np.uint8)
        strength = float(rng.choice([10.0, 5.5, 3.0, 1.0, 20.0]))
        got = prlib_amd.denoise(torch.from_numpy(img).to(dev), strength).cpu().numpy()
        want = oc.denoise(img, strength, threads=8)
        note("denoise", np.array_equal(got, want), {"shape": [h, w, ch], "strength": strength})
    else:             # local variances without filters
        h, w = int(rng.integers(1, 400)), int(rng.integers(1, 600))
        g = synth.page_numpy(max(h, 8), max(w, 8), index=int(rng.integers(0, 1 << 20)))[:h, :w]
        img = colour(np.ascontiguousarray(g), 3)
        coeff, mv = float(rng.choice([0.125, 0.05, 0.3])), int(rng.choice([10, 25, 2]))
        got = prlib_amd.binarizeByLocalVariancesWithoutFilters(torch.from_numpy(img).to(dev), coeff, mv).cpu().numpy()
        want = oc.binarize_lv_nofilters(img, coeff, mv)
        note("lv_nofilters", np.array_equal(got, want), {"shape": [h, w], "coeff": coeff, "min_var": mv})
print(json.dumps({"seconds": a.seconds, "seed": a.seed, "calls": counts, "mismatches": bad}))
sys.exit(1 if bad else 0)
