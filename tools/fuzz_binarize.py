#!/usr/bin/env python3
"""Randomised differential run of the binarizers against the CPU oracle (not part of the test suite: a confidence run).

    python tools/fuzz_binarize.py --seconds 120 [--seed 1]

Random method, window, k, morphology, page size, batch size, page content (document, noise, flat, saturated, binary, gradients,
ties on the threshold) and exec mode; consecutive calls share the stream's workspace, so the self-cleaning state of small calls
(StreamWs::clean_pages) sees every transition.  Prints one JSON line; exit code 1 on any mismatch."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import prlib_amd
from prlib_amd import synth
from oracle import capi as oc

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120.0)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--methods", default="0,1,2,3,4", help="comma-separated method ids to draw from (2 = Wolf-Jolion)")
ap.add_argument("--wide", type=float, default=0.0, help="probability of a window from {41..129} (the wide-window paths)")
ap.add_argument("--hooks", type=int, default=0, help="1: load libprlib_hip_testhooks.so (reads the PRL_HIP_* knobs)")
ap.add_argument("--real", type=float, default=0.0, help="probability that a page is cut from one of the reference's scans (tests/golden/scans, "
                "tests/golden/stages: random crop, flip, transposition, gain / offset) instead of being synthetic")
ap.add_argument("--adversarial", type=float, default=0.0, help="probability of a call on large pages of stripes whose levels sit inside the "
                "float32 decision band on every second pixel (bench.adversarial_stripes; methods 0, 1, 3): the refine queue overflows, the "
                "exact sweep redoes the pages")
a = ap.parse_args()
if a.hooks:
    prlib_amd._capi.use_library(prlib_amd._capi.HOOKS_LIB_PATH)
rng = np.random.default_rng(a.seed)
dev = torch.device("cuda:0")
METHODS = [int(m) for m in a.methods.split(",")]


REAL = []
if a.real > 0:
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in sorted(glob.glob(os.path.join(root, "tests", "golden", "scans", "*.npz"))):
        REAL.append(np.load(f)["gray"])
    for f in sorted(glob.glob(os.path.join(root, "tests", "golden", "stages", "chain_*.npz"))):
        REAL.append(oc.bgr2gray(np.load(f)["bgr"]))


def real_page(h, w):
    g = REAL[int(rng.integers(0, len(REAL)))]
    if rng.random() < 0.3:
        g = g.T
    if g.shape[0] < h or g.shape[1] < w:   # tile up to the size
        g = np.tile(g, (-(-h // g.shape[0]), -(-w // g.shape[1])))
    y0, x0 = int(rng.integers(0, g.shape[0] - h + 1)), int(rng.integers(0, g.shape[1] - w + 1))
    c = g[y0:y0 + h, x0:x0 + w]
    if rng.random() < 0.5:
        c = c[:, ::-1]
    if rng.random() < 0.4:   # exposure: gain / offset, clipped (saturated regions as over- and under-exposed scans have)
        c = np.clip(c.astype(np.float32) * float(rng.uniform(0.6, 1.6)) + float(rng.uniform(-60, 60)), 0, 255)
    return np.ascontiguousarray(c).astype(np.uint8)


def page(h, w, kind, i):
    if REAL and rng.random() < a.real:
        return real_page(h, w)
    if kind == 0:
        return synth.page_numpy(h, w, index=int(rng.integers(0, 1 << 20)))
    if kind == 1:
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    if kind == 2:
        return np.full((h, w), int(rng.integers(0, 256)), np.uint8)
    if kind == 3:
        return (rng.integers(0, 2, (h, w)) * 255).astype(np.uint8)
    if kind == 4:
        g = np.add.outer(np.arange(h), np.arange(w)) * (255.0 / max(1, h + w - 2))
        return np.clip(g + rng.normal(0, 2, (h, w)), 0, 255).astype(np.uint8)
    p = synth.page_numpy(h, w, index=i)
    p[rng.integers(0, h):, :] = 255 if rng.integers(0, 2) else 0     # saturated band
    return p


t_end = time.time() + a.seconds
calls = pixels = bad_calls = 0
first_bad = None
stats = {"refined": 0, "exact": 0, "literal_pages": 0, "exact_sweep_pages": 0, "adversarial_calls": 0}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
while time.time() < t_end:
    method = METHODS[int(rng.integers(0, len(METHODS)))]
    win = int(rng.choice([3, 5, 9, 15, 21, 31, 41, 63, 101])) if rng.random() < 0.9 else int(rng.integers(1, 60)) * 2 + 1
    if rng.random() < a.wide:
        win = int(rng.integers(20, 65)) * 2 + 1
    h = int(rng.integers(win + 2, 700)); w = int(rng.integers(win + 2, 1500))
    n = int(rng.choice([1, 1, 2, 3, 5, 8]))
    k = float(rng.choice([0.34, 0.2, -0.2, 0.01, -0.1, 0.5, 0.0])) if rng.random() < 0.8 else float(rng.normal(0, 0.4))
    morph = int(rng.choice([0, 0, 0, 1, 2, -1, -2, 3]))
    pages = None
    if a.adversarial > 0 and rng.random() < a.adversarial and win % 2 == 1 and win >= 5:
        import bench
        method = int(rng.choice([m for m in METHODS if m in (0, 1, 3)] or [0]))
        adv = bench.adversarial_stripes(method, win, k, None)
        if adv is not None and adv[2] < 5e-3:
            lv_a, lv_b, _ = adv
            h, w, n = int(rng.integers(1400, 2400)), int(rng.integers(1600, 2600)), int(rng.choice([1, 2, 3]))
            base = np.where((np.arange(w) + int(rng.integers(0, 2))) % 2 == 0, lv_a, lv_b).astype(np.uint8)[None, :].repeat(h, 0)
            pgs = []
            for i in range(n):
                pg = base.copy()
                if rng.random() < 0.5:   # an island of other content
                    y0, x0 = int(rng.integers(0, h - 200)), int(rng.integers(0, w - 300))
                    pg[y0:y0 + 200, x0:x0 + 300] = page(200, 300, int(rng.integers(0, 6)), i)
                pgs.append(pg)
            pages = np.stack(pgs)
            stats["adversarial_calls"] += 1
    if pages is None:
        pages = np.stack([page(h, w, int(rng.integers(0, 6)), i) for i in range(n)])
    mode = 1 if rng.random() < 0.05 else 0
    p = prlib_amd.make_params(method, win, k, morph)
    if mode:
        prlib_amd.set_exec_mode(1)
    try:
        got = prlib_amd.binarize(torch.from_numpy(pages).to(dev), p).cpu().numpy()
    finally:
        if mode:
            prlib_amd.set_exec_mode(0)
    st = prlib_amd.last_stats()
    stats["refined"] += int(st.refined_pixels); stats["exact"] += int(st.exact_pixels); stats["literal_pages"] += int(st.literal_pages)
    stats["exact_sweep_pages"] += int(st.exact_sweep_pages)
    po = oc.make_params(method, win, k, morph)
    bad = sum(int((got[i] != oc.binarize(pages[i], po)).sum()) for i in range(n))
    calls += 1
    pixels += int(got.size)
    if bad:
        bad_calls += 1
        first_bad = first_bad or {"method": int(method), "window": win, "k": k, "morph": morph, "shape": [n, h, w], "mode": mode, "bad": bad}
print(json.dumps({"seconds": a.seconds, "seed": a.seed, "calls": calls, "Mpixels": round(pixels / 1e6, 1), "mismatching_calls": bad_calls,
                  "first_mismatch": first_bad, **stats}))
sys.exit(1 if bad_calls else 0)
