#!/usr/bin/env python3
"""Randomised differential run of the one-call chain (prl_hip_chain_pages_device / _batch_device through prlib_amd.process_pages,
and prl_hip_chain_batch_host through process_pages_host) against the oracle's stages composed on the host.  Random stage
subsets, page sizes, batch sizes (mixing skewed, straight and blank pages: runs of different result sizes), channels.

    python tools/fuzz_chain.py --seconds 150 [--seed 1]"""
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
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
dev = torch.device("cuda:0")
METHODS = {prlib_amd.SAUVOLA: oc.SAUVOLA, prlib_amd.NIBLACK: oc.NIBLACK, prlib_amd.NICK: oc.NICK, prlib_amd.WOLFJOLION: oc.WOLFJOLION}


def oracle_chain(img, method, w, k, morph, strength, thin, bgnorm, deskew):
    cur, angle = img, 0.0
    if deskew:
        cur, info = oc.deskew(img)
        angle = info["angle"]
    if strength is not None:
        cur = oc.denoise(np.ascontiguousarray(cur), strength, threads=8)
    if bgnorm:
        cur = oc.bgnorm(np.ascontiguousarray(cur))
    if cur.ndim == 3:
        cur = oc.bgr2gray(np.ascontiguousarray(cur))
    mask = oc.binarize(np.ascontiguousarray(cur), oc.make_params(METHODS[method], w, k, morph))
    return (mask if thin < 0 else oc.thin(255 - mask, thin)), angle


t_end = time.time() + a.seconds
calls = pages_done = 0
bad = None
while time.time() < t_end and bad is None:
    ch = int(rng.choice([1, 3, 4]))
    h, w = int(rng.integers(70, 150)), int(rng.integers(80, 200))
    n = int(rng.integers(1, 6))
    method = int(rng.choice(list(METHODS)))
    win = int(rng.choice([15, 21, 31]))
    k = {prlib_amd.SAUVOLA: 0.34, prlib_amd.NIBLACK: -0.2, prlib_amd.NICK: -0.1, prlib_amd.WOLFJOLION: 0.3}[method]
    morph = int(rng.choice([0, 0, 1, 2]))
    deskew = bool(rng.random() < 0.7)
    bgnorm = bool(rng.random() < 0.6)
    strength = float(rng.choice([10.0, 3.0])) if (ch != 1 and rng.random() < 0.5) else None
    thin = int(rng.choice([-1, 0, 1]))
    pages = []
    for i in range(n):
        r = rng.random()
        if r < 0.15:
            g = np.full((h, w), int(rng.integers(150, 256)), np.uint8)
        else:
            g = synth.text_page_numpy(h, w, int(rng.integers(0, 1 << 20)), skew_deg=0.0 if r < 0.35 else float(rng.uniform(-4, 4)),
                                      shading=float(rng.uniform(0, 0.4)))
        pages.append(g if ch == 1 else np.clip(g[..., None].astype(np.int32) + rng.integers(-5, 6, (h, w, ch)), 0, 255).astype(np.uint8))
    batch = np.stack(pages)
    host = rng.random() < 0.3
    if host:
        outs, angles = prlib_amd.process_pages_host(list(batch), method, win, k, morph, denoise_strength=strength, thin=thin, deskew=deskew,
                                                    background_normalization=bgnorm, n_devices=1)
        outs = [np.asarray(o) for o in outs]
    else:
        r = prlib_amd.process_pages(torch.from_numpy(batch).to(dev), ch, method, win, k, morph, denoise_strength=strength, thin=thin,
                                    deskew=deskew, background_normalization=bgnorm)
        if deskew:
            outs, angles = [o.cpu().numpy() for o in r[0]], r[1]
        else:
            outs, angles = [o for o in r.cpu().numpy()], np.zeros(n)
    for i in range(n):
        want, ang = oracle_chain(batch[i], method, win, k, morph, strength, thin, bgnorm, deskew)
        if outs[i].shape != want.shape or not np.array_equal(outs[i], want) or float(angles[i]) != ang:
            bad = {"page": i, "n": n, "shape": [h, w, ch], "method": method, "window": win, "morph": morph, "deskew": deskew, "bgnorm": bgnorm,
                   "strength": strength, "thin": thin, "host": bool(host), "angle_gpu": float(angles[i]), "angle_cpu": ang}
            break
    calls += 1
    pages_done += n
print(json.dumps({"seconds": a.seconds, "seed": a.seed, "chain_calls": calls, "pages": pages_done, "first_mismatch": bad}))
sys.exit(1 if bad else 0)
