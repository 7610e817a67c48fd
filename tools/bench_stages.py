#!/usr/bin/env python3
"""Round-2 stages on one GPU, device resident, A4 pages: time, Mpixels/s and the fraction of the 8 TB/s HBM roofline on each
stage's ALGORITHMIC bytes (read + written once).  One JSON object."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import prlib_amd
from prlib_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--pages", type=int, default=64)
ap.add_argument("--width", type=int, default=2480)
ap.add_argument("--height", type=int, default=3508)
ap.add_argument("--steps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0")
gray, _ = synth.text_pages_torch(a.pages, a.height, a.width, dev, channels=1)
col, _ = synth.text_pages_torch(a.pages, a.height, a.width, dev, channels=3, seed=7100)
px = a.pages * a.width * a.height


def timed(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.steps


def row(name, dt, bytes_per_px, pixels=px):
    return {"stage": name, "ms": round(dt * 1e3, 3), "Mpx_s": round(pixels / dt / 1e6, 1),
            "algorithmic_B_per_px": bytes_per_px, "hbm_frac": round(bytes_per_px * pixels / dt / 8e12, 4)}


rows = []
rows.append(row("backgroundNormalization 1ch", timed(lambda: prlib_amd.backgroundNormalization(gray)), 2))
rows.append(row("backgroundNormalization 3ch", timed(lambda: prlib_amd.backgroundNormalization(col)), 6))
rows.append(row("binarizeByLocalVariancesWithoutFilters", timed(lambda: prlib_amd.binarizeByLocalVariancesWithoutFilters(col)), 4))
rows.append(row("binarizeByLocalVariances", timed(lambda: prlib_amd.binarizeByLocalVariances(col)), 4))
ang = np.full(a.pages, 3.0)
ln = max(a.width, a.height)
rows.append(row("rotate 3.0 deg 1ch (len x len canvas)", timed(lambda: prlib_amd.rotate(gray, ang)), 2, a.pages * ln * ln))
rows.append(row("rotate 3.0 deg 3ch", timed(lambda: prlib_amd.rotate(col, ang)), 6, a.pages * ln * ln))
rows.append(row("rotate 90 deg 3ch", timed(lambda: prlib_amd.rotate(col, np.full(a.pages, 90.0))), 6))
rows.append(row("cvtColor BGR2GRAY", timed(lambda: prlib_amd.cvtColorBGR2GRAY(col)), 4))
mask = prlib_amd.binarizeSauvola(gray, 31, 0.34, 0)
inv = prlib_amd.bitwise_not(mask)
rows.append(row("thinZhangSuen", timed(lambda: prlib_amd.thinZhangSuen(inv)), 2, int(inv.numel())))
print(json.dumps({"workload": f"{a.pages} x {a.width}x{a.height} text pages, 1 GPU, device resident", "stages": rows}))
