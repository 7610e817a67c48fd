#!/usr/bin/env python3
"""Secondary measurement (BASELINE config 4): prl::denoise on N x 4096^2 x 3 noisy scans, 1 GPU."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import prlib_amd
from prlib_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--pages", type=int, default=8)
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--strength", type=float, default=10.0)
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--check", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda:0")
gray = synth.pages_torch(a.pages, a.size, a.size, dev)
gen = torch.Generator(device=dev); gen.manual_seed(7)
img = gray[..., None].float().expand(-1, -1, -1, 3) + torch.randn((a.pages, a.size, a.size, 3), device=dev, generator=gen) * 15.0
img = img.round_().clamp_(0, 255).to(torch.uint8).contiguous()
out = torch.empty_like(img)
prlib_amd.denoise(img, a.strength, out=out); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    prlib_amd.denoise(img, a.strength, out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
px = a.pages * a.size * a.size
res = {"workload": f"prl::denoise strength={a.strength} on {a.pages} x {a.size}^2 x 3", "ms_per_batch": round(dt * 1e3, 2),
       "Mpixels/s": round(px / dt / 1e6, 1), "algorithmic_GB/s (6 B/px)": round(6 * px / dt / 1e9, 2),
       "frac_of_8TB/s": round(6 * px / dt / 8e12, 5)}
if a.check:
    from oracle import capi as oc
    sub = img[0, :256, :256].cpu().numpy().copy()
    got = prlib_amd.denoise(torch.from_numpy(sub).to(dev), a.strength).cpu().numpy()
    res["mismatch_256x256_vs_oracle"] = int((got != oc.denoise(sub, a.strength, threads=os.cpu_count())).sum())
print(json.dumps(res))
