#!/usr/bin/env python3
"""End-to-end throughput of the host-page path (SURVEY.md §8d "second number"): pages in host memory ->
prl_hip_binarize_batch_host (H2D + kernels + D2H, sharded over --devices GPUs) -> masks in host memory."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import prlib_amd
from prlib_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--pages", type=int, default=256)
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--devices", type=int, default=0)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--pinned", type=int, default=0, help="1: pages and masks in pinned memory (prl_hip_alloc_host): DMA straight from / to the caller's pages")
a = ap.parse_args()
base = [synth.page_numpy(a.size, a.size, index=i) for i in range(4)]
pages = [np.roll(base[i % 4], 131 * i, axis=1).copy() for i in range(a.pages)]
p = prlib_amd.make_params(prlib_amd.SAUVOLA, 31, 0.34, 0)
g = prlib_amd.geometry(p, a.size, a.size)
dst = np.zeros((a.pages, g.out_h, g.out_w), np.uint8)   # the caller's mask buffers, allocated (and touched) once and re-used by every call
if a.pinned:
    pin_in, pin_out = prlib_amd.PinnedPages(a.pages, a.size, a.size), prlib_amd.PinnedPages(a.pages, a.size - 1, a.size - 1)
    for i in range(a.pages):
        pin_in.array[i] = pages[i]
    pages, dst = list(pin_in.array), pin_out.array
out = prlib_amd.binarize_pages_host(pages[:8], p, a.devices)   # warm-up: library load, workspaces
out = prlib_amd.binarize_pages_host(pages, p, a.devices, out=dst)   # and the chunk slots of this batch shape
best = 1e9
for _ in range(a.reps):
    t0 = time.perf_counter()
    out = prlib_amd.binarize_pages_host(pages, p, a.devices, out=dst)
    best = min(best, time.perf_counter() - t0)
from oracle import capi as oc
po = oc.make_params(oc.SAUVOLA, 31, 0.34, 0)
bad = sum(int((out[i] != oc.binarize(pages[i], po)).sum()) for i in (0, a.pages // 2, a.pages - 1))
px = a.pages * out.shape[1] * out.shape[2]
print(json.dumps({"workload": f"{a.pages} x {a.size}x{a.size} u8 host pages, Sauvola w=31 k=0.34 morph=0, prl_hip_binarize_batch_host, devices={a.devices or 'all'}, {'pinned' if a.pinned else 'pageable'} caller memory",
                  "end_to_end_s": round(best, 4), "Mpx_per_s": round(px / best / 1e6, 1),
                  "host_GB_per_s": round((a.pages * a.size * a.size + px) / best / 1e9, 2),
                  "ms_per_page": round(best / a.pages * 1e3, 3), "mismatching_pixels_3_pages": bad,
                  "host_cores": os.cpu_count()}))
