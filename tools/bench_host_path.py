#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point (prl_hip_binarize_host, what the cv::Mat wrapper calls):
one 4096x4096 page per call, pageable numpy buffers in and out, output buffer reused (as a cv::Mat would be).
Never the bench `value`."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import prlib_amd
from prlib_amd import _capi, synth

PINNED = "--pinned" in sys.argv   # page and mask in prl_hip_alloc_host memory: DMA straight from / to the caller's pixels
page = synth.page_numpy(4096, 4096, index=1)
params = prlib_amd.make_params(prlib_amd.SAUVOLA, 31, 0.34, 0)
g = prlib_amd.geometry(params, 4096, 4096)
out = np.zeros((g.out_h, g.out_w), np.uint8)
L = _capi.lib()
if PINNED:
    pin_in, pin_out = prlib_amd.PinnedPages(1, 4096, 4096), prlib_amd.PinnedPages(1, g.out_h, g.out_w)
    pin_in.array[0] = page
    page, out = pin_in.array[0], pin_out.array[0]


def call():
    _capi.check(L.prl_hip_binarize_host(C.byref(params), page.ctypes.data, page.strides[0], 4096, 4096,
                                       out.ctypes.data, out.strides[0], None, 0))


call(); call()
ts = []
for _ in range(20):
    t0 = time.perf_counter(); call(); ts.append(time.perf_counter() - t0)
ts.sort()
dt = ts[len(ts) // 2]
print(json.dumps({"workload": "prl_hip_binarize_host, 1 x 4096x4096, sauvola w=31 k=0.34 morph=0, " + ("pinned" if PINNED else "pageable") + " host buffers, output reused",
                  "ms_per_page_median": round(dt * 1e3, 3), "ms_min": round(ts[0] * 1e3, 3), "Mpixels/s": round(page.size / dt / 1e6, 1),
                  "GB/s host traffic (in + out)": round((page.size + out.size) / dt / 1e9, 2)}))
