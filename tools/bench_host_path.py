#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point (prl_hip_binarize_host, what the cv::Mat wrapper calls):
one 4096x4096 page per call, pageable numpy buffers in and out.  Never the bench `value`."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import prlib_amd
from prlib_amd import synth

page = synth.page_numpy(4096, 4096, index=1)
prlib_amd.binarizeSauvola(page, 31, 0.34, 0)
n = 10
t0 = time.perf_counter()
for _ in range(n):
    out = prlib_amd.binarizeSauvola(page, 31, 0.34, 0)
dt = (time.perf_counter() - t0) / n
print(json.dumps({"workload": "prl_hip_binarize_host, 1 x 4096x4096, sauvola w=31 k=0.34 morph=0, pageable host buffers",
                  "ms_per_page": round(dt * 1e3, 3), "Mpixels/s": round(page.size / dt / 1e6, 1),
                  "GB/s_over_PCIe (in + out + padded copy back)": round((page.size + out.size) / dt / 1e9, 2)}))
