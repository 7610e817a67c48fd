#!/usr/bin/env python3
"""Real-scan batch benchmark (VERDICT r3 "next" 1): the reference's own scans (tests/golden/scans/*.npz, data only) tiled
to a resident batch of pages of the benchmark's size, beside the synthetic batch bench.py times.

Page i of the real batch is scan (i mod n_scans), repeated in both directions to cover the page and shifted by a
page-dependent offset (so the 256 pages differ); the seams of the tiling are ordinary edges.  Same call, same timing protocol
as bench.py (deferred completion, K steps, prl_hip_finish + synchronize), and the same queue statistics; three pages are
checked against the CPU oracle.  One JSON line per configuration:

    python tools/bench_real.py                      # headline: 256 x 4096^2, Sauvola w=31 k=0.34
    python tools/bench_real.py --defaults 1         # + the five header-default calls
"""
import argparse
import ctypes as C
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import prlib_amd  # noqa: E402
from prlib_amd import _capi, synth  # noqa: E402


def tiled_page(gray: np.ndarray, h: int, w: int, index: int) -> np.ndarray:
    ry, rx = -(-h // gray.shape[0]) + 1, -(-w // gray.shape[1]) + 1
    big = np.tile(gray, (ry, rx))
    oy, ox = (index * 97) % gray.shape[0], (index * 211) % gray.shape[1]
    return np.ascontiguousarray(big[oy:oy + h, ox:ox + w])


def run(pages, params, steps, warmup, dev):
    g = prlib_amd.geometry(params, pages.shape[2], pages.shape[1])
    out, _ = prlib_amd.binarizations.alloc_output(pages.shape[0], g.out_w, g.out_h, dev)
    L = _capi.lib()
    prlib_amd.set_deferred_completion(True)
    for _ in range(warmup):
        prlib_amd.binarize(pages, params, out=out)
    prlib_amd.finish(dev)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        prlib_amd.binarize(pages, params, out=out)
    prlib_amd.finish(dev)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    _capi.check(L.prl_hip_set_profiling(1))
    prlib_amd.binarize(pages, params, out=out)
    kms, cms = C.c_float(0), C.c_float(0)
    _capi.check(L.prl_hip_last_kernel_ms(C.byref(kms)))
    _capi.check(L.prl_hip_last_call_ms(C.byref(cms)))
    _capi.check(L.prl_hip_set_profiling(0))
    st = prlib_amd.last_stats()
    prlib_amd.set_deferred_completion(False)
    px = pages.shape[0] * g.out_w * g.out_h
    return out, g, {"Mpixels/s": round(px / dt / 1e6, 1), "ms_per_step": round(dt * 1e3, 4), "kernel_ms": round(kms.value, 4),
                    "call_ms": round(cms.value, 4), "refined_pixels": int(st.refined_pixels), "exact_pixels": int(st.exact_pixels),
                    "literal_pages": int(st.literal_pages), "wolf_candidates": int(st.wolf_candidates)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pages", type=int, default=256)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--defaults", type=int, default=0, help="1: also the five header-default calls")
    ap.add_argument("--only", default="", help="comma-separated configuration names (default: all selected)")
    ap.add_argument("--lib", default=None, help="A/B tooling: another build of libprlib_hip.so")
    ap.add_argument("--hooks", type=int, default=0, help="A/B tooling: 1 = libprlib_hip_testhooks.so (reads the PRL_HIP_* knobs)")
    a = ap.parse_args()
    if a.lib or a.hooks:
        _capi.use_library(a.lib or _capi.HOOKS_LIB_PATH)
    H, W = a.height or a.size, a.size
    dev = torch.device("cuda:0")
    scans = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "scans", "*.npz")))
    grays = [np.load(p)["gray"] for p in scans]
    pitch = (W + 255) // 256 * 256
    buf = torch.empty((a.pages, H, pitch), dtype=torch.uint8, device=dev)
    for i in range(a.pages):
        buf[i, :, :W] = torch.from_numpy(tiled_page(grays[i % len(grays)], H, W, i)).to(dev)
    real = buf[:, :, :W]
    syn = synth.pages_torch(a.pages, H, W, dev, seed=1000, pitch=pitch)
    cfgs = [("sauvola_headline", prlib_amd.make_params(prlib_amd.SAUVOLA, 31, 0.34, 0))]
    if a.defaults:
        cfgs += [(n + "_default", prlib_amd.default_params(m)) for n, m in
                 (("sauvola", prlib_amd.SAUVOLA), ("niblack", prlib_amd.NIBLACK), ("wolfjolion", prlib_amd.WOLFJOLION),
                  ("nick", prlib_amd.NICK), ("feng", prlib_amd.FENG))]
    from oracle import capi as oc

    if a.only:
        cfgs = [c for c in cfgs if c[0] in a.only.split(",")]
    if cfgs:   # (a throw-away run first: workspaces allocated, clocks up - the first configuration measured 25 % low without it)
        run(real, cfgs[0][1], 5, 3, dev)
    for name, p in cfgs:
        out, g, r_real = run(real, p, a.steps, a.warmup, dev)
        po = oc.make_params(p.method, p.window_size, p.k, p.morph_iterations, p.feng_alpha1, p.feng_k1, p.feng_k2, p.feng_gamma)
        idx = sorted({0, a.pages // 2, a.pages - 1})
        bad = 0
        for i in idx:
            want = oc.binarize(real[i].cpu().numpy().copy(), po)
            bad += int((want != out[i, :, : g.out_w].cpu().numpy()).sum())
        del out
        _, _, r_syn = run(syn, p, a.steps, a.warmup, dev)
        print(json.dumps({"config": name, "workload": f"{a.pages} x {W}x{H} u8 pages, method {p.method} w={p.window_size} k={p.k} "
                                                      f"morph={p.morph_iterations}",
                          "real_scans": dict(r_real, source=f"{len(grays)} scans of the reference's test_data/binarize tiled to the page size",
                                             mismatching_pixels_vs_oracle=bad, checked_pages=idx),
                          "synthetic": r_syn,
                          "real_over_synthetic": round(r_real["Mpixels/s"] / r_syn["Mpixels/s"], 4)}), flush=True)


if __name__ == "__main__":
    main()
