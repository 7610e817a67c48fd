#!/usr/bin/env python3
"""Real-scan batch benchmark (VERDICT r3 "next" 1): the reference's own scans (tests/golden/scans/*.npz, data only) tiled
to a resident batch of pages of the benchmark's size, beside the synthetic batch bench.py times.

Page i of the real batch is scan (i mod n_scans), repeated in both directions to cover the page and shifted by a
page-dependent offset (so the 256 pages differ); the seams of the tiling are ordinary edges.  Same call, same timing protocol
as bench.py (deferred completion, K steps, prl_hip_finish + synchronize), and the same queue statistics; three pages are
checked against the CPU oracle.  One JSON line per configuration:

    python tools/bench_real.py                      # headline: 256 x 4096^2, Sauvola w=31 k=0.34
    python tools/bench_real.py --defaults 1         # + the five header-default calls
    python tools/bench_real.py --chain 1            # config 5 on the reference's colour scans tiled to 256 A4 pages
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


def tiled_colour_page(bgr: np.ndarray, h: int, w: int, index: int) -> np.ndarray:
    ry, rx = -(-h // bgr.shape[0]) + 1, -(-w // bgr.shape[1]) + 1
    big = np.tile(bgr, (ry, rx, 1))
    oy, ox = (index * 97) % bgr.shape[0], (index * 211) % bgr.shape[1]
    return np.ascontiguousarray(big[oy:oy + h, ox:ox + w])


def chain_stage_times(pages, window, strength):
    """BASELINE config 5 stage by stage through the public entry points (device tensors in between), then as the one call."""

    def timed(fn, reps=2):
        best, r = None, None
        for _ in range(reps):
            r = None
            torch.cuda.synchronize()
            t = time.perf_counter()
            r = fn()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
            best = dt if best is None else min(best, dt)
        return best, r

    n, h, w = pages.shape[0], pages.shape[1], pages.shape[2]
    prlib_amd.deskew_stats(reset=True)
    t_desk, (outs, ang) = timed(lambda: prlib_amd.deskew(pages), 2)
    hough = prlib_amd.deskew_stats().as_dict()
    side = max(h, w)
    rot = [o for o in outs if o.shape[0] == side and o.shape[1] == side]
    sq = torch.stack(rot) if rot else pages
    del outs, rot
    m, px_sq = sq.shape[0], sq.shape[0] * sq.shape[1] * sq.shape[2]
    t_den, den = timed(lambda: prlib_amd.denoise(sq, strength))
    t_bg, bg = timed(lambda: prlib_amd.backgroundNormalization(den))
    t_gray, g8 = timed(lambda: prlib_amd.cvtColorBGR2GRAY(bg))
    t_bin, mask = timed(lambda: prlib_amd.binarizeSauvola(g8, window, 0.34, 0))
    st = prlib_amd.last_stats()
    t_inv, inv = timed(lambda: prlib_amd.bitwise_not(mask))
    t_thin, sk = timed(lambda: prlib_amd.thinZhangSuen(inv))
    res = {"pages": n, "rotated_pages": m, "stage_pixels": px_sq,
           "deskew_ms": round(t_desk * 1e3, 2), "denoise_ms": round(t_den * 1e3, 2), "bgnorm_ms": round(t_bg * 1e3, 3),
           "bgr2gray_ms": round(t_gray * 1e3, 3), "sauvola_ms": round(t_bin * 1e3, 3), "invert_ms": round(t_inv * 1e3, 3),
           "thin_ms": round(t_thin * 1e3, 3),
           # per page (deskew: per input page; the later stages per page they ran on) so that sets with different numbers of
           # rotated pages compare
           "per_page_ms": {"deskew": round(t_desk * 1e3 / n, 4), "denoise": round(t_den * 1e3 / m, 4), "bgnorm": round(t_bg * 1e3 / m, 5),
                           "bgr2gray": round(t_gray * 1e3 / m, 5), "sauvola": round(t_bin * 1e3 / m, 5), "invert": round(t_inv * 1e3 / m, 5),
                           "thin": round(t_thin * 1e3 / m, 4)},
           "hough": hough, "hough_segments_per_page": round(hough["segments"] / max(1, hough["pages"]), 1),
           "hough_points_per_page": round(hough["points"] / max(1, hough["pages"]), 1),
           "sauvola_refined_pixels": int(st.refined_pixels), "sauvola_literal_pages": int(st.literal_pages),
           "skeleton_fraction": round(float((sk > 0).float().mean()), 5),
           "angles_nonzero": int((np.asarray(ang) != 0).sum())}
    del den, bg, g8, mask, inv, sk, sq
    torch.cuda.empty_cache()
    return res


def chain_main(a):
    """--chain 1: the reference's colour scans (tests/golden/stages/chain_*.npz) tiled to A4 colour pages, beside the synthetic
    text scans tools/bench_chain5.py uses: the stages one by one on --stage-pages pages and the one-call chain on --pages."""
    dev = torch.device("cuda:0")
    H, W = a.height or 3508, a.size if a.size != 4096 else 2480
    files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "stages", "chain_*.npz")))
    scans = [np.load(p)["bgr"] for p in files]
    real = torch.empty((a.pages, H, W, 3), dtype=torch.uint8, device=dev)
    for i in range(a.pages):
        real[i] = torch.from_numpy(tiled_colour_page(scans[i % len(scans)], H, W, i)).to(dev)
    syn, _ = synth.text_pages_torch(a.pages, H, W, dev, seed=7000, channels=3)
    out = {"workload": f"{a.pages} x {W}x{H}x3 pages, deskew -> denoise({a.strength}) -> backgroundNormalization -> Sauvola w={a.window} "
                       f"k=0.34 morph=0 -> Zhang-Suen; stages on the first {min(a.stage_pages, a.pages)} pages",
           "real_source": f"{len(scans)} colour scans of the reference's test_data/binarize tiled to the page size"}
    ns = min(a.stage_pages, a.pages)
    for tag, pages in (("real_scans", real), ("synthetic", syn)):
        chain_stage_times(pages[:min(8, ns)], a.window, a.strength)   # workspaces, clocks
        stages = chain_stage_times(pages[:ns], a.window, a.strength)
        best = None
        for _ in range(max(2, a.repeat)):
            torch.cuda.synchronize()
            t = time.perf_counter()
            outs, angles = prlib_amd.process_pages(pages, 3, prlib_amd.SAUVOLA, a.window, 0.34, 0, denoise_strength=a.strength,
                                                   thin=0, deskew=True, background_normalization=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
            best = dt if best is None else min(best, dt)
            rotated = int(sum(1 for o in outs if o.shape[0] == o.shape[1]))
            del outs
        out[tag] = {"stages": stages, "chain_one_call_s": round(best, 4), "pages_per_s": round(a.pages / best, 2),
                    "chain_input_Mpx_s": round(a.pages * H * W / best / 1e6, 1), "rotated_pages": rotated}
    r, s = out["real_scans"], out["synthetic"]
    out["real_over_synthetic"] = {k: round(r["stages"]["per_page_ms"][k] / s["stages"]["per_page_ms"][k], 3) for k in r["stages"]["per_page_ms"]}
    out["real_over_synthetic"]["chain_per_page"] = round(r["chain_one_call_s"] / s["chain_one_call_s"], 3)
    # three real pages against the composed CPU oracle (NL-means on the host: slow, so few)
    if a.check_pages:
        from oracle import capi as oc
        bad = 0
        idx = sorted({0, a.pages // 2, a.pages - 1})[:a.check_pages]
        outs, angles = prlib_amd.process_pages(real[idx], 3, prlib_amd.SAUVOLA, a.window, 0.34, 0, denoise_strength=a.strength, thin=0,
                                               deskew=True, background_normalization=True)
        for j, i in enumerate(idx):
            cur, info = oc.deskew(real[i].cpu().numpy())
            cur = oc.denoise(np.ascontiguousarray(cur), a.strength, threads=os.cpu_count() or 8)
            cur = oc.bgr2gray(np.ascontiguousarray(oc.bgnorm(np.ascontiguousarray(cur))))
            want = oc.thin(255 - oc.binarize(np.ascontiguousarray(cur), oc.make_params(oc.SAUVOLA, a.window, 0.34, 0)), 0)
            got = outs[j].cpu().numpy()
            bad += int(got.shape != want.shape or angles[j] != info["angle"] or (got != want).sum())
        out["real_scans"]["checked_pages"] = idx
        out["real_scans"]["mismatches_vs_oracle"] = bad
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chain", type=int, default=0, help="1: BASELINE config 5 on the reference's colour scans tiled to A4 pages, beside the synthetic scans")
    ap.add_argument("--stage-pages", type=int, default=64)
    ap.add_argument("--window", type=int, default=31)
    ap.add_argument("--strength", type=float, default=10.0)
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--check-pages", type=int, default=0)
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
    if a.chain:
        return chain_main(a)
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
