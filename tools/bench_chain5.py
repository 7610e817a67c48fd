#!/usr/bin/env python3
"""BASELINE config 5: deskew -> NL-means -> backgroundNormalization -> Sauvola -> Zhang-Suen thinning on a batch of A4 colour
scans, device resident.  Prints one JSON object: per-stage times (stages called one after the other through the public entry
points) and the one-call chain (prl_hip_chain_pages_device).

    python tools/bench_chain5.py --pages 1024                 one GPU
    python tools/bench_chain5.py --pages 1024 --gpus 8        the 1024-page list split over 8 ranks (one process per GPU, started
                                                              here as a child torch.distributed.run; pages are independent: no
                                                              data-path collective, RCCL only for the barrier / max of the times)
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _spawn(n):
    import socket, subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


if "--gpus" in sys.argv and "WORLD_SIZE" not in os.environ:
    _n = int(sys.argv[sys.argv.index("--gpus") + 1])
    if _n > 1:
        sys.exit(_spawn(_n))   # before anything touches a GPU

import numpy as np
import torch
import prlib_amd
from prlib_amd import dist as pdist, synth

ap = argparse.ArgumentParser()
ap.add_argument("--pages", type=int, default=1024, help="pages in the whole list (split over the ranks)")
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--width", type=int, default=2480)
ap.add_argument("--height", type=int, default=3508)
ap.add_argument("--channels", type=int, default=3)
ap.add_argument("--window", type=int, default=31)
ap.add_argument("--strength", type=float, default=10.0)
ap.add_argument("--stages", type=int, default=1, help="also time the stages one by one on the first --stage-pages pages")
ap.add_argument("--stage-pages", type=int, default=64)
ap.add_argument("--hooks", type=int, default=0, help="1: libprlib_hip_testhooks.so (reads the PRL_HIP_* knobs, e.g. PRL_HIP_CHAIN_PASS)")
ap.add_argument("--repeat", type=int, default=1, help="run the one-call chain this many times (the first call allocates the workspaces)")
ap.add_argument("--host", type=int, default=0, help="also time prl_hip_chain_batch_host on the same pages in host memory (end to end)")
ap.add_argument("--check-pages", type=int, default=0, help="pages compared with the composed CPU oracle (slow: NL-means on the host)")
ap.add_argument("--cpu-pages", type=int, default=-1, help="pages of the CPU baseline leg: the composed oracle chain on that many of the same pages, "
                "run side by side on all host cores (-1: one page per 32 cores, at least 2; 0: skip)")
a = ap.parse_args()
if a.hooks:
    prlib_amd._capi.use_library(prlib_amd._capi.HOOKS_LIB_PATH)
world, rank, local_rank = pdist.init()
if world != a.gpus:
    raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
dev = torch.device("cuda", torch.cuda.current_device())
mine = pdist.page_range(a.pages, world, rank)
total_pages = a.pages
a.pages = len(mine)
t0 = time.perf_counter()
pages, skews = synth.text_pages_torch(a.pages, a.height, a.width, dev, seed=7000 + mine.start, channels=a.channels)
torch.cuda.synchronize()
gen_s = time.perf_counter() - t0
px_in = a.pages * a.width * a.height


def timed(fn, reps=1):
    best = None
    for _ in range(reps):
        r = None
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    return best, r


res = {"workload": f"{a.pages} x {a.width}x{a.height}x{a.channels} synthetic text scans (skew +-4 deg, shaded), 1 GPU, device resident; "
                   f"deskew -> denoise({a.strength}) -> backgroundNormalization -> Sauvola w={a.window} k=0.34 morph=0 -> Zhang-Suen",
       "generate_s": round(gen_s, 2)}
if a.stages and world == 1:
    n = min(a.stage_pages, a.pages)
    sub = pages[:n]
    t_desk, (outs, ang) = timed(lambda: prlib_amd.deskew(sub), 2)
    sq = torch.stack([o for o in outs if o.shape[0] == o.shape[1] == max(a.width, a.height)]) if any(o.shape[0] == o.shape[1] for o in outs) else sub
    m = sq.shape[0]
    px_sq = m * sq.shape[1] * sq.shape[2]
    t_den, den = timed(lambda: prlib_amd.denoise(sq, a.strength), 2)
    t_bg, bg = timed(lambda: prlib_amd.backgroundNormalization(den), 2)
    t_gray, g8 = timed(lambda: prlib_amd.cvtColorBGR2GRAY(bg), 2)
    t_bin, mask = timed(lambda: prlib_amd.binarizeSauvola(g8, a.window, 0.34, 0), 2)
    t_inv, inv = timed(lambda: prlib_amd.bitwise_not(mask), 2)
    t_thin, sk = timed(lambda: prlib_amd.thinZhangSuen(inv), 2)
    res["stages"] = {"pages": n, "rotated_pages": m, "deskew_ms": round(t_desk * 1e3, 1),
                     "deskew_Mpx_s": round(n * a.width * a.height / t_desk / 1e6, 1),
                     "denoise_ms": round(t_den * 1e3, 1), "denoise_Mpx_s": round(px_sq / t_den / 1e6, 1),
                     "bgnorm_ms": round(t_bg * 1e3, 2), "bgnorm_Mpx_s": round(px_sq / t_bg / 1e6, 1),
                     "bgr2gray_ms": round(t_gray * 1e3, 2), "sauvola_ms": round(t_bin * 1e3, 2),
                     "invert_ms": round(t_inv * 1e3, 2), "thin_ms": round(t_thin * 1e3, 2),
                     "angle_abs_err_deg_mean": round(float(np.abs(ang - skews[:n]).mean()), 3)}
    del outs, sq, den, bg, g8, mask, inv, sk
    torch.cuda.empty_cache()
pdist.barrier()
t_chain, (outs, angles) = timed(lambda: prlib_amd.process_pages(pages, a.channels, prlib_amd.SAUVOLA, a.window, 0.34, 0,
                                                                denoise_strength=a.strength, thin=0, deskew=True,
                                                                background_normalization=True))
t_first = t_chain
for _ in range(a.repeat - 1):
    del outs
    t2, (outs, angles) = timed(lambda: prlib_amd.process_pages(pages, a.channels, prlib_amd.SAUVOLA, a.window, 0.34, 0,
                                                               denoise_strength=a.strength, thin=0, deskew=True,
                                                               background_normalization=True))
    t_chain = min(t_chain, t2)
t_chain = pdist.max_over_ranks(t_chain, device=dev)   # the job is done when the slowest rank is
px_in = total_pages * a.width * a.height
res.update({"n_gpus": world, "pages_total": total_pages, "pages_this_rank": a.pages,
            "chain_one_call_s": round(t_chain, 3), "chain_first_call_s": round(t_first, 3), "chain_input_Mpx_s": round(px_in / t_chain / 1e6, 1),
            "pages_per_s": round(total_pages / t_chain, 2),
            "rotated_pages": int(sum(1 for o in outs if o.shape[0] == o.shape[1])),
            "angle_abs_err_deg_mean": round(float(np.abs(angles - skews).mean()), 3),
            "skeleton_fraction": round(float(np.mean([float((o > 0).float().mean()) for o in outs[:8]])), 5)})
if a.host and world == 1:
    host_pages = [pg for pg in pages.cpu().numpy()]
    t_host = None
    for _ in range(max(1, a.repeat)):
        t0 = time.perf_counter()
        h_out, h_ang = prlib_amd.process_pages_host(host_pages, prlib_amd.SAUVOLA, a.window, 0.34, 0, denoise_strength=a.strength, thin=0,
                                                    deskew=True, background_normalization=True, n_devices=1)
        dt = time.perf_counter() - t0
        t_host = dt if t_host is None else min(t_host, dt)
    same = bool(np.array_equal(h_ang, angles)) and all(np.array_equal(h_out[i], outs[i].cpu().numpy()) for i in range(0, a.pages, max(1, a.pages // 8)))
    res["host_pages_end_to_end"] = {"s": round(t_host, 3), "pages_per_s": round(a.pages / t_host, 2),
                                    "input_Mpx_s": round(px_in / t_host / 1e6, 1), "equals_device_entry": same}
if a.cpu_pages != 0 and world == 1:
    # CPU baseline beside the number: the oracle's stages composed on the host, several pages side by side so
    # that every core is busy (the Hough transform and the thinning are single-threaded per page, NL-means takes the threads
    # it is given); pages per second over the wall time of the sample
    from concurrent.futures import ThreadPoolExecutor
    from oracle import capi as oc
    cores = os.cpu_count() or 1
    n_cpu = a.cpu_pages if a.cpu_pages > 0 else max(2, cores // 32)
    n_cpu = min(n_cpu, a.pages)
    per = max(1, cores // n_cpu)
    sample = [np.ascontiguousarray(pages[i].cpu().numpy()) for i in np.linspace(0, a.pages - 1, n_cpu).round().astype(int)]
    def one(pg):
        cur, info = oc.deskew(pg)
        cur = oc.denoise(np.ascontiguousarray(cur), a.strength, threads=per)
        cur = oc.bgnorm(np.ascontiguousarray(cur))
        cur = oc.bgr2gray(np.ascontiguousarray(cur)) if cur.ndim == 3 else cur
        mask = oc.binarize(np.ascontiguousarray(cur), oc.make_params(oc.SAUVOLA, a.window, 0.34, 0))
        return oc.thin(255 - mask, 0).shape
    t0 = time.perf_counter()
    with ThreadPoolExecutor(n_cpu) as ex:
        list(ex.map(one, sample))
    tc = time.perf_counter() - t0
    res["cpu_baseline"] = {"value": round(n_cpu / tc, 3), "unit": "pages/s", "cores": cores, "kind": "port",
                           "sample": f"{n_cpu} of the benchmark's pages through the composed oracle chain, side by side, {per} OpenMP threads each for NL-means, {tc:.1f} s"}
    res["speedup_vs_cpu_baseline"] = round(res["pages_per_s"] / max(res["cpu_baseline"]["value"], 1e-9), 1)
if a.check_pages:
    from oracle import capi as oc
    bad = 0
    for i in np.linspace(0, a.pages - 1, a.check_pages).round().astype(int):
        cur, info = oc.deskew(np.ascontiguousarray(pages[i].cpu().numpy()))
        cur = oc.denoise(np.ascontiguousarray(cur), a.strength, threads=os.cpu_count() or 1)
        cur = oc.bgnorm(np.ascontiguousarray(cur))
        cur = oc.bgr2gray(np.ascontiguousarray(cur)) if cur.ndim == 3 else cur
        mask = oc.binarize(np.ascontiguousarray(cur), oc.make_params(oc.SAUVOLA, a.window, 0.34, 0))
        want = oc.thin(255 - mask, 0)
        got = outs[i].cpu().numpy()
        bad += int(got.shape != want.shape or (got != want).sum() or info["angle"] != angles[i])
    res["parity"] = {"checked_pages": int(a.check_pages), "pages_with_any_difference": bad}
if rank == 0:
    print(json.dumps(res))
pdist.finish()
