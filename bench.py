#!/usr/bin/env python3
"""Headline benchmark: Mpixels/s of Sauvola w=31 on a batch of 4K pages (BASELINE.json metric).

One "step" = one pass of prl_hip_binarize_batch_device over the whole resident batch (256 pages of
4096x4096 u8 per GPU by default).  Pages are generated on the device before the timed region; every
rank owns its own batch (independent pages => no data-path collective; weak scaling).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (see README/DESIGN.md for the fields).  `roofline.achieved` is the
algorithmic byte count of one launch (1 B read + 1 B written per output pixel) over the average
duration of the dominant kernel (k_fused), measured with HIP events inside the library on the
stream the kernel runs on.  `cpu_baseline` times the CPU oracle (oracle/, the literal restatement of
the reference) on a bounded sample of the same pages on this box's host cores; it is a reported
baseline, never part of the measured path.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pages", type=int, default=256, help="pages per GPU")
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--method", default="sauvola")
    ap.add_argument("--window", type=int, default=31)
    ap.add_argument("--k", type=float, default=0.34)
    ap.add_argument("--morph", type=int, default=0)
    ap.add_argument("--mode", default="auto", choices=["auto", "literal"])
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--check-pages", type=int, default=8,
                    help="pages verified against the oracle after timing (spread over the batch: first, last, evenly between)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong", "both"],
                    help="weak: --pages per GPU; strong: --pages in total, split over the ranks by dist.page_range; both (only "
                         "with --gpus > 1): the weak line is THE line, the strong-scaling measurement rides in it as `strong`")
    ap.add_argument("--traffic", type=int, default=1,
                    help="1: measure roofline.traffic in this run - two child passes of this benchmark under rocprofv3 --pmc "
                         "(FETCH_SIZE, WRITE_SIZE), started before this process touches the GPU (N=1 only); 0: traffic = null")
    ap.add_argument("--ceilings", type=int, default=1, help="1: time the hand-written read / write / copy kernels on this batch")
    ap.add_argument("--worst-case", type=int, default=1,
                    help="1 (N=1 only): after the timed region, the same pages through the literal pipeline and a batch of adversarial "
                         "pages (every second pixel inside the float32 decision band) through auto mode -> `worst_case`")
    ap.add_argument("--adversarial-pages", type=int, default=0, help="pages of the adversarial batch (0: as many as the headline's batch)")
    ap.add_argument("--end-to-end", type=int, default=1,
                    help="1 (N=1 only): the batch as a host page list in pinned memory through prl_hip_binarize_batch_host "
                         "(H2D + kernels + D2H) -> `end_to_end` (SURVEY.md 8d's second number; never `value`)")
    ap.add_argument("--digest", type=int, default=0,
                    help="1: the line carries `page_digests`, the CRC-32 of every page's mask in page-list order (all ranks) - "
                         "lets a test compare an N-rank run with the single-process run of the same page list")
    ap.add_argument("--lib", default=None, help="A/B tooling: load this build of libprlib_hip.so instead of the in-tree one")
    ap.add_argument("--hooks", type=int, default=0, help="A/B tooling: 1 = load libprlib_hip_testhooks.so (reads the PRL_HIP_* tuning knobs)")
    return ap.parse_args()


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks with torch.distributed.run as a CHILD process and hand
    its exit code back.  Runs before this process has imported torch or touched a GPU (a process that has initialised
    the GPU must never exec another program; children are fresh processes)."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def measure_traffic(args):
    """HBM-side bytes of one k_fused launch, measured NOW on this box (MI355X_MICROARCH.md, HBM / rocprofv3 section): two child
    runs of this same benchmark under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (the TCC slots do not hold both),
    counters only, the program itself after `--`.  gfx950: FETCH_SIZE tallies 128-B requests as 64 B (x2), WRITE_SIZE is
    exact; both are in KiB.  Called before this process has imported torch or touched the GPU.  -> (bytes or None, note)"""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None, "rocprofv3 not found", {"shader_clock_ghz": None, "gui_active_cycles_per_launch": None, "clock_note": "rocprofv3 not found"}
    vals = {}
    clock = {"shader_clock_ghz": None, "gui_active_cycles_per_launch": None, "clock_note": None}
    for counter in ("FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE"):
        tmp = tempfile.mkdtemp(prefix="prl_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", tmp, "--", sys.executable, os.path.abspath(__file__),
               "--gpus", "1", "--steps", "3", "--warmup", "1", "--pages", str(args.pages), "--size", str(args.size),
               "--height", str(args.height), "--method", args.method, "--window", str(args.window), "--k", str(args.k),
               "--morph", str(args.morph), "--mode", args.mode, "--cpu-seconds", "0", "--check-pages", "0", "--traffic", "0",
               "--ceilings", "0", "--worst-case", "0", "--end-to-end", "0"] + (["--lib", args.lib] if args.lib else []) + (["--hooks", "1"] if args.hooks else [])
        env = dict(os.environ, TMPDIR="/tmp")
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
            per, dur = [], []
            for f in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row.get("Counter_Name") == counter and "k_fused" in row.get("Kernel_Name", ""):
                            per.append(float(row["Counter_Value"]))
                            try:
                                dur.append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
                            except (KeyError, ValueError):
                                pass
            if counter == "GRBM_GUI_ACTIVE":
                # The clock the kernel really ran at (boxes of the pool differ by > 10 %: power-capped clocks): GRBM_GUI_ACTIVE
                # sums the busy cycles of the 8 XCDs; over the same dispatches' own duration in this pass.
                if per and len(dur) == len(per) and sum(dur) > 0:
                    clock["gui_active_cycles_per_launch"] = round(sum(per) / len(per) / 8.0)
                    clock["shader_clock_ghz"] = round(sum(per) / 8.0 / sum(dur), 4)
                    clock["clock_note"] = "GRBM_GUI_ACTIVE / 8 XCDs over the dispatches' duration, rocprofv3 --pmc child pass of this run"
                else:
                    clock["clock_note"] = f"GRBM_GUI_ACTIVE: no usable k_fused rows (rc {r.returncode})"
                continue
            if not per:
                return None, f"{counter}: no k_fused rows (rc {r.returncode}): {(r.stderr or '')[-200:]}", clock
            vals[counter] = sum(per) / len(per)
        except Exception as e:  # (timeout, profiler refused)
            if counter == "GRBM_GUI_ACTIVE":
                clock["clock_note"] = f"GRBM_GUI_ACTIVE: {e!r}"
                continue
            return None, f"{counter}: {e!r}", clock
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return (int((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0), "2 x FETCH_SIZE + WRITE_SIZE (KiB), rocprofv3 --pmc, separate passes, this run",
            clock)


def simd_cycles_per_row(clock, args, W, g, n_pages):
    """Shader cycles of one k_fused launch x the chip's 1024 SIMDs / the wavefront-rows of the launch (pages x output rows x strips of
    512 padded columns; the strip count is the library's own: a host-side helper of the test-hooks build)."""
    if not clock.get("gui_active_cycles_per_launch"):
        return None
    try:
        from prlib_amd import _capi as _c
        LH = C.CDLL(_c.HOOKS_LIB_PATH)
        uo = (C.c_int * 2)()
        LH.prl_hip_internal_strip_layout.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_int)]
        from prlib_amd import binarizations as _b
        strips = LH.prl_hip_internal_strip_layout(_b.METHODS[args.method], g.w, W, g.out_w, 1 if args.morph else 0, uo)
        if strips <= 0:
            return None
        return round(clock["gui_active_cycles_per_launch"] * 1024.0 / (n_pages * g.out_h * strips), 1)
    except Exception:
        return None


def cpu_baseline(pages_host, params_oracle, budget_s):
    """Time the CPU oracle (kind 'port') on a bounded sample of the same pages, all host cores."""
    from oracle import capi as oc

    cores = os.cpu_count() or 1
    n_avail, h, w = pages_host.shape
    t0 = time.perf_counter()
    out1 = oc.binarize_batch(pages_host[:1], params_oracle, threads=1)
    t1 = time.perf_counter() - t0
    px_page = out1.shape[1] * out1.shape[2]
    # pages for roughly budget_s seconds with `cores` threads (assume ~linear up to memory bandwidth)
    n = int(max(cores, min(n_avail, (budget_s / max(t1, 1e-3)) * max(1, cores // 2))))
    n = min(n, n_avail)
    # the sample is passed over as many times as fit the budget (about 10 s of wall time on a many-core host)
    reps, tn = 0, 0.0
    while reps == 0 or (tn < min(budget_s, 10.0) and reps < 8):
        t0 = time.perf_counter()
        oc.binarize_batch(pages_host[:n], params_oracle, threads=cores)
        tn += time.perf_counter() - t0
        reps += 1
    return {
        "value": round(reps * n * px_page / tn / 1e6, 2),
        "unit": "Mpixels/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{n} of the benchmark's pages ({w}x{h}) x {reps} passes, oracle/prl_oracle.c with {cores} OpenMP threads, "
                  f"{tn:.1f} s; single-thread 1 page: {px_page / t1 / 1e6:.2f} Mpixels/s",
    }


def adversarial_stripes(method, w, k, eps_hint):
    """Two gray levels (a, b) for a page of vertical stripes of period 2 (a on even columns): every interior (w-1)x(w-1) window
    holds (w-1)/2 columns of each, so every interior pixel sees the same sums, and for the pixels on the a-columns the
    exact-arithmetic threshold T* lies as close to a - 0.5 as any integer pair allows - inside the float32 decision band, so
    the fast test settles none of them.  Closed forms of SURVEY.md A.1 / A.2 / A.4; None for the other methods."""
    import numpy as np

    n = (w - 1) * (w - 1) // 2                       # pixels of each level in a window ((w-1) even)
    f = 1.0 / (w * w)
    a, b = np.meshgrid(np.arange(1, 256, dtype=np.float64), np.arange(0, 256, dtype=np.float64), indexing="ij")
    m = f * n * (a + b)
    q = f * n * (a * a + b * b)
    s = np.sqrt(np.maximum(q - m * m, 0.0))
    if method == 0:
        t = m * (s * (k / 128.0) + (1.0 - k))
    elif method == 1:
        t = s * k + m
    elif method == 3:
        t = m + k * np.sqrt(q)
    else:
        return None
    d = np.abs(t - (a - 0.5))
    d[a == b] = np.inf                               # (a flat page is a different experiment)
    i = np.unravel_index(np.argmin(d), d.shape)
    return int(a[i]), int(b[i]), float(d[i])


def _time_steps(prlib_amd, dev, fn, reps):
    import torch

    fn()
    prlib_amd.finish(dev)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    prlib_amd.finish(dev)
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / reps


def worst_case_legs(args, prlib_amd, _capi, L, dev, pages, out, params, method, g, W, H):
    """What the same call costs when the input defeats the fast path.  (a) `literal`: the benchmark's own pages through
    PRL_MODE_LITERAL (float64 integral planes, 16 B per padded pixel, in chunks) - what every page costs that overflows a
    queue.  (b) `adversarial`: pages of two-level stripes chosen so that half of all pixels sit inside the float32 decision
    band (adversarial_stripes): the threshold sweep queues them, the queue overflows, the page is flagged and redone by the exact sweep
    (k_fused_exact: integer sums, the float64 interval test inline; until round 5: by the
    literal pipeline - auto mode's worst case = fused pass + literal pass + the host round trip per flagged page."""
    import numpy as np
    import torch

    res = {}
    px_page = g.out_w * g.out_h
    n_lit = min(pages.shape[0], 32)
    prlib_amd.set_exec_mode(1)
    try:
        t = _time_steps(prlib_amd, dev, lambda: prlib_amd.binarize(pages[:n_lit], params, out=out[:n_lit]), 2)
    finally:
        prlib_amd.set_exec_mode(0)
    res["literal"] = {"pages": int(n_lit), "ms_per_step": round(t * 1e3, 3), "value": round(n_lit * px_page / t / 1e6, 1),
                      "unit": "Mpixels/s", "note": "PRL_MODE_LITERAL on the first pages of the benchmark batch"}
    adv = adversarial_stripes(method, g.w, args.k, None)
    if adv is None:
        res["adversarial"] = None
        return res
    a, b, margin = adv
    b8 = np.zeros(8, np.float64)
    eps1 = None
    from prlib_amd import _capi as _c
    if os.path.exists(_c.HOOKS_LIB_PATH):   # the decision margin is a host-side helper of the test-hooks build (not part of the ABI)
        LH = C.CDLL(_c.HOOKS_LIB_PATH)
        LH.prl_hip_internal_fused_bounds.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        if LH.prl_hip_internal_fused_bounds(C.byref(params), W, H, b8.ctypes.data) == 0:
            eps1 = float(b8[5])
    n_adv = min(pages.shape[0], args.adversarial_pages) if args.adversarial_pages > 0 else pages.shape[0]   # (default: the headline's batch)
    row = torch.tensor([a, b], dtype=torch.uint8, device=dev).repeat((pages.shape[2] + 1) // 2)[: pages.shape[2]]
    adv_pages = row.expand(n_adv, pages.shape[1], pages.shape[2]).contiguous()
    t = _time_steps(prlib_amd, dev, lambda: prlib_amd.binarize(adv_pages, params, out=out[:n_adv]), 2)
    st = prlib_amd.last_stats()
    from oracle import capi as oc

    want = oc.binarize(adv_pages[0, :, :W].cpu().numpy().copy(), oc.make_params(method, args.window, args.k, args.morph))
    bad = int((want != out[0, :, : g.out_w].cpu().numpy()).sum())
    res["adversarial"] = {
        "pages": int(n_adv), "ms_per_step": round(t * 1e3, 3), "value": round(n_adv * px_page / t / 1e6, 1), "unit": "Mpixels/s",
        "pattern": f"vertical stripes of period 2, levels {a} / {b}: T* - (p - 0.5) = {margin:.2e} on every interior pixel of the "
                   f"{a}-columns (decision band eps1 = {eps1})",
        "exact_sweep_pages": int(st.exact_sweep_pages), "literal_pages": int(st.literal_pages), "refined_pixels": int(st.refined_pixels),
        "exact_pixels": int(st.exact_pixels), "mismatching_pixels_page0": bad,
    }
    return res


def end_to_end_leg(prlib_amd, dev, pages, params, g, W, H):
    """SURVEY.md 8d's second number: the same batch as a host page list (pinned memory: the DMA engines read and write the
    caller's pages directly) through prl_hip_binarize_batch_host on this one GPU - H2D + kernels + D2H."""
    n = int(pages.shape[0])
    try:
        pin_in, pin_out = prlib_amd.PinnedPages(n, H, W), prlib_amd.PinnedPages(n, g.out_h, g.out_w)
    except Exception as e:   # (not enough lockable host memory on this box)
        return {"value": None, "note": f"pinned allocation failed: {e!r}"}
    try:
        for i0 in range(0, n, 32):
            pin_in.array[i0:i0 + 32] = pages[i0:i0 + 32, :, :W].cpu().numpy()
        host_pages = list(pin_in.array)
        prlib_amd.binarize_pages_host(host_pages[:8], params, 1, out=pin_out.array[:8])   # (workspaces, chunk slots)
        prlib_amd.binarize_pages_host(host_pages, params, 1, out=pin_out.array)
        best = 1e30
        for _ in range(3):
            t0 = time.perf_counter()
            prlib_amd.binarize_pages_host(host_pages, params, 1, out=pin_out.array)
            best = min(best, time.perf_counter() - t0)
        px = n * g.out_w * g.out_h
        return {"value": round(px / best / 1e6, 1), "unit": "Mpixels/s", "seconds": round(best, 4),
                "host_gb_per_s": round((n * H * W + px) / best / 1e9, 2), "pages": n,
                "note": "prl_hip_binarize_batch_host, pages and masks in pinned host memory, one GPU, best of 3; PCIe-bound, never `value`"}
    finally:
        pin_in.close()
        pin_out.close()


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    if args.scaling == "both" and args.gpus == 1:
        args.scaling = "weak"
    elif args.scaling == "weak" and args.gpus > 1:
        args.scaling = "both"   # N > 1: the weak-scaling line carries the strong-scaling measurement as well (`strong`)
    traffic, traffic_note = None, "not measured (--traffic 0 or N > 1)"
    clock = {"shader_clock_ghz": None, "gui_active_cycles_per_launch": None, "clock_note": "not measured (--traffic 0 or N > 1)"}
    under_profiler = any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB"))
    if under_profiler:
        # a profiler's preloaded library has initialised the GPU in this process already: starting the counter passes from
        # here would be an exec() from a GPU-initialised process, which the pool refuses
        traffic_note = "not measured: this run is itself under a profiler"
    elif args.traffic and args.gpus == 1 and args.mode == "auto" and os.environ.get("PRL_BENCH_DRYRUN") != "1":
        traffic, traffic_note, clock = measure_traffic(args)   # child processes, before anything here touches the GPU
    import torch

    if args.lib or args.hooks:
        from prlib_amd import _capi as _capi0

        _capi0.use_library(args.lib or _capi0.HOOKS_LIB_PATH)

    from prlib_amd import dist as pdist

    world, rank, local_rank = pdist.init()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if os.environ.get("PRL_BENCH_DRYRUN") == "1":
        # launcher/collective plumbing only (CPU test of the N>1 path): no kernels, no number
        total_pages = args.pages if args.scaling == "strong" else world * args.pages
        mine = pdist.page_range(total_pages, world, rank)
        pdist.barrier()
        slowest = pdist.max_over_ranks(float(rank + 1))
        total = pdist.sum_over_ranks(float(len(mine)))
        blocks = pdist.gather_lists([[mine.start, mine.stop]])   # every rank's block, in rank order
        if rank == 0:
            print(json.dumps({"dryrun": True, "n_gpus": world, "pages_total": int(total), "max_rank_plus_1": slowest,
                              "first_block": [mine.start, mine.stop], "blocks": blocks,
                              "scaling": "weak" if args.scaling == "both" else args.scaling,
                              "strong_ride_along": args.scaling == "both"}), flush=True)
        pdist.finish()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU path in prlib_amd)")
    dev = torch.device("cuda", torch.cuda.current_device())  # pdist.init() bound this process to its GPU

    import prlib_amd
    from prlib_amd import _capi, synth

    H = args.height or args.size
    W = args.size
    method = prlib_amd.binarizations.METHODS[args.method]
    params = prlib_amd.make_params(method, args.window, args.k, args.morph)
    g = prlib_amd.geometry(params, W, H)
    prlib_amd.set_exec_mode(1 if args.mode == "literal" else 0)

    # synthetic pages, resident in HBM before the timed region.  The job's page list has
    # world * pages_per_gpu pages (weak scaling); this rank owns a contiguous block of it.
    # (--scaling strong: the list has args.pages pages in all and the ranks split it)
    total_pages = args.pages if args.scaling == "strong" else world * args.pages
    mine = pdist.page_range(total_pages, world, rank)
    n_mine = len(mine)
    if n_mine == 0:
        raise SystemExit(f"rank {rank}: no pages to process ({total_pages} pages over {world} ranks)")
    pitch = (W + 255) // 256 * 256
    pages = synth.pages_torch(n_mine, H, W, dev, seed=1000 + mine.start, pitch=pitch)
    out, out_pitch = prlib_amd.binarizations.alloc_output(n_mine, g.out_w, g.out_h, dev)
    L = _capi.lib()

    def step():
        prlib_amd.binarize(pages, params, out=out)

    # the device entry point only enqueues (deferred completion): the K steps of the timed region pipeline on the stream
    # and prl_hip_finish - flag check of every call + stream wait - closes the region on every rank
    prlib_amd.set_deferred_completion(True)

    def barrier():
        prlib_amd.finish(dev)
        torch.cuda.synchronize(dev)
        pdist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = pdist.max_over_ranks(time.perf_counter() - t0, device=dev)

    # dominant-kernel duration: HIP events recorded by the library around k_fused on the launch stream
    # ... and around everything the call enqueues (prl_hip_last_call_ms): the other sweeps of Wolf-Jolion, refinement, fix-up,
    # the morphology pass - what one prl::binarize*() call costs on the device
    _capi.check(L.prl_hip_set_profiling(1))
    kms, cms = [], []
    for _ in range(max(3, min(args.steps, 10))):
        step()
        ms = C.c_float(0)
        _capi.check(L.prl_hip_last_kernel_ms(C.byref(ms)))
        kms.append(ms.value)
        _capi.check(L.prl_hip_last_call_ms(C.byref(ms)))
        cms.append(ms.value)
    _capi.check(L.prl_hip_set_profiling(0))
    kernel_ms = sum(kms) / len(kms)
    call_ms = sum(cms) / len(cms)
    stats = prlib_amd.last_stats()

    # measured ceilings of this box beside the 8 TB/s spec peak: hand-written dwordx4 kernels (16 B per lane,
    # non-temporal, grid-stride) on THIS batch - read-only (the page batch), write-only (the mask batch), copy (pages -> a scratch
    # batch) - tools/ubench/stream_probe.hip, not part of the product library
    ceil = {"measured_read_gbs": None, "measured_write_gbs": None, "measured_copy_gbs": None}
    if rank == 0 and args.ceilings:
        so = os.path.join(ROOT, "tools", "ubench", "libstream_probe.so")
        try:
            P = C.CDLL(so)
            P.prl_probe_stream.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_float)]
            nbytes = min(pages.numel(), out.numel()) // 16 * 16     # (pitched batches: the whole allocations are streamed)
            src_t = pages if pages.is_contiguous() else pages.contiguous()
            scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize(dev)
            for mode, key, moved in ((0, "measured_read_gbs", nbytes), (1, "measured_write_gbs", nbytes), (2, "measured_copy_gbs", 2 * nbytes)):
                ms = C.c_float(0)
                dst_ptr = scratch.data_ptr()
                if P.prl_probe_stream(mode, src_t.data_ptr(), dst_ptr, nbytes, 5, C.byref(ms)) == 0 and ms.value > 0:
                    ceil[key] = round(moved / (ms.value * 1e-3) / 1e9, 1)
            del scratch
        except OSError:
            pass

    px_per_step_total = total_pages * g.out_w * g.out_h
    bytes_per_px = 3 if method == prlib_amd.WOLFJOLION else 2
    alg_bytes = n_mine * (H * W * (bytes_per_px - 1) + g.out_w * g.out_h)   # rank 0's launch
    dominant_gbs = alg_bytes / (kernel_ms * 1e-3) / 1e9
    call_gbs = alg_bytes / (call_ms * 1e-3) / 1e9
    # calls with more than one heavy kernel (Wolf-Jolion's three sweeps, a morphology pass) are priced on the WHOLE call
    multi_kernel = method == prlib_amd.WOLFJOLION or args.morph != 0 or args.mode != "auto"
    achieved_gbs = call_gbs if multi_kernel else dominant_gbs

    # parity spot check (outside the timed region): first pages against the CPU oracle
    mismatches = None
    checked = []
    cpu = None
    vs_opencv = None
    vs_opencv_why = None
    if rank == 0:
        from oracle import capi as oc

        po = oc.make_params(method, args.window, args.k, args.morph)
        n_chk = min(args.check_pages, n_mine)
        if n_chk > 0:
            # first, last and evenly spaced pages of this rank's block
            idx = sorted({round(i * (n_mine - 1) / max(1, n_chk - 1)) for i in range(n_chk)})
            sel = torch.tensor(idx, device=dev)
            host = pages.index_select(0, sel).cpu().numpy()
            want = oc.binarize_batch(host.copy(), po, threads=os.cpu_count() or 1)
            got = out.index_select(0, sel)[:, :, : g.out_w].cpu().numpy()
            mismatches = int((want != got).sum())
            checked = idx
        if world == 1:
            try:   # the oracle against real OpenCV where this box has it (None: not installed, parity unpinned)
                from oracle import opencv_check

                rep = opencv_check.report(timeout=300)
                vs_opencv = None if (rep is None or rep.get("opencv") is None) else rep
                if vs_opencv is None:   # say WHY the pin did not run: headers_absent / build_failed / make_failed / run_failed
                    vs_opencv_why = (rep or {}).get("why", "no report")
                    if (rep or {}).get("stderr"):
                        vs_opencv_why += ": " + rep["stderr"][-300:]
            except Exception as e:
                vs_opencv = None
                vs_opencv_why = "exception: " + repr(e)
        if args.cpu_seconds > 0 and world == 1:   # the CPU baseline leg runs at N=1 only
            n_host = min(n_mine, 64)
            cpu = cpu_baseline(pages[:n_host].cpu().numpy().copy(), po, args.cpu_seconds)

    # ---- worst case and end to end (N = 1, outside the timed region; VERDICT r3 "next" 2) -----------------------------------
    worst = None
    e2e = None
    if rank == 0 and world == 1 and args.worst_case and args.mode == "auto":
        worst = worst_case_legs(args, prlib_amd, _capi, L, dev, pages, out, params, method, g, W, H)
    if rank == 0 and world == 1 and args.end_to_end and args.mode == "auto":
        e2e = end_to_end_leg(prlib_amd, dev, pages, params, g, W, H)

    page_digests = None
    if args.digest:   # (small test batches: the masks go through the host)
        import zlib

        mine_d = [zlib.crc32(out[i, :, : g.out_w].cpu().numpy().tobytes()) for i in range(n_mine)]
        page_digests = pdist.gather_lists(mine_d)

    # --scaling both (N > 1): the same job size as one GPU's weak-scaling share, now split over the ranks (strong scaling);
    # every rank re-uses the first pages of its batch, the timing protocol is the one above
    strong = None
    if args.scaling == "both" and world > 1:
        s_mine = pdist.page_range(args.pages, world, rank)
        sp, so_ = pages[: len(s_mine)], out[: len(s_mine)]
        for _ in range(args.warmup if len(s_mine) else 0):   # (fewer pages than ranks: this rank only joins the barriers)
            prlib_amd.binarize(sp, params, out=so_)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps if len(s_mine) else 0):
            prlib_amd.binarize(sp, params, out=so_)
        barrier()
        s_elapsed = pdist.max_over_ranks(time.perf_counter() - t0, device=dev)
        strong = {"pages_total": args.pages, "pages_per_gpu": len(s_mine), "ms_per_step": round(s_elapsed / args.steps * 1e3, 4),
                  "value": round(args.pages * g.out_w * g.out_h * args.steps / s_elapsed / 1e6, 1), "unit": "Mpixels/s",
                  "note": "strong scaling: the job of ONE GPU's weak-scaling share split over all ranks"}

    if rank == 0:
        value = px_per_step_total * args.steps / elapsed / 1e6
        line = {
            "metric": "Mpixels/s Sauvola w=31 on batched 4K pages" if (args.method == "sauvola" and args.window == 31)
            else f"Mpixels/s {args.method} w={args.window}",
            "value": round(value, 1),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak" if args.scaling == "both" else args.scaling,
            "vs_baseline": None,
            "dtype": "u8 in/out; exact integer window sums (u32; f32 below 2^24 for windows up to 31); f32 decision with f64/literal refinement",
            "data": "synthetic",
            "config": {
                "workload": f"{args.pages} x {W}x{H} u8 pages {'in total' if args.scaling == 'strong' else 'per GPU'}, "
                            f"{args.method} k={args.k} w={args.window} morph={args.morph}, mode={args.mode}",
                "pages_per_gpu": n_mine,
                "pages_total": total_pages,
                "parallelism": f"pages sharded over {world} GPU(s), no collectives",
                "dist_backend": pdist.backend_name(),   # only the barrier and the max-over-ranks of the time go through it
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved_gbs, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved_gbs / HBM_PEAK_GBS, 4),
                "frac_basis": "whole call (several heavy kernels: algorithmic bytes / call_ms)" if multi_kernel
                              else "dominant kernel (algorithmic bytes / kernel_ms; the call's other kernels are in frac_whole_call)",
                "frac_dominant_kernel": round(dominant_gbs / HBM_PEAK_GBS, 4),
                "frac_whole_call": round(call_gbs / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_note": traffic_note,
                # which box this was: the kernel's own clock and what a wavefront-row cost in SIMD cycles (a build's speed is the
                # second number, a box's the first)
                "shader_clock_ghz": clock["shader_clock_ghz"],
                "simd_cycles_per_wavefront_row": simd_cycles_per_row(clock, args, W, g, n_mine),
                "clock_note": clock["clock_note"],
                # (windows of 33..129 columns run their threshold sweep in k_fused_q)
                "kernel": ("k_fused_q" if 32 <= args.window - 1 <= 128 else "k_fused") if args.mode == "auto" else "literal chain",
                "kernel_ms": round(kernel_ms, 4),
                "call_ms": round(call_ms, 4),
                "algorithmic_bytes_per_launch": alg_bytes,
                "measured_read_gbs": ceil["measured_read_gbs"],
                "measured_write_gbs": ceil["measured_write_gbs"],
                "measured_copy_gbs": ceil["measured_copy_gbs"],
                "frac_of_measured_copy": round(achieved_gbs / ceil["measured_copy_gbs"], 4) if ceil["measured_copy_gbs"] else None,
            },
            "cpu_baseline": cpu,
            "parity": {"checked_pages": checked, "mismatching_pixels": mismatches,
                       "refined_pixels": int(stats.refined_pixels), "exact_pixels": int(stats.exact_pixels),
                       "literal_pages": int(stats.literal_pages), "vs_opencv": vs_opencv, "vs_opencv_why": vs_opencv_why},
        }
        if strong is not None:
            line["strong"] = strong
        if page_digests is not None:
            line["page_digests"] = page_digests
        if worst and worst.get("adversarial"):   # (the same call on the same number of pages: how much a page that defeats the fast path costs)
            adv = worst["adversarial"]
            adv["per_page_vs_headline"] = round((adv["ms_per_step"] / adv["pages"]) / (line["ms_per_step"] / line["config"]["pages_per_gpu"]), 2)
        line["worst_case"] = worst
        line["end_to_end"] = e2e
        print(json.dumps(line), flush=True)
    pdist.finish()


if __name__ == "__main__":
    main()
