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
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --pages per GPU; strong: --pages in total, split over the ranks by dist.page_range")
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


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    import torch

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
        if rank == 0:
            print(json.dumps({"dryrun": True, "n_gpus": world, "pages_total": int(total), "max_rank_plus_1": slowest,
                              "first_block": [mine.start, mine.stop], "scaling": args.scaling}), flush=True)
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
    _capi.check(L.prl_hip_set_profiling(1))
    kms = []
    for _ in range(max(3, min(args.steps, 10))):
        step()
        ms = C.c_float(0)
        _capi.check(L.prl_hip_last_kernel_ms(C.byref(ms)))
        kms.append(ms.value)
    _capi.check(L.prl_hip_set_profiling(0))
    kernel_ms = sum(kms) / len(kms)
    stats = prlib_amd.last_stats()

    # measured device-copy ceiling of this box (SURVEY.md §8d asks for it beside the 8 TB/s spec peak): a plain
    # device-to-device copy of the page batch, bytes read + bytes written over the average of 5 copies
    copy_gbs = None
    if rank == 0:
        flat = torch.empty(min(n_mine * H * W, 1 << 32), dtype=torch.uint8, device=dev)  # (pages may be a pitched view)
        scratch = torch.empty_like(flat)
        scratch.copy_(flat)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            scratch.copy_(flat)
        e1.record()
        torch.cuda.synchronize(dev)
        copy_gbs = 2 * flat.numel() * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del scratch, flat

    px_per_step_total = total_pages * g.out_w * g.out_h
    bytes_per_px = 3 if method == prlib_amd.WOLFJOLION else 2
    alg_bytes = n_mine * (H * W * (bytes_per_px - 1) + g.out_w * g.out_h)   # rank 0's launch
    achieved_gbs = alg_bytes / (kernel_ms * 1e-3) / 1e9

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

    if rank == 0:
        value = px_per_step_total * args.steps / elapsed / 1e6
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        key = f"{args.method}_w{args.window}_{n_mine}x{W}x{H}_{args.mode}"
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(key)
            except Exception:
                traffic = None
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
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u8 in/out; exact integer window sums (u32, f32 below 2^24 in interior strips); f32 decision with f64/literal refinement",
            "data": "synthetic",
            "config": {
                "workload": f"{args.pages} x {W}x{H} u8 pages {'per GPU' if args.scaling == 'weak' else 'in total'}, "
                            f"{args.method} k={args.k} w={args.window} morph={args.morph}, mode={args.mode}",
                "pages_per_gpu": n_mine,
                "pages_total": total_pages,
                "parallelism": f"pages sharded over {world} GPU(s), no collectives",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved_gbs, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved_gbs / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "kernel": "k_fused" if args.mode == "auto" else "literal chain",
                "kernel_ms": round(kernel_ms, 4),
                "algorithmic_bytes_per_launch": alg_bytes,
                "measured_copy_gbs": round(copy_gbs, 1) if copy_gbs else None,
            },
            "cpu_baseline": cpu,
            "parity": {"checked_pages": checked, "mismatching_pixels": mismatches,
                       "refined_pixels": int(stats.refined_pixels), "exact_pixels": int(stats.exact_pixels),
                       "literal_pages": int(stats.literal_pages), "vs_opencv": vs_opencv, "vs_opencv_why": vs_opencv_why},
        }
        print(json.dumps(line), flush=True)
    pdist.finish()


if __name__ == "__main__":
    main()
