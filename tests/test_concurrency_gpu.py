"""The C ABI's threading and memory promises (include/prl_hip.h "Threading"; SURVEY.md §8b "thread-safe per (device, stream)"):

  * four host threads, each on its own stream, mixed methods (Wolf-Jolion included: it owns a side stream, lazily launched
    literal kernels and its own events) - every mask equals the oracle's
  * two threads sharing ONE stream (they share that stream's workspace: the library serialises them)
  * 2 000 calls of random sizes and kinds; after prl_hip_release_workspace() the device's free memory (hipMemGetInfo) is back
    at the baseline taken after the warm-up - nothing leaks per call

Inputs and expected masks are prepared before the threads start (the oracle is the checker; the threads only call the
product and compare bytes), so the calls really overlap: ctypes releases the GIL for the duration of each call.
"""
import threading
import time

import numpy as np
import pytest

from prlib_amd import synth

pytestmark = pytest.mark.gpu


def _cases(prl, oracle, seed, n_cases, methods):
    rng = np.random.default_rng(seed)
    cases = []
    for it in range(n_cases):
        m = methods[it % len(methods)]
        h, w = int(rng.integers(120, 420)), int(rng.integers(520, 1100))
        win = int(rng.choice([15, 31, 63]))
        k = float(rng.choice([0.01, 0.3, -0.2, 0.5])) if m != prl.FENG else 0.0
        morph = int(rng.choice([0, 2]))
        n = int(rng.choice([1, 2, 5]))
        pages = np.stack([synth.page_numpy(h, w, index=int(rng.integers(0, 1 << 20))) for _ in range(n)])
        if it % 6 == 0:
            pages[0][:, :] = int(rng.integers(0, 256))     # a flat page: ties everywhere, Wolf-Jolion's every pixel a candidate
        po = oracle.make_params(m, win, k, morph)
        want = np.stack([oracle.binarize(pages[i], po) for i in range(n)])
        cases.append((pages, prl.make_params(m, win, k, morph), want))
    return cases


def _run_threads(workers):
    errors = []

    def guard(fn):
        def run():
            try:
                fn()
            except BaseException as e:   # noqa: BLE001 - reported by the test
                errors.append(repr(e))
        return run

    ts = [threading.Thread(target=guard(w)) for w in workers]
    t0 = time.perf_counter()
    [t.start() for t in ts]
    [t.join() for t in ts]
    return errors, time.perf_counter() - t0


def test_four_threads_own_streams_mixed_methods(prl, oracle, cuda_device):
    import torch

    order = [prl.WOLFJOLION, prl.SAUVOLA, prl.NICK, prl.FENG, prl.NIBLACK]
    per_thread = [_cases(prl, oracle, 500 + t, 12, order[t:] + order[:t]) for t in range(4)]
    bad = [0, 0, 0, 0]

    def worker(t):
        def run():
            s = torch.cuda.Stream(device=cuda_device)
            for rep in range(2):
                for pages, p, want in per_thread[t]:
                    with torch.cuda.stream(s):
                        got = prl.binarize(torch.from_numpy(pages).to(cuda_device), p)
                        s.synchronize()
                    bad[t] += int((got.cpu().numpy() != want).sum())
        return run

    errors, dt = _run_threads([worker(t) for t in range(4)])
    assert not errors, errors
    assert bad == [0, 0, 0, 0], f"mismatching pixels per thread: {bad}"
    print(f"4 threads x own streams x {2 * 12} calls each (5 methods): {dt:.2f} s, all masks equal the oracle's")


def test_two_threads_sharing_one_stream(prl, oracle, cuda_device):
    import torch

    shared = torch.cuda.Stream(device=cuda_device)
    per_thread = [_cases(prl, oracle, 700 + t, 10, [prl.WOLFJOLION, prl.SAUVOLA, prl.FENG]) for t in range(2)]
    bad = [0, 0]

    def worker(t):
        def run():
            for pages, p, want in per_thread[t]:
                with torch.cuda.stream(shared):
                    got = prl.binarize(torch.from_numpy(pages).to(cuda_device), p)
                    shared.synchronize()
                bad[t] += int((got.cpu().numpy() != want).sum())
        return run

    errors, dt = _run_threads([worker(0), worker(1)])
    assert not errors, errors
    assert bad == [0, 0], bad
    # and the legacy default stream (NULL) from two threads, the way two cv::Mat callers without streams arrive
    bad = [0, 0]

    def host_worker(t):
        def run():
            for pages, p, want in per_thread[t][:5]:
                for i in range(pages.shape[0]):
                    bad[t] += int((prl.binarize(pages[i], p) != want[i]).sum())   # numpy in: prl_hip_binarize_host
        return run

    errors, _ = _run_threads([host_worker(0), host_worker(1)])
    assert not errors, errors
    assert bad == [0, 0], bad


def test_soak_2000_calls_free_memory_returns_to_baseline(prl, cuda_device):
    import torch
    from prlib_amd import _capi

    L = _capi.lib()
    rng = np.random.default_rng(42)
    methods = [prl.SAUVOLA, prl.NIBLACK, prl.WOLFJOLION, prl.NICK, prl.FENG]
    streams = [torch.cuda.Stream(device=cuda_device) for _ in range(3)]

    def one_call(it):
        kind = it % 40
        h, w = int(rng.integers(60, 360)), int(rng.integers(64, 900))
        n = int(rng.choice([1, 1, 2, 4]))
        if kind == 7:      # NL-means (colour)
            t = torch.randint(0, 256, (1, min(h, 96), min(w, 128), 3), dtype=torch.uint8, device=cuda_device)
            prl.denoise(t, 5.5)
        elif kind == 13:   # thinning
            t = (torch.rand((n, h, w), device=cuda_device) > 0.6).to(torch.uint8) * 255
            prl.thinZhangSuen(t)
        elif kind == 19:   # background normalisation
            prl.backgroundNormalization(torch.randint(100, 256, (n, h, w), dtype=torch.uint8, device=cuda_device))
        elif kind == 23:   # deskew (Hough lists sized from the page's ink: they grow and shrink with the input)
            prl.deskew(torch.from_numpy(synth.text_page_numpy(150, 208, it, skew_deg=2.0))[None].to(cuda_device))
        elif kind == 29:   # the one-call chain
            prl.process_pages(torch.randint(0, 256, (2, 90, 130, 3), dtype=torch.uint8, device=cuda_device), 3, prl.SAUVOLA, 15, 0.34, 1,
                              denoise_strength=5.5, thin=0, deskew=True, background_normalization=True)
        elif kind == 31:   # a host image through the staging area
            prl.binarize(np.full((h, max(w, 70)), 200, np.uint8), prl.make_params(prl.SAUVOLA, 15, 0.34, 0))
        else:
            m = methods[it % 5]
            win = int(rng.choice([3, 15, 31, 101]))
            t = torch.randint(0, 256, (n, h, w), dtype=torch.uint8, device=cuda_device)
            p = prl.make_params(m, win, 0.2 if m != prl.FENG else 0.0, int(it % 3))
            with torch.cuda.stream(streams[it % 3]):
                if win < min(h, w):
                    prl.binarize(t, p)
                else:   # a page no larger than the window: Wolf-Jolion / NICK / Feng have no pixel to produce (the reference's
                    try:   # empty-ROI cv::Exception); an error return must not leak either
                        prl.binarize(t, p)
                    except (ValueError, RuntimeError):
                        pass

    def settle():
        torch.cuda.synchronize()
        _capi.check(L.prl_hip_release_workspace())
        torch.cuda.empty_cache()
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info(cuda_device)[0]

    for it in range(80):            # warm-up: every code object loaded, every lazily created stream / event exists
        one_call(it)
    base = settle()
    t0 = time.perf_counter()
    low = base
    for it in range(2000):
        one_call(it)
        if it % 500 == 499:
            low = min(low, torch.cuda.mem_get_info(cuda_device)[0])
    dt = time.perf_counter() - t0
    after = settle()
    print(f"soak: 2000 calls in {dt:.1f} s; free memory baseline {base >> 20} MiB, lowest while running {low >> 20} MiB, "
          f"after release {after >> 20} MiB")
    assert after >= base - (8 << 20), f"free device memory fell by {(base - after) >> 20} MiB over 2000 calls"
    assert dt < 120
