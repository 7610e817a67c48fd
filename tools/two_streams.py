#!/usr/bin/env python3
"""Two streams on one device: small batches issued alternately on two HIP streams with deferred
completion; per-stream HIP events show whether the calls overlap.  Also prints the per-call latency of one 4K page
(the C2 configuration) in the default and in the deferred mode."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import prlib_amd
from prlib_amd import synth

dev = torch.device("cuda:0")
res = {}
page = synth.pages_torch(1, 4096, 4096, dev)
p15 = prlib_amd.make_params(prlib_amd.SAUVOLA, 15, 0.34, 0)
out = torch.empty((1, 4095, 4096), dtype=torch.uint8, device=dev)[:, :, :4095]
for mode in (False, True):
    prlib_amd.set_deferred_completion(mode)
    for _ in range(20):
        prlib_amd.binarize(page, p15, out=out)
    prlib_amd.finish(dev)
    n = 400
    t0 = time.perf_counter()
    for _ in range(n):
        prlib_amd.binarize(page, p15, out=out)
    prlib_amd.finish(dev)
    res["one_4k_page_w15_ms_per_call_" + ("deferred" if mode else "default")] = round((time.perf_counter() - t0) / n * 1e3, 4)

# overlap: 8 pages per call (far from filling the chip), alternately on two streams
prlib_amd.set_deferred_completion(True)
pages = synth.pages_torch(8, 2048, 2048, dev)
p31 = prlib_amd.make_params(prlib_amd.SAUVOLA, 31, 0.34, 0)
s = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
outs = [torch.empty((8, 2047, 2048), dtype=torch.uint8, device=dev)[:, :, :2047] for _ in range(2)]
def run(streams, reps=50):
    for st in streams:
        with torch.cuda.stream(st):
            prlib_amd.binarize(pages, p31, out=outs[0]); prlib_amd.finish(dev)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for r in range(reps):
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                prlib_amd.binarize(pages, p31, out=outs[i])
    for st in streams:
        with torch.cuda.stream(st):
            prlib_amd.finish(dev)
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / (reps * len(streams)) * 1e3
res["8x2k_pages_ms_per_call_one_stream"] = round(run([s[0]]), 4)
res["8x2k_pages_ms_per_call_two_streams"] = round(run(s), 4)
prlib_amd.set_deferred_completion(False)
print(json.dumps(res))
