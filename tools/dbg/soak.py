"""Repeated chain / binarize / denoise calls of varying sizes: free device memory must settle (no growth per call)."""
import sys, json
sys.path.insert(0, '.')
import numpy as np, torch
import prlib_amd
from prlib_amd import synth
dev = torch.device("cuda:0")
free = []
shapes = [(700, 500), (1200, 900), (640, 480), (1200, 900), (700, 500)]
for it in range(15):
    h, w = shapes[it % len(shapes)]
    n = [48, 96, 32][it % 3]
    pages, _ = synth.text_pages_torch(n, h, w, dev, seed=100 + it, channels=3)
    outs, ang = prlib_amd.process_pages(pages, 3, prlib_amd.SAUVOLA, 31, 0.34, 0, denoise_strength=10.0, thin=0, deskew=True, background_normalization=True)
    if it % 4 == 0:
        host = [p for p in pages[:16].cpu().numpy()]
        prlib_amd.process_pages_host(host, prlib_amd.SAUVOLA, 31, 0.34, 0, denoise_strength=10.0, thin=0, deskew=True, background_normalization=True, n_devices=1)
    g = prlib_amd.cvtColorBGR2GRAY(pages)
    prlib_amd.binarize_pages_host([q for q in g[: 8 + 8 * (it % 3)].cpu().numpy()], prlib_amd.make_params(prlib_amd.SAUVOLA, 31, 0.34, 2 * (it % 2)), n_devices=1)
    prlib_amd.binarize(g, prlib_amd.make_params(prlib_amd.WOLFJOLION, 31, 0.3, 2))
    prlib_amd.binarizeByLocalVariances(pages[:8])
    del pages, outs, g
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    free.append(torch.cuda.mem_get_info()[0] >> 20)
print(json.dumps({"free_MiB_after_each_round": free, "settled": max(free[-5:]) - min(free[-5:]) < 64}))
