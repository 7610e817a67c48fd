"""A/B of builds of the library on prl::deskew: synthetic A4 text pages at several batch sizes and the reference's colour scans tiled
to A4.   python tools/dbg/ppht_ab.py <lib.so or ''> [pages ...]"""
import glob, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import prlib_amd
from prlib_amd import _capi, synth
from bench_real import tiled_colour_page

lib = sys.argv[1] if len(sys.argv) > 1 else ""
if lib:
    _capi.use_library(os.path.abspath(lib))
sizes = [int(x) for x in sys.argv[2:]] or [1, 64, 256]
H, W = 3508, 2480
dev = torch.device("cuda:0")
prlib_amd.deskew(synth.text_pages_torch(2, H, W, dev, channels=1)[0])
res = {"lib": lib or "product"}
for n in sizes:
    pages, _ = synth.text_pages_torch(n, H, W, dev, channels=1)
    best = 1e9
    for _ in range(2):
        torch.cuda.synchronize(); t = time.perf_counter(); prlib_amd.deskew(pages); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    res[f"synthetic_{n}_s"] = round(best, 3)
    del pages
scans = [np.load(p)["bgr"] for p in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "stages", "chain_*.npz")))]
real = torch.empty((64, H, W, 3), dtype=torch.uint8, device=dev)
for i in range(64):
    real[i] = torch.from_numpy(tiled_colour_page(scans[i % len(scans)], H, W, i)).to(dev)
best = 1e9
for _ in range(2):
    torch.cuda.synchronize(); t = time.perf_counter(); outs, ang = prlib_amd.deskew(real); torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t)
res["real_64_s"] = round(best, 3)
res["angles_crc"] = int(np.frombuffer(np.asarray(ang).tobytes(), np.uint8).sum())
print(json.dumps(res))
