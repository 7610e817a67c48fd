"""Where k_ppht_mw's time goes on the reference's colour scans tiled to A4 pages (hooks build, PRL_HIP_PPHT_PROF=1: cycles per phase
and event counts of the heaviest pages on stderr) beside the synthetic text scans.   python tools/dbg/ppht_real.py [pages]"""
import glob, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("PRL_HIP_PPHT_PROF", "1")
os.environ.setdefault("PRL_HIP_DEBUG", "1")
import numpy as np, torch
import prlib_amd
from prlib_amd import _capi, synth
from bench_real import tiled_colour_page

_capi.use_library(os.environ.get('PRL_LIB', _capi.HOOKS_LIB_PATH))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H, W = 3508, 2480
dev = torch.device("cuda:0")
scans = [np.load(p)["bgr"] for p in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "stages", "chain_*.npz")))]
real = torch.empty((n, H, W, 3), dtype=torch.uint8, device=dev)
for i in range(n):
    real[i] = torch.from_numpy(tiled_colour_page(scans[i % len(scans)], H, W, i)).to(dev)
syn, _ = synth.text_pages_torch(n, H, W, dev, seed=7000, channels=3)
for tag, pages in (("real", real), ("synthetic", syn)):
    for rep in range(2):
        prlib_amd.deskew_stats(reset=True)
        torch.cuda.synchronize(); t = time.perf_counter()
        outs, ang = prlib_amd.deskew(pages)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        print(tag, rep, f"{dt*1e3:.1f} ms", prlib_amd.deskew_stats().as_dict(), file=sys.stderr, flush=True)
        del outs
# the heaviest real page alone: is it the page or the company it keeps?
heavy = real[2:3]
for rep in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    prlib_amd.deskew(heavy)
    torch.cuda.synchronize(); print("one page alone", f"{(time.perf_counter()-t)*1e3:.1f} ms", file=sys.stderr, flush=True)
