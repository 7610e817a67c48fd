"""prl::deskew on n synthetic A4 text pages with the phase counters of the group kernel (hooks build, PRL_HIP_PPHT_PROF=1).
    python tools/dbg/ppht_group_prof.py 1 8 64"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("PRL_HIP_PPHT_PROF", "1")
os.environ.setdefault("PRL_HIP_DEBUG", "1")
import torch
import prlib_amd
from prlib_amd import _capi, synth
_capi.use_library(os.environ.get('PRL_LIB', _capi.HOOKS_LIB_PATH))
dev = torch.device("cuda:0")
for n in [int(v) for v in sys.argv[1:]] or [1]:
    pages, _ = synth.text_pages_torch(n, 3508, 2480, dev, channels=1)
    torch.cuda.synchronize()
    prlib_amd.deskew(pages[:1])
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        outs, ang = prlib_amd.deskew(pages)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"pages": n, "deskew_s": round(dt, 4), "pages_per_s": round(n / dt, 1)}), flush=True)
    del pages, outs
