"""One A4 page: HoughLinesP with the real threshold vs a threshold nothing reaches (votes only, no line walks)."""
import sys, time, json
sys.path.insert(0, '.')
import numpy as np, torch
import prlib_amd
from prlib_amd import synth
from oracle import capi as oc
p = synth.text_page_numpy(3508, 2480, 7, skew_deg=2.0, shading=0.2)
thr, binary = oc.otsu(p)
inv = torch.from_numpy(255 - binary).cuda()
res = {}
for name, t in (("real", 100), ("votes_only", 10**9)):
    prlib_amd.houghp(inv, t, 310, 20); torch.cuda.synchronize()
    t0 = time.perf_counter(); lines = prlib_amd.houghp(inv, t, 310, 20); torch.cuda.synchronize()
    res[name] = {"s": round(time.perf_counter() - t0, 3), "segments": int(len(lines))}
res["points"] = int((binary == 0).sum())
print(json.dumps(res))
