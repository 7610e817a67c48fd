"""Wolf-Jolion from four host threads at once, each on its own stream (own workspace, own side stream): every mask equals the
oracle's.  The side-stream schedule (events between the caller's stream and the workspace's side stream) under concurrency."""
import json, os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import prlib_amd
from prlib_amd import synth
from oracle import capi as oc

dev = torch.device("cuda:0")
bad, calls = [0], [0]
lock = threading.Lock()


def worker(tid):
    rng = np.random.default_rng(100 + tid)
    s = torch.cuda.Stream(device=dev)
    for it in range(25):
        h, w = int(rng.integers(200, 700)), int(rng.integers(520, 1500))
        win = int(rng.choice([15, 31, 63, 101]))
        k = float(rng.choice([0.01, 0.3, -0.2, 0.5]))
        morph = int(rng.choice([0, 2]))
        n = int(rng.choice([1, 3, 9]))
        pages = np.stack([synth.page_numpy(h, w, index=int(rng.integers(0, 1 << 20))) for _ in range(n)])
        if it % 5 == 0:
            pages[0][:, :] = int(rng.integers(0, 256))     # a flat page: every pixel a maximum candidate
        p = prlib_amd.make_params(prlib_amd.WOLFJOLION, win, k, morph)
        with torch.cuda.stream(s):
            got = prlib_amd.binarize(torch.from_numpy(pages).to(dev), p).cpu().numpy()
        po = oc.make_params(oc.WOLFJOLION, win, k, morph)
        b = sum(int((got[i] != oc.binarize(pages[i], po)).sum()) for i in range(n))
        with lock:
            bad[0] += b
            calls[0] += 1


ts = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
[t.start() for t in ts]
[t.join() for t in ts]
print(json.dumps({"threads": 4, "calls": calls[0], "mismatching_pixels": bad[0]}))
sys.exit(1 if bad[0] else 0)
