import sys, time, json, os
sys.path.insert(0, '.')
import torch, numpy as np
import prlib_amd
from prlib_amd import synth
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pages, skews = synth.text_pages_torch(n, 3508, 2480, dev, channels=1)
torch.cuda.synchronize()
prlib_amd.deskew(pages[:2])
torch.cuda.synchronize()
t = time.perf_counter(); outs, ang = prlib_amd.deskew(pages); torch.cuda.synchronize(); dt = time.perf_counter() - t
print(json.dumps({"pages": n, "work_mb": os.environ.get("PRL_HIP_DESKEW_WORK_MB"), "deskew_s": round(dt, 3), "pages_per_s": round(n / dt, 1)}))
