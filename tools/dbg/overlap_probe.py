"""NL-means beside the angle search of deskew: time of denoise alone and while prlib_amd.deskew runs on another stream /
thread with n pages (is the slowdown of the overlapped chain pass a memory-system effect - grows with n - or a scheduling one?)"""
import sys, time, json, threading
sys.path.insert(0, '.')
import torch, numpy as np
import prlib_amd
from prlib_amd import synth
dev = torch.device('cuda:0')
pages, _ = synth.text_pages_torch(384, 3508, 2480, dev, channels=1)
col, _ = synth.text_pages_torch(48, 3508, 2480, dev, channels=3, seed=5)
torch.cuda.synchronize()
prlib_amd.deskew(pages[:2]); prlib_amd.denoise(col[:2], 10.0); torch.cuda.synchronize()
def t_denoise():
    cs = torch.cuda.current_stream(dev)   # (a device-wide synchronize would wait for the search on the other stream)
    cs.synchronize(); t = time.perf_counter(); prlib_amd.denoise(col, 10.0); cs.synchronize(); return time.perf_counter() - t
print(json.dumps({"denoise_alone_s": round(t_denoise(), 3)}))
for n in (8, 32, 128, 384):
    s2 = torch.cuda.Stream(device=dev)
    res = {}
    def bg():
        with torch.cuda.stream(s2):
            t = time.perf_counter(); prlib_amd.deskew(pages[:n]); s2.synchronize(); res["deskew_s"] = time.perf_counter() - t
    t = time.perf_counter(); prlib_amd.deskew(pages[:n]); torch.cuda.synchronize(); alone = time.perf_counter() - t
    th = threading.Thread(target=bg); th.start()
    time.sleep(0.15)   # let the search kernel start
    ts = []
    while th.is_alive():
        ts.append(t_denoise())
    th.join()
    print(json.dumps({"deskew_pages": n, "deskew_alone_s": round(alone, 3), "deskew_beside_s": round(res["deskew_s"], 3),
                      "denoise_beside_s": [round(x, 3) for x in ts]}))
