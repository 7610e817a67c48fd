"""What slows the angle search inside the chain: deskew of 256 pages alone, beside denoise only, beside the other stages only,
beside the whole rest of the chain (denoise + backgroundNormalization + Sauvola + thinning on pre-rotated pages)."""
import sys, time, json, threading
sys.path.insert(0, '.')
import torch, numpy as np
import prlib_amd
from prlib_amd import synth
dev = torch.device('cuda:0')
pages, _ = synth.text_pages_torch(256, 3508, 2480, dev, channels=1)
col, _ = synth.text_pages_torch(48, 3508, 3508, dev, seed=5, channels=3)
torch.cuda.synchronize()
def rest(den, others):
    if den and others:
        return lambda: prlib_amd.process_pages(col, 3, prlib_amd.SAUVOLA, 31, 0.34, 0, denoise_strength=10.0, thin=0, background_normalization=True)
    if den:
        return lambda: prlib_amd.denoise(col, 10.0)
    return lambda: prlib_amd.process_pages(col, 3, prlib_amd.SAUVOLA, 31, 0.34, 0, thin=0, background_normalization=True)
prlib_amd.deskew(pages[:2]); rest(True, True)(); torch.cuda.synchronize()
res = {}
t = time.perf_counter(); prlib_amd.deskew(pages); torch.cuda.synchronize(); res["deskew_alone_s"] = round(time.perf_counter() - t, 3)
for name, fn in (("denoise", rest(True, False)), ("other_stages", rest(False, True)), ("whole_rest", rest(True, True))):
    cs = torch.cuda.current_stream(dev)
    cs.synchronize(); t = time.perf_counter(); fn(); cs.synchronize(); alone = time.perf_counter() - t
    s2 = torch.cuda.Stream(device=dev)
    out = {}
    def bg():
        with torch.cuda.stream(s2):
            t0 = time.perf_counter(); prlib_amd.deskew(pages); s2.synchronize(); out["d"] = time.perf_counter() - t0
    th = threading.Thread(target=bg); th.start(); time.sleep(0.1)
    ts = []
    while th.is_alive():
        cs.synchronize(); t = time.perf_counter(); fn(); cs.synchronize(); ts.append(round(time.perf_counter() - t, 3))
    th.join()
    res[name] = {"alone_s": round(alone, 3), "beside_s": ts, "deskew_beside_s": round(out["d"], 3)}
print(json.dumps(res))
