"""Does a CU-masked stream keep the angle search and NL-means out of each other's way?  deskew of 256 pages on a stream limited
to N CUs, denoise on a stream limited to the other CUs: alone and together."""
import sys, time, json, threading, ctypes as C
sys.path.insert(0, '.')
import torch, numpy as np
import prlib_amd
from prlib_amd import synth
hip = C.CDLL("libamdhip64.so")
dev = torch.device('cuda:0')
torch.zeros(1, device=dev)
NCU = 256

def masked_stream(cus):
    mask = (C.c_uint32 * 8)()
    for cu in cus:
        mask[cu // 32] |= 1 << (cu % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, mask)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)

n_search = int(sys.argv[1]) if len(sys.argv) > 1 else 96
# CU numbering: interleave so that both sets have CUs in every XCD
search_cus = [cu for cu in range(NCU) if (cu % 8) < (n_search * 8 // NCU)]
other_cus = [cu for cu in range(NCU) if cu not in search_cus]
s_search, s_nlm = masked_stream(search_cus), masked_stream(other_cus)
pages, _ = synth.text_pages_torch(256, 3508, 2480, dev, channels=1)
col, _ = synth.text_pages_torch(64, 3508, 2480, dev, seed=5, channels=3)
torch.cuda.synchronize()
prlib_amd.deskew(pages[:2]); prlib_amd.denoise(col[:2], 10.0); torch.cuda.synchronize()
def t_on(stream, fn):
    with torch.cuda.stream(stream):
        t = time.perf_counter(); fn(); stream.synchronize(); return time.perf_counter() - t
res = {"search_cus": len(search_cus)}
d = torch.cuda.default_stream(dev)
res["deskew_all_cus"] = round(t_on(d, lambda: prlib_amd.deskew(pages)), 3)
res["denoise_all_cus"] = round(t_on(d, lambda: prlib_amd.denoise(col, 10.0)), 3)
res["deskew_masked_alone"] = round(t_on(s_search, lambda: prlib_amd.deskew(pages)), 3)
res["denoise_masked_alone"] = round(t_on(s_nlm, lambda: prlib_amd.denoise(col, 10.0)), 3)
out = {}
def bg():
    out["deskew"] = t_on(s_search, lambda: prlib_amd.deskew(pages))
th = threading.Thread(target=bg); th.start(); time.sleep(0.1)
ts = []
while th.is_alive():
    ts.append(round(t_on(s_nlm, lambda: prlib_amd.denoise(col, 10.0)), 3))
th.join()
res["deskew_masked_beside"] = round(out["deskew"], 3); res["denoise_masked_beside"] = ts
out2 = {}
s2 = torch.cuda.Stream(device=dev)
def bg2():
    out2["deskew"] = t_on(s2, lambda: prlib_amd.deskew(pages))
th = threading.Thread(target=bg2); th.start(); time.sleep(0.1)
ts = []
while th.is_alive():
    ts.append(round(t_on(d, lambda: prlib_amd.denoise(col, 10.0)), 3))
th.join()
res["deskew_unmasked_beside"] = round(out2["deskew"], 3); res["denoise_unmasked_beside"] = ts
print(json.dumps(res))
