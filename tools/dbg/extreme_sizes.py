"""Extreme page shapes against the oracle: very large, very wide, very tall, tiny."""
import sys, json, time
sys.path.insert(0, '.')
import numpy as np, torch
import prlib_amd
from prlib_amd import synth
from oracle import capi as oc
dev = torch.device("cuda:0")
res = {}
for name, (h, w) in {"16384x16384": (16384, 16384), "64x32767": (64, 32767), "32767x64": (32767, 64), "33x33": (33, 33), "2x40000": (40, 40000)}.items():
    rng = np.random.default_rng(h + w)
    page = synth.page_numpy(min(h, 2048), min(w, 2048), index=h % 97)
    page = np.tile(page, ((h + page.shape[0] - 1) // page.shape[0], (w + page.shape[1] - 1) // page.shape[1]))[:h, :w].copy()
    page[rng.integers(0, h, 2000), rng.integers(0, w, 2000)] = rng.integers(0, 256, 2000)
    out = {}
    for method, win, k, morph in ((prlib_amd.SAUVOLA, 31, 0.34, 2), (prlib_amd.WOLFJOLION, 15, 0.3, 0), (prlib_amd.NICK, 101 if min(h, w) > 110 else 21, -0.1, 0)):
        if min(h, w) <= win:
            continue
        got = prlib_amd.binarize(torch.from_numpy(page).to(dev)[None], prlib_amd.make_params(method, win, k, morph))[0].cpu().numpy()
        want = oc.binarize(page, oc.make_params(method, win, k, morph))
        out[f"m{method}_w{win}"] = int((got != want).sum()) if got.shape == want.shape else f"shape {got.shape} vs {want.shape}"
    if h * w <= 1 << 26:
        m = (page < 128).astype(np.uint8) * 255
        out["thin"] = int((prlib_amd.thinZhangSuen(torch.from_numpy(m).to(dev)).cpu().numpy() != oc.thin(m, 0)).sum())
        if w <= 40950:
            out["bgnorm"] = int((prlib_amd.backgroundNormalization(torch.from_numpy(page).to(dev)).cpu().numpy() != oc.bgnorm(page)).sum())
        if max(h, w) <= 32767 and max(h, w) ** 2 <= 1 << 28:
            got = prlib_amd.rotate(torch.from_numpy(page).to(dev)[None], [7.0])[0].cpu().numpy()
            out["rotate7"] = int((got != oc.rotate(page, 7.0)).sum())
    res[name] = out
print(json.dumps(res))
