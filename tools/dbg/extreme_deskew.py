"""prl::deskew / prl::findAngle on extreme page shapes against the oracle (accumulator rows of odd and even length, pairs of cells that
straddle two angle rows, very wide / very tall pages, tiny pages, all-dark and all-white pages)."""
import sys, json
sys.path.insert(0, '.')
import numpy as np, torch
import prlib_amd
from prlib_amd import synth
from oracle import capi as oc
dev = torch.device("cuda:0")
res = {}
shapes = {"300x421": (300, 421), "301x420": (301, 420), "97x4001": (97, 4001), "4000x98": (4000, 98), "40x50": (40, 50), "1200x1600": (1200, 1600),
          "64x20000": (64, 20000), "20000x64": (20000, 64), "2047x2049": (2047, 2049)}
for name, (h, w) in shapes.items():
    g = synth.text_page_numpy(h, w, 11 + h % 7, skew_deg=2.5 if h % 2 else -1.5)
    out = {}
    for tag, page in (("text", g), ("dark", np.full((h, w), 10, np.uint8)), ("white", np.full((h, w), 255, np.uint8)),
                      ("noise", np.random.default_rng(h + w).integers(0, 256, (h, w), dtype=np.uint8))):
        if tag in ("dark", "noise") and h * w > 3_000_000:
            continue   # (the oracle takes minutes on millions of points)
        want, info = oc.deskew(page)
        outs, ang = prlib_amd.deskew(torch.from_numpy(page).to(dev)[None])
        got = outs[0].cpu().numpy()
        ok = ang[0] == info["angle"] and got.shape == want.shape and np.array_equal(got, want)
        _, binary = oc.otsu(page)
        a2, n2 = prlib_amd.findAngle(torch.from_numpy(binary).to(dev), return_segments=True)
        ok = ok and (a2, n2) == oc.find_angle(binary)
        out[tag] = {"ok": bool(ok), "angle": float(ang[0]), "segments": int(info["n_lines"])}
    res[name] = out
bad = [(k, t) for k, v in res.items() for t, r in v.items() if not r["ok"]]
print(json.dumps({"shapes": res, "mismatches": bad}))
sys.exit(1 if bad else 0)
