"""Group kernel of HoughLinesP (ppht_group.hip) against k_ppht_mw and the oracle: segment lists and times.

    python tools/dbg/ppht_group_check.py [--mode child] ...   (the parent starts itself twice: PRL_HIP_PPHT_GROUP=1 / 0, hooks build)

Per case: the segments of prl_hip_houghp_device as a CRC-32 + count, the time of the second of two calls; then prl::deskew on
1 / 8 / 64 / 256 A4 text pages.  The parent compares the two runs and, for the small cases, the oracle.
"""
import argparse
import json
import os
import subprocess
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def cases():
    import numpy as np
    from prlib_amd import synth
    from oracle import capi as oc

    out = []
    img = np.zeros((200, 300), np.uint8)
    img[50, 20:280] = 255
    img[20:190, 150] = 255
    rng = np.random.default_rng(1)
    img[rng.integers(0, 200, 900), rng.integers(0, 300, 900)] = 128
    for thr, ll, gap in ((100, 100, 5), (60, 40, 3), (1000, 10, 2), (1, 5, 1)):
        out.append((f"lines_{thr}_{ll}_{gap}", img, thr, ll, gap, True))
    for (h, w, idx, skew) in ((300, 420, 5, 2.0), (390, 300, 21, -4.0), (700, 500, 3, 1.0), (1200, 900, 4, -2.5), (64, 64, 1, 0.0)):
        p = synth.text_page_numpy(h, w, idx, skew_deg=skew)
        _, binary = oc.otsu(p)
        out.append((f"text_{h}x{w}", 255 - binary, 100, int(round(w / 8.0)), 20, True))
    dense = (rng.integers(0, 256, (240, 320)) < 200).astype(np.uint8) * 255     # 78 % of the pixels are points
    out.append(("dense_240x320", dense, 100, 40, 20, True))
    for idx, skew in ((7, 1.5), (8, -3.0)):
        p = synth.text_page_numpy(3508, 2480, idx, skew_deg=skew)
        _, binary = oc.otsu(p)
        out.append((f"a4_{idx}", 255 - binary, 100, 310, 20, idx == 7))
    return out


def child(args):
    import numpy as np
    import torch
    import prlib_amd
    from prlib_amd import _capi, synth

    _capi.use_library(os.environ.get('PRL_LIB', _capi.HOOKS_LIB_PATH))
    dev = torch.device("cuda:0")
    res = {}
    for name, img, thr, ll, gap, _ in cases():
        t = torch.from_numpy(np.ascontiguousarray(img)).to(dev)
        seg = prlib_amd.houghp(t, thr, ll, gap)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        seg = prlib_amd.houghp(t, thr, ll, gap)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[name] = {"n": int(len(seg)), "crc": zlib.crc32(np.ascontiguousarray(seg).tobytes()), "ms": round(dt * 1e3, 2)}
        print(name, res[name], file=sys.stderr, flush=True)
    for n in args.pages:
        pages, _ = synth.text_pages_torch(n, 3508, 2480, dev, channels=1)
        torch.cuda.synchronize()
        prlib_amd.deskew(pages[:1])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs, ang = prlib_amd.deskew(pages)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[f"deskew_{n}"] = {"s": round(dt, 4), "angles_crc": zlib.crc32(np.asarray(ang, np.float64).tobytes())}
        print(f"deskew_{n}", res[f"deskew_{n}"], file=sys.stderr, flush=True)
        del pages, outs
    print("RESULT " + json.dumps(res))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="parent")
    ap.add_argument("--pages", type=int, nargs="*", default=[1, 8, 64])
    ap.add_argument("--oracle", type=int, default=1)
    ap.add_argument("--timeout", type=int, default=900)
    args = ap.parse_args()
    if args.mode == "child":
        return child(args)
    runs = {}
    for g in ("1", "0"):
        env = dict(os.environ, PRL_HIP_PPHT_GROUP=g, PRL_HIP_DEBUG="1")
        r = subprocess.run([sys.executable, __file__, "--mode", "child", "--pages"] + [str(p) for p in args.pages], capture_output=True, text=True,
                           timeout=args.timeout, env=env)
        sys.stderr.write(r.stderr[-6000:])
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
        if r.returncode != 0 or not line:
            print(json.dumps({"group": g, "rc": r.returncode, "stdout": r.stdout[-2000:]}))
            runs[g] = None
            continue
        runs[g] = json.loads(line[0][7:])
    bad = 0
    report = {}
    if runs["1"] and runs["0"]:
        for k in runs["1"]:
            a, b = runs["1"][k], runs["0"][k]
            same = all(a[f] == b[f] for f in a if f not in ("ms", "s"))
            bad += not same
            report[k] = {"same": same, "group": a, "mw": b}
    if args.oracle and runs["1"]:
        import numpy as np
        from oracle import capi as oc
        for name, img, thr, ll, gap, check in cases():
            if not check:
                continue
            want = oc.houghp(np.ascontiguousarray(img), thr, ll, gap)
            ok = runs["1"][name]["n"] == len(want) and runs["1"][name]["crc"] == zlib.crc32(np.ascontiguousarray(want).tobytes())
            report[name]["oracle"] = bool(ok)
            bad += not ok
    print(json.dumps({"bad": bad, "report": report}, indent=1))
    return 1 if bad or not runs["1"] else 0


if __name__ == "__main__":
    sys.exit(main())
