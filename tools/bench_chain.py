#!/usr/bin/env python3
"""Secondary measurement: the device-resident part of BASELINE config 5 that is in scope so far —
prl::denoise -> prl::binarizeSauvola -> (invert) -> prl::thinZhangSuen — plus thinning alone.
deskew and backgroundNormalization (Leptonica-backed) are not built."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import prlib_amd
from prlib_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--pages", type=int, default=16)
ap.add_argument("--width", type=int, default=2480)
ap.add_argument("--height", type=int, default=3508)
ap.add_argument("--steps", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda:0")
gray = synth.pages_torch(a.pages, a.height, a.width, dev)
gen = torch.Generator(device=dev); gen.manual_seed(11)
bgr = (gray[..., None].float().expand(-1, -1, -1, 3) + torch.randn((a.pages, a.height, a.width, 3), device=dev, generator=gen) * 10.0).round_().clamp_(0, 255).to(torch.uint8).contiguous()


def timed(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.steps, r


px = a.pages * a.width * a.height
t_den, den = timed(lambda: prlib_amd.denoise(bgr, 10.0))
t_gray, g8 = timed(lambda: prlib_amd.cvtColorBGR2GRAY(den))
t_bin, mask = timed(lambda: prlib_amd.binarizeSauvola(g8, 31, 0.34, 0))
t_inv, inv = timed(lambda: prlib_amd.bitwise_not(mask))
t_thin_zs, sk = timed(lambda: prlib_amd.thinZhangSuen(inv))
t_thin_gh, _ = timed(lambda: prlib_amd.thinGuoHall(inv))
t_chain, sk2 = timed(lambda: prlib_amd.process_pages(bgr, 3, prlib_amd.SAUVOLA, 31, 0.34, 0, denoise_strength=10.0, thin=0))
t_chain_nd, _ = timed(lambda: prlib_amd.process_pages(bgr, 3, prlib_amd.SAUVOLA, 31, 0.34, 0, thin=0))
assert torch.equal(sk, sk2)
res = {"workload": f"{a.pages} x {a.width}x{a.height} pages, 1 GPU, device resident",
       "denoise_ms": round(t_den * 1e3, 2), "sauvola_w31_ms": round(t_bin * 1e3, 3),
       "thin_zhangsuen_ms": round(t_thin_zs * 1e3, 3), "thin_guohall_ms": round(t_thin_gh * 1e3, 3),
       "bgr2gray_ms": round(t_gray * 1e3, 3), "invert_ms": round(t_inv * 1e3, 3),
       "thin_zhangsuen_Mpx_s": round(px / t_thin_zs / 1e6, 1),
       "chain_one_call_ms": round(t_chain * 1e3, 2), "chain_Mpx_s": round(px / t_chain / 1e6, 1),
       "chain_without_denoise_ms": round(t_chain_nd * 1e3, 3), "chain_without_denoise_Mpx_s": round(px / t_chain_nd / 1e6, 1),
       "skeleton_fraction": round(float((sk > 0).float().mean()), 5)}
print(json.dumps(res))
