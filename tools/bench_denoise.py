#!/usr/bin/env python3
"""Secondary measurement (BASELINE config 4): prl::denoise on N x 4096^2 x 3 noisy scans, 1 GPU."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import prlib_amd
from prlib_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--pages", type=int, default=8)
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--strength", type=float, default=10.0)
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--check", type=int, default=0)
ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg (oracle NL-means on crops of the same scans, all host cores; 0 = skip)")
a = ap.parse_args()
dev = torch.device("cuda:0")
gray = synth.pages_torch(a.pages, a.size, a.size, dev)
gen = torch.Generator(device=dev); gen.manual_seed(7)
img = gray[..., None].float().expand(-1, -1, -1, 3) + torch.randn((a.pages, a.size, a.size, 3), device=dev, generator=gen) * 15.0
img = img.round_().clamp_(0, 255).to(torch.uint8).contiguous()
out = torch.empty_like(img)
prlib_amd.denoise(img, a.strength, out=out); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    prlib_amd.denoise(img, a.strength, out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
px = a.pages * a.size * a.size
res = {"workload": f"prl::denoise strength={a.strength} on {a.pages} x {a.size}^2 x 3", "ms_per_batch": round(dt * 1e3, 2),
       "Mpixels/s": round(px / dt / 1e6, 1), "algorithmic_GB/s (6 B/px)": round(6 * px / dt / 1e9, 2),
       "frac_of_8TB/s": round(6 * px / dt / 8e12, 5)}
# NL-means is not HBM-bound: the roofs that matter are the LDS pipe and vector-ALU issue.  Counts from the committed counter
# profile of these kernels (profiles/r02/pmc_nlm.txt, 8 x 4096^2, summed over the chip): SQ_INSTS_VALU 1.241e10 + 1.697e10
# wavefront instructions for the L and ab planes, SQ_INSTS_LDS 2.75e9 + 3.86e9, SQ_LDS_IDX_ACTIVE 9.90e9 + 1.33e10 cycles =
# 87 % / 78 % of each kernel's cycles on every CU; peak issue = one wave64 vector instruction per 4 cycles and SIMD (what
# k_fused's counters show on this chip), 1024 SIMDs, 2.4 GHz; LDS roof = every CU's LDS busy every cycle.
VALU_WAVE_INSTR_PER_PX = (1.2414e10 + 1.6972e10) / (8 * 4096 * 4096)
LDS_WAVE_INSTR_PER_PX = (2.7452e9 + 3.8617e9) / (8 * 4096 * 4096)
LDS_ACTIVE_CYCLES_PER_PX = (9.8972e9 + 1.3287e10) / (8 * 4096 * 4096)
peak_valu = 1024 * 2.4e9 / 4
res["alu_roofline"] = {"bound": "lds", "valu_wave_instr_per_px": round(VALU_WAVE_INSTR_PER_PX, 1),
                       "lds_wave_instr_per_px": round(LDS_WAVE_INSTR_PER_PX, 1),
                       "achieved_valu_Ginstr_s": round(VALU_WAVE_INSTR_PER_PX * px / dt / 1e9, 1),
                       "peak_valu_Ginstr_s": round(peak_valu / 1e9, 1),
                       "valu_frac": round(VALU_WAVE_INSTR_PER_PX * px / dt / peak_valu, 3),
                       "lds_active_cycles_per_px": round(LDS_ACTIVE_CYCLES_PER_PX, 1),
                       "lds_frac": round(LDS_ACTIVE_CYCLES_PER_PX * px / dt / (256 * 2.4e9), 3),   # of 256 CUs x 2.4 GHz
                       "source": "profiles/r02/pmc_nlm.txt"}
if a.cpu_seconds > 0:
    # CPU baseline beside the number: the oracle's prl::denoise (oracle/prl_oracle_nlm.c, OpenMP over rows) on crops
    # of the same scans with every host core, for about --cpu-seconds; extrapolation: NL-means costs the same per pixel everywhere
    from oracle import capi as oc
    cores = os.cpu_count() or 1
    side = 1024
    crop = img[0, :side, :side].cpu().numpy().copy()
    t0 = time.perf_counter(); oc.denoise(crop, a.strength, threads=cores); t1 = time.perf_counter() - t0
    reps, tn = 1, t1
    while tn < a.cpu_seconds and reps < 64:
        crop = img[reps % a.pages, :side, :side].cpu().numpy().copy()
        t0 = time.perf_counter(); oc.denoise(crop, a.strength, threads=cores); tn += time.perf_counter() - t0
        reps += 1
    res["cpu_baseline"] = {"value": round(reps * side * side / tn / 1e6, 3), "unit": "Mpixels/s", "cores": cores, "kind": "port",
                           "sample": f"{reps} crops of {side}x{side}x3 of the benchmark's scans, oracle/prl_oracle_nlm.c with {cores} OpenMP threads, {tn:.1f} s"}
    res["speedup_vs_cpu_baseline"] = round(res["Mpixels/s"] / max(res["cpu_baseline"]["value"], 1e-9), 1)
if a.check:
    from oracle import capi as oc
    sub = img[0, :256, :256].cpu().numpy().copy()
    got = prlib_amd.denoise(torch.from_numpy(sub).to(dev), a.strength).cpu().numpy()
    res["mismatch_256x256_vs_oracle"] = int((got != oc.denoise(sub, a.strength, threads=os.cpu_count())).sum())
print(json.dumps(res))
