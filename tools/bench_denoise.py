#!/usr/bin/env python3
"""Secondary measurement (BASELINE config 4): prl::denoise on N x 4096^2 x 3 noisy scans, 1 GPU."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pmc_pass(argv):
    """Counters of the NL-means kernels of THIS workload, measured now: one child run of this script under
    `rocprofv3 --pmc` per counter set (counters only, the program itself after `--`), before this process touches the GPU.
    -> {kernel: {counter: per-launch average}} or (None, why)."""
    import csv, glob, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None, "rocprofv3 not found"
    acc = {}
    for cset in (["SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_BUSY_CYCLES"], ["SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE"]):
        tmp = tempfile.mkdtemp(prefix="prl_pmc_nlm_", dir="/tmp")
        cmd = [exe, "--pmc"] + cset + ["--output-format", "csv", "-d", tmp, "--", sys.executable, os.path.abspath(__file__)] + argv + \
              ["--pmc", "0", "--steps", "1", "--cpu-seconds", "0", "--check", "0"]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=400)
            rows = 0
            for f in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        k = row.get("Kernel_Name", "")
                        if "k_nlm" in k:
                            name = k[k.index("k_nlm"):].split("(")[0]
                            acc.setdefault(name, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
                            rows += 1
            if not rows:
                return None, f"{cset[0]}: no k_nlm rows (rc {r.returncode}): {(r.stderr or '')[-200:]}"
        except Exception as e:   # (timeout, profiler refused)
            return None, f"{cset[0]}: {e!r}"
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    # the child runs the batch twice (the warm-up call and one timed step) and a batch may be several launches (chunks of pages):
    # per batch = the sum over every launch / 2
    return {k: {c: sum(v) / 2.0 for c, v in cs.items()} for k, cs in acc.items()}, "rocprofv3 --pmc, two passes, this run; per batch = sum over the launches / 2 batches"


_pmc = None
if "--pmc" in sys.argv and sys.argv[sys.argv.index("--pmc") + 1] == "1":
    _argv = [x for i, x in enumerate(sys.argv[1:]) if x != "--pmc" and sys.argv[i] != "--pmc"]
    _pmc = pmc_pass(_argv)   # child processes, before torch is imported here

import torch
import prlib_amd
from prlib_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--pages", type=int, default=8)
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--strength", type=float, default=10.0)
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--check", type=int, default=0)
ap.add_argument("--pmc", type=int, default=0, help="1: measure the instruction / LDS counters of this workload in child rocprofv3 --pmc passes (else the fractions "
                "use the committed counts of profiles/r02/pmc_nlm.txt, labelled as constants)")
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
# NL-means is not HBM-bound: the roofs that matter are the LDS pipe and vector-ALU issue.
#   counts   : wave64 instructions and LDS-active cycles of the two kernels (L plane, ab planes), per launch, summed over the chip -
#              measured in THIS run with --pmc 1, else the committed profile of the same kernels (8 x 4096^2: profiles/r02/pmc_nlm.txt),
#              scaled by the pixel count and labelled as a constant
#   VALU peak: class-weighted.  A SIMD issues a wave64 vector instruction in ~2.4, ~4.25 or ~8.2 cycles depending on its class
#              (profiles/r01/valu_issue_costs.txt); the kernels' inner loops are 51 % / 36 % two-cycle instructions
#              (profiles/r05/nlm_issue_mix.json, tools/isa_budget.py --mix nlm), i.e. 3.30 / 3.59 cycles per instruction on average.
#              (Round 4 divided by one instruction per 4 cycles, which made a constant read "1.0".)
#   clock    : GRBM_GUI_ACTIVE / kernel time when measured, else the 2.4 GHz of the data sheet (the chip holds less under load,
#              which makes the constant-based fractions lower bounds)
mix = json.load(open(os.path.join(ROOT, "profiles", "r05", "nlm_issue_mix.json")))["kernels"]
cpi = {("k_nlm_y_xl<1" if "ILi1E" in k else "k_nlm_y_xl<2"): v["cycles_per_valu_instruction"] for k, v in mix.items()}
CONST = {"k_nlm_y_xl<1": {"SQ_INSTS_VALU": 1.2414e10, "SQ_INSTS_LDS": 2.7452e9, "SQ_LDS_IDX_ACTIVE": 9.8972e9},
         "k_nlm_y_xl<2": {"SQ_INSTS_VALU": 1.6972e10, "SQ_INSTS_LDS": 3.8617e9, "SQ_LDS_IDX_ACTIVE": 1.3287e10}}
scale = px / (8 * 4096 * 4096)
measured = _pmc is not None and _pmc[0] is not None
counts = {}
for key in CONST:
    got = next((v for k, v in (_pmc[0].items() if measured else []) if k.startswith(key)), None)
    counts[key] = got if got else {c: v * scale for c, v in CONST[key].items()}
valu_cycles = sum(counts[k]["SQ_INSTS_VALU"] * cpi[k] for k in counts)          # SIMD cycles of vector issue, whole chip
valu_instr = sum(counts[k]["SQ_INSTS_VALU"] for k in counts)
lds_cycles = sum(counts[k]["SQ_LDS_IDX_ACTIVE"] for k in counts)
clock = 2.4e9
if measured and all("GRBM_GUI_ACTIVE" in counts[k] for k in counts):
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs: / 8 = cycles the kernels ran; against the wall time of one batch
    clock = sum(counts[k]["GRBM_GUI_ACTIVE"] for k in counts) / 8 / dt
res["alu_roofline"] = {"bound": "lds + valu issue",
                       "counts": "measured (" + _pmc[1] + ")" if measured else "CONSTANT: profiles/r02/pmc_nlm.txt scaled by pixels" + (f" (pmc pass failed: {_pmc[1]})" if _pmc else ""),
                       "valu_wave_instr_per_px": round(valu_instr / px, 1),
                       "valu_cycles_per_instr_class_weighted": {k: cpi[k] for k in cpi},
                       "clock_GHz": round(clock / 1e9, 3),
                       "valu_frac": round(valu_cycles / (1024 * clock * dt), 3),             # of 1024 SIMDs issuing every cycle
                       "valu_frac_if_every_instruction_took_4_cycles": round(valu_instr * 4 / (1024 * clock * dt), 3),
                       "lds_active_cycles_per_px": round(lds_cycles / px, 1),
                       "lds_frac": round(lds_cycles / (256 * clock * dt), 3)}               # of 256 CUs' LDS busy every cycle
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
