#!/usr/bin/env python3
"""Issue budget of a k_fused row loop, read from the ISA the compiler emits (VERDICT r4 "next" 3).

    python tools/isa_budget.py [--kernel 'k_fusedILi0ELi6ELb0EE'] [--asm /tmp/fused.s] [--ms 3.204 --pages 256 --size 4096 --window 31]

Compiles prlib_amd/csrc/binarize_fused.hip to gfx950 assembly (hipcc -S, the flags of prlib_amd/csrc/Makefile), finds the
innermost loops of the named kernel instantiation (a backward s_cbranch to a label), and classifies every instruction of each
loop by the pipe it issues on and - for the vector ALU - by the issue class measured on this chip
(profiles/r01/valu_issue_costs.txt: wave64 instructions cost ~2, ~4 or ~8 SIMD cycles).  Prints, per loop, the counts and the
cycle budget of each pipe per loop iteration (= one 512-column wavefront-row) and, with --ms, the measured cycles per
wavefront-row next to it: which pipe is how full, and the ceiling a perfect overlap of the four would reach.

The classes are an approximation (two- and four-cycle lists below are the measured ones; anything not listed is counted in
the four-cycle class and named in `unlisted`), and a loop body is taken as straight-line (rare side exits - the queue push -
are separate blocks and not counted).
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# measured issue classes (profiles/r01/valu_issue_costs.txt), SIMD cycles per wave64 instruction
TWO = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32",
       "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32", "v_not_b32",
       "v_cndmask_b32", "v_accvgpr_read_b32", "v_accvgpr_write_b32", "v_add_nc_u32", "v_readfirstlane_b32"}
EIGHT = {"v_sqrt_f32", "v_rcp_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32", "v_rcp_iflag_f32"}
SIXTEEN = {"v_sqrt_f64", "v_rcp_f64", "v_rsq_f64", "v_fma_f64", "v_mul_f64", "v_add_f64", "v_div_scale_f64", "v_div_fmas_f64", "v_div_fixup_f64"}
COST = {"valu2": 2.4, "valu4": 4.25, "valu8": 8.2, "valu16": 16.0}


def classify(op: str, text: str):
    base = op
    for suf in ("_e32", "_e64", "_sdwa", "_dpp", "_e64_dpp"):
        if base.endswith(suf):
            base = base[: -len(suf)]
    if op.startswith("v_"):
        if "dpp" in op or "sdwa" in op or " row_" in text or " quad_perm" in text or "wave_sh" in text or "row_bcast" in text or "sdwa" in text.lower() and "src0_sel" in text:
            return "valu4", base + "(dpp/sdwa)"
        if base in EIGHT:
            return "valu8", base
        if base in SIXTEEN or base.endswith("_f64"):
            return "valu16", base
        if base in TWO:
            # modifiers (neg / abs / clamp) on a float op cost a little more, still the two-cycle class
            return "valu2", base
        return "valu4", base
    if op.startswith("ds_"):
        return "lds", base
    if op.startswith(("buffer_", "tbuffer_", "global_", "flat_", "scratch_")):
        return "vmem", base
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_sleep"):
        return "wait", base
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem", base
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch", base
    if op.startswith("s_"):
        return "salu", base
    return "other", base


def branch_target(ln):
    m = re.match(r"^\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", ln)
    return m.group(1) if m else None


def row_loops(lines, min_sqrt=8):
    """Row loops of the kernel: loop headers (targets of backward branches) whose body holds the 8 v_sqrt_f32 of a row.
    Returns (head, last_back_edge, cold) per loop, `cold` = the line indices inside the loop that the usual iteration does not
    execute: nested loops (the queue push, `#pragma unroll 1`) and every region a forward branch skips that holds an atomic
    or a nested loop (the `if (__ballot(unsure))` block: taken for ~1 row in 10^4)."""
    label_at = {}
    for i, ln in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            label_at[m.group(1)] = i
    back = []
    for i, ln in enumerate(lines):
        t = branch_target(ln)
        if t in label_at and label_at[t] < i:
            back.append((label_at[t], i))
    heads = sorted({lo for lo, hi in back if sum("v_sqrt_f32" in l for l in lines[lo:hi + 1]) >= min_sqrt})
    # an outer loop (strip segments) around a row loop also qualifies: keep the innermost header of each nest
    heads = [h for h in heads if not any(h < o and max(b for a, b in back if a == o) <= max(b for a, b in back if a == h) for o in heads)]
    out = []
    for h in heads:
        hi = max(b for a, b in back if a == h)
        nested = [(a, b) for a, b in back if h < a and b <= hi]
        cold = set()
        for a, b in nested:
            cold.update(range(a, b + 1))
        for i in range(h, hi + 1):
            t = branch_target(lines[i])
            if t in label_at and i < label_at[t] <= hi:   # (a target past the last back edge is a loop exit, not a skip)
                region = lines[i + 1:label_at[t]]
                if any("atomic" in l for l in region) or any(i <= a and b < label_at[t] for a, b in nested):
                    cold.update(range(i + 1, label_at[t]))
        out.append((h, hi, cold))
    return out


def budget(lines, lo, hi, cold=()):
    counts = collections.Counter()
    names = collections.Counter()
    for i in range(lo, hi + 1):
        if i in cold:
            continue
        ln = lines[i].split(";")[0]
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*)$", ln)
        if not m or m.group(1).startswith("."):
            continue
        op, rest = m.group(1), m.group(2)
        cls, base = classify(op, " " + rest)
        counts[cls] += 1
        names[(cls, base)] += 1
    return counts, names


def hot_loop_mix(src, kernels, marker, min_marker):
    """Static instruction-class mix of the innermost loops that hold `marker` (the arithmetic of an inner loop whose trip
    counts dwarf everything else in the kernel) -> {kernel: {valu2, valu4, valu8, lds, vmem, salu, cycles_per_valu_instruction}}."""
    asm = "/tmp/prl_mix_" + os.path.basename(src) + ".s"
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
           "-Wno-pass-failed", "-Wno-unused-command-line-argument", "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-o", asm, src]
    subprocess.run(cmd, check=True)
    text = open(asm).read().splitlines()
    res = {}
    for kern in kernels:
        start = next(i for i, ln in enumerate(text) if re.match(r"^_Z\w*" + re.escape(kern) + r"\w*:", ln))
        end = next(i for i in range(start + 1, len(text)) if text[i].startswith(".Lfunc_end"))
        body = text[start:end]
        label_at = {m.group(1): i for i, ln in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", ln)] if m}
        back = [(label_at[t], i) for i, ln in enumerate(body) for t in [branch_target(ln)] if t in label_at and label_at[t] < i]
        inner = [lp for lp in back if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in back)]
        tot = collections.Counter()
        loops = []
        for lo, hi in inner:
            if sum(marker in ln for ln in body[lo:hi + 1]) < min_marker:
                continue
            c, _ = budget(body, lo, hi)
            tot.update(c)
            loops.append({"lines": [lo, hi], "valu2": c["valu2"], "valu4": c["valu4"], "valu8": c["valu8"], "lds": c["lds"], "vmem": c["vmem"]})
        nv = tot["valu2"] + tot["valu4"] + tot["valu8"]
        res[text[start].split(":")[0]] = {
            "loops": loops, "valu2": tot["valu2"], "valu4": tot["valu4"], "valu8": tot["valu8"], "lds": tot["lds"], "vmem": tot["vmem"],
            "salu": tot["salu"] + tot["branch"],
            "cycles_per_valu_instruction": round((tot["valu2"] * COST["valu2"] + tot["valu4"] * COST["valu4"] + tot["valu8"] * COST["valu8"]) / max(nv, 1), 3)}
    return res


def hot_loop_fingerprints(asm):
    """{kernel tag: sha1 of the opcode sequences of its row loops' usual paths} for every threshold instantiation of k_fused in
    an assembly file - what `--fingerprint` prints and tests/test_frozen_loop.py compares with profiles/r05/k_fused_hot_loops.json.
    Register numbers and labels are left out on purpose: what is frozen is the instruction ORDER (worth +-20 %)."""
    import hashlib
    text = open(asm).read().splitlines()
    res = {}
    for i, ln in enumerate(text):
        m = re.match(r"^(_Z\w*k_fusedILi(\d+)ELi(\d+)ELb(\d)EE\w*):", ln)
        if not m:
            continue
        end = next(j for j in range(i + 1, len(text)) if text[j].startswith(".Lfunc_end"))
        body = text[i:end]
        h = hashlib.sha1()
        n_loops = 0
        for lo, hi, cold in row_loops(body, min_sqrt=1):
            n_loops += 1
            for k in range(lo, hi + 1):
                if k in cold:
                    continue
                mm = re.match(r"^\s+([a-z_0-9]+)\s", body[k].split(";")[0] + " ")
                if mm and not mm.group(1).startswith("."):
                    h.update(mm.group(1).encode() + b"\n")
            h.update(b"--\n")
        if n_loops:
            res[f"k_fused<{m.group(2)},{m.group(3)},{'true' if m.group(4) == '1' else 'false'}>"] = {"row_loops": n_loops, "sha1": h.hexdigest()}
    return res


def compile_asm(out):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
           "-Wno-pass-failed", "-Wno-unused-command-line-argument", "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-o", out,
           os.path.join(ROOT, "prlib_amd", "csrc", "binarize_fused.hip")]
    subprocess.run(cmd, check=True)
    return out


def main():
    if "--fingerprint" in sys.argv:   # python tools/isa_budget.py --fingerprint [file.s] : the frozen row loops' opcode order, hashed
        i = sys.argv.index("--fingerprint")
        asm = sys.argv[i + 1] if len(sys.argv) > i + 1 else compile_asm("/tmp/prl_fused_isa.s")
        import hashlib
        fp = hot_loop_fingerprints(asm)
        ver = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True).stdout
        fp["_compiler"] = {"hipcc_version_sha1": hashlib.sha1(ver.encode()).hexdigest(), "first_lines": ver.strip().splitlines()[:2]}
        print(json.dumps(fp, indent=1, sort_keys=True))
        return
    if "--mix" in sys.argv:   # python tools/isa_budget.py --mix nlm : the NL-means kernels' inner-loop mix (tools/bench_denoise.py reads it)
        which = sys.argv[sys.argv.index("--mix") + 1]
        assert which == "nlm"
        out = {"source": "static instruction mix of the innermost loops holding v_dot4_u32_u8 (the 441-offset loops), hipcc -S of prlib_amd/csrc/nlm.hip; "
                         "issue classes of profiles/r01/valu_issue_costs.txt (2.4 / 4.25 / 8.2 SIMD cycles per wave64 instruction)",
               "kernels": hot_loop_mix(os.path.join(ROOT, "prlib_amd", "csrc", "nlm.hip"), ["k_nlm_y_xlILi1ELb1EE", "k_nlm_y_xlILi2ELb1EE"], "v_dot4", 20)}
        print(json.dumps(out, indent=1))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="k_fusedILi0ELi6ELb0EE", help="substring of the mangled kernel name (default: Sauvola, shift 6, not wide = the headline)")
    ap.add_argument("--asm", default="", help="an existing hipcc -S output of binarize_fused.hip (default: compile now, ~1 min)")
    ap.add_argument("--ms", type=float, default=0.0, help="measured k_fused launch duration")
    ap.add_argument("--pages", type=int, default=256)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--window", type=int, default=31)
    ap.add_argument("--ghz", type=float, default=2.4, help="shader clock to turn --ms into cycles (rocm-smi under sustained load: 2.04-2.08)")
    ap.add_argument("--gui-active", type=float, default=0.0, help="GRBM_GUI_ACTIVE per launch from a rocprofv3 --pmc pass (summed over the 8 XCDs): "
                    "the launch's duration in shader cycles without assuming a clock; overrides --ghz")
    ap.add_argument("--only", default="", help="substring of `kind` (e.g. 'interior') to keep")
    a = ap.parse_args()
    asm = a.asm
    if not asm:
        asm = "/tmp/prl_fused_isa.s"
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
               "-Wno-pass-failed", "-Wno-unused-command-line-argument", "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-o", asm,
               os.path.join(ROOT, "prlib_amd", "csrc", "binarize_fused.hip")]
        subprocess.run(cmd, check=True)
    text = open(asm).read().splitlines()
    start = next(i for i, ln in enumerate(text) if re.match(r"^_Z\w*" + re.escape(a.kernel) + r"\w*:", ln))
    end = next(i for i in range(start + 1, len(text)) if text[i].startswith(".Lfunc_end"))
    body = text[start:end]
    meta = {}
    for ln in text[end:end + 80]:
        m = re.match(r"^\s*;\s*(NumVgprs|NumAgprs|NumSgprs|Occupancy|ScratchSize|SGPRBlocks|VGPRBlocks|sgpr_spill_count|vgpr_spill_count)\s*[:=]?\s*(\d+)", ln)
        if m:
            meta[m.group(1)] = int(m.group(2))
    out = {"kernel": text[start].split(":")[0], "registers": meta, "loops": []}
    for lo, hi, cold in row_loops(body):
        counts, names = budget(body, lo, hi, cold)
        n = sum(counts.values())
        valu = {k: counts.get(k, 0) for k in ("valu2", "valu4", "valu8", "valu16")}
        valu_cycles = sum(valu[k] * COST[k] for k in valu)
        loads = {b: k for (c, b), k in names.items() if c == "vmem"}
        kind = ("float32 loop, interior strip (typed loads)" if any("format" in b for b in loads) else
                "float32 loop, edge strip (byte loads, clamped)" if counts.get("lds", 0) == 14 else "integer loop")
        rec = {"label": body[lo].split(":")[0], "kind": kind, "lines": [lo, hi], "instructions_on_the_usual_path": n, "valu": valu,
               "valu_total": sum(valu.values()),
               "salu": counts.get("salu", 0) + counts.get("branch", 0), "smem": counts.get("smem", 0), "lds": counts.get("lds", 0), "vmem": counts.get("vmem", 0),
               "waitcnt": counts.get("wait", 0),
               "valu_issue_cycles_per_row": round(valu_cycles, 1),
               # the four SIMDs of a CU share one LDS pipe and one vector-memory pipe: a wave64 ds_bpermute occupies the LDS pipe ~6 CU
               # cycles (measured with all four SIMDs issuing), a wave64 VMEM instruction the address path 16 - 25 cycles (typed /
               # dwordx2 loads: 64 lanes at 4 per cycle, more when a request straddles lines) - x4: the other three SIMDs do the same
               "lds_pipe_cycles_per_row_x4_simds": counts.get("lds", 0) * 6 * 4,
               "vmem_pipe_cycles_per_row_x4_simds": [counts.get("vmem", 0) * 16 * 4, counts.get("vmem", 0) * 25 * 4],
               "by_name": {f"{c}:{b}": k for (c, b), k in sorted(names.items(), key=lambda kv: -kv[1]) if c.startswith("valu") or c in ("lds", "vmem")}}
        if a.ms:
            # wavefront-rows of the launch: pages x output rows x strips per row (each strip = one wavefront's 512 padded columns,
            # 512 - (w - 2) output columns)
            uo = 512 - (a.window - 2)
            uo -= uo % 8
            n_strips = -(-(a.size - 1) // uo)
            rows = a.pages * (a.size - 1) * n_strips
            simds = 256 * 4
            cyc = (a.gui_active / 8.0 if a.gui_active else a.ms * 1e-3 * a.ghz * 1e9) * simds / rows
            rec["measured"] = {"launch_ms": a.ms, "wavefront_rows": rows, "strips_per_row": n_strips, "simd_cycles_per_row": round(cyc, 1),
                               "valu_pipe_fill": round(valu_cycles / cyc, 3),
                               "lds_pipe_fill": round(counts.get("lds", 0) * 6 * 4 / cyc, 3),
                               "vmem_pipe_fill": [round(counts.get("vmem", 0) * 16 * 4 / cyc, 3), round(counts.get("vmem", 0) * 25 * 4 / cyc, 3)],
                               "ceiling_if_valu_alone": round(cyc / valu_cycles, 3)}
        if a.only in rec["kind"]:
            out["loops"].append(rec)
    # The float32 loops are instantiated per lane offset LO = (w - 1) / 8 (0..3), three each in source order: edge strip, interior
    # strip, interior strip with every lane full (FAST: the usual case).  Summary for --window: the wavefront-rows of a page row
    # are (strips - 2) interior FAST strips and 2 edge strips.
    flt = [r for r in out["loops"] if r["kind"].startswith("float32")]
    if a.ms and len(flt) == 12 and not a.only and a.window <= 31:
        lo_idx = min((a.window - 1) // 8, 3)
        edge, _, fast = flt[3 * lo_idx:3 * lo_idx + 3]
        ns = fast["measured"]["strips_per_row"]
        cyc_meas = fast["measured"]["simd_cycles_per_row"]
        wavg = lambda k: ((ns - 2) * fast[k] + 2 * edge[k]) / ns   # noqa: E731
        valu = wavg("valu_issue_cycles_per_row")
        out["headline_summary"] = {
            "window": a.window, "lane_offset": lo_idx, "interior_loop": fast["label"], "edge_loop": edge["label"], "strips_per_row": ns,
            "cycles_from": (f"GRBM_GUI_ACTIVE {a.gui_active:.5g} / 8 XCDs = {a.gui_active / 8:.4g} cycles per launch (= {a.gui_active / 8 / (a.ms * 1e-3) / 1e9:.3f} GHz over {a.ms} ms)"
                            if a.gui_active else f"{a.ms} ms at an assumed {a.ghz} GHz"),
            "measured_simd_cycles_per_wavefront_row": cyc_meas,
            "per_wavefront_row_weighted": {
                "valu_instructions": round(wavg("valu_total"), 1), "valu_issue_cycles": round(valu, 1),
                "salu_and_waitcnt_instructions": round(wavg("salu") + wavg("waitcnt"), 1),
                "lds_instructions": round(wavg("lds"), 1), "lds_pipe_cycles_x4_simds": round(wavg("lds") * 24, 1),
                "vmem_instructions": round(wavg("vmem"), 1), "vmem_pipe_cycles_x4_simds": [round(wavg("vmem") * 64, 1), round(wavg("vmem") * 100, 1)]},
            "pipe_fill": {"valu": round(valu / cyc_meas, 3), "lds": round(wavg("lds") * 24 / cyc_meas, 3),
                          "vmem": [round(wavg("vmem") * 64 / cyc_meas, 3), round(wavg("vmem") * 100 / cyc_meas, 3)]},
            "speedup_if_only_valu_issue_remained": round(cyc_meas / valu, 3)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
