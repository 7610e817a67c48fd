"""The row loops of k_fused are frozen: on this kernel the ORDER of the instructions is worth +-20 %
(profiles/r03/sched_strategy_ab.txt), and an edit anywhere in binarize_fused.hip - even in the cold queue-push block - moves it
(tools/experiments/README.md, round 5).  This test compiles the file to gfx950 assembly (hipcc -S, ~1 min, no GPU) and compares the
opcode order of every threshold instantiation's row loops with the fingerprint of the measured build
(profiles/r06/k_fused_hot_loops.json = round 4's code, opcode for opcode, MINUS one v_readlane_b32 per row loop - see below -, plus the
identity of the compiler that produced it: another
hipcc orders instructions differently without any source change, and the test then skips instead of failing).  If it fails after an
intended change: measure the headline (python bench.py on a GPU box) and refresh the file with
`python tools/isa_budget.py --fingerprint > profiles/r06/k_fused_hot_loops.json`.  Round 6 added k_fused_exact (its own kernel, its own
instantiation of the integer loop): the threshold instantiations of k_fused were unchanged by that.  Refreshed once, in round 6: the early
exit of a flagged page's strips (one load in k_fused's prologue) made the register allocator drop the SGPR reload that sat in every row loop
("cosmetic") and moved nothing else; interleaved A/B on one box: profiles/r06/headline_ab_early_exit.txt, configs_ab_early_exit.txt.
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") and not shutil.which("hipcc"), reason="needs hipcc")
def test_k_fused_row_loops_are_the_measured_ones(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_budget

    want = json.load(open(os.path.join(ROOT, "profiles", "r06", "k_fused_hot_loops.json")))
    comp = want.pop("_compiler")
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")
    ver = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout
    if hashlib.sha1(ver.encode()).hexdigest() != comp["hipcc_version_sha1"]:
        pytest.skip("another compiler than the one the fingerprint was taken with (" + " / ".join(comp["first_lines"]) + "): re-measure, then refresh the file")
    asm = str(tmp_path / "fused.s")
    isa_budget.compile_asm(asm)
    got = isa_budget.hot_loop_fingerprints(asm)
    assert set(got) == set(want), sorted(set(got) ^ set(want))
    moved = [k for k in sorted(want) if got[k] != want[k]]
    assert not moved, "row loops whose instruction order changed: " + ", ".join(moved)
    # the headline instantiation is among them, with its 12 float32 and 2 integer loops
    assert want["k_fused<0,6,false>"]["row_loops"] == 14
