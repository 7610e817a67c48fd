"""Short randomised differential runs against the oracle (tools/fuzz_*.py, a few seconds each; the long runs are recorded in
profiles/r02/fuzz_*.txt).  A different seed every day, printed on failure."""
import json
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,seconds", [("fuzz_binarize.py", 8), ("fuzz_stages.py", 10), ("fuzz_chain.py", 8)])
def test_fuzz(tool, seconds, cuda_device):
    seed = int(time.time() // 86400) % 100000
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), "--seconds", str(seconds), "--seed", str(seed)],
                       capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and line, f"{tool} seed {seed}: {r.stdout[-2000:]} {r.stderr[-2000:]}"
    res = json.loads(line[-1])
    n = res.get("calls") or res.get("chain_calls")
    assert (sum(n.values()) if isinstance(n, dict) else n) > 20, res
