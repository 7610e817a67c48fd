"""Runs tests/cpp/test_vs_opencv (the oracle against REAL OpenCV / Leptonica) and returns its JSON
report, or None when the program cannot be built.  TEST INFRASTRUCTURE (used by tests/ and by bench.py's parity field);
without OpenCV on the machine the report is {"opencv": null, ...} and the oracle stays "parity unpinned"."""
from __future__ import annotations

import json
import os
import subprocess

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_DIR = os.path.join(_ROOT, "tests", "cpp")


def report(timeout: float = 600.0):
    from . import capi

    try:
        capi.build()
        subprocess.run(["make", "-C", _DIR, "-s", "test_vs_opencv"], check=True, capture_output=True, timeout=timeout)
        r = subprocess.run([os.path.join(_DIR, "test_vs_opencv")], capture_output=True, text=True, timeout=timeout)
        if r.returncode != 0:
            return None
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception:
        return None
