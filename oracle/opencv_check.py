"""Runs tests/cpp/test_vs_opencv (the oracle against REAL OpenCV / Leptonica) and returns its JSON
report.  TEST INFRASTRUCTURE (used by tests/ and by bench.py's parity field).  Without OpenCV on the machine the report is
{"opencv": null, "why": "headers_absent", ...} and the oracle stays "parity unpinned"; when OpenCV is there but the
program does not build, link or run, the report says so ("why": "build_failed" / "make_failed" / "run_failed" with the
compiler's or the program's stderr) instead of looking like a machine without OpenCV."""
from __future__ import annotations

import json
import os
import subprocess

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_DIR = os.path.join(_ROOT, "tests", "cpp")


def _tail(text: str, n: int = 1500) -> str:
    return (text or "")[-n:]


def report(timeout: float = 600.0):
    from . import capi

    try:
        capi.build()
        m = subprocess.run(["make", "-C", _DIR, "-s", "test_vs_opencv"], capture_output=True, text=True, timeout=timeout)
        if m.returncode != 0:
            return {"opencv": None, "leptonica": None, "why": "make_failed", "returncode": m.returncode, "stderr": _tail(m.stderr)}
        r = subprocess.run([os.path.join(_DIR, "test_vs_opencv")], capture_output=True, text=True, timeout=timeout)
        if r.returncode != 0:
            return {"opencv": None, "leptonica": None, "why": "run_failed", "returncode": r.returncode, "stderr": _tail(r.stderr)}
        rep = json.loads(r.stdout.strip().splitlines()[-1])
        err = os.path.join(_DIR, "test_vs_opencv.build_error")
        if rep.get("why") == "build_failed" and os.path.exists(err):
            rep["stderr"] = _tail(open(err).read())
        return rep
    except Exception as e:  # (timeouts, a missing compiler)
        return {"opencv": None, "leptonica": None, "why": "exception", "stderr": repr(e)}
