"""The C++ host layer (prlib_amd/csrc/prl/prl.h: prl::binarize*(cv::Mat&, cv::Mat&, ...), prl::denoise),
driven through tests/cpp/test_prl_host.cpp exactly like the reference's sample mains drive PRLib."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "test_prl_host")


def _build():
    import __graft_entry__ as ge

    ge.build_hip_library()
    from oracle import capi

    capi.build()
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "-s"], check=True)


def test_cpp_host_validation_and_loud_failure_without_device():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a device is present; the no-device behaviour is checked on the CPU box")
    _build()
    r = subprocess.run([BIN, "cpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_cpp_host_parity_on_device():
    if not os.path.exists(BIN):
        _build()
    r = subprocess.run([BIN, "gpu"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr


def _build_dropin():
    _build()
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "-s", "test_dropin_sample"], check=True)
    return os.path.join(ROOT, "tests", "cpp", "test_dropin_sample")


def test_dropin_headers_compile_the_sample_pattern_and_fail_loudly_without_device():
    """include/prl/binarizeSauvola.h ... denoiseNLM.h: a caller that keeps the reference's #include lines
    (samples/binarizations/binarizeSauvola_sample.cpp:25) builds with only -I include/prl."""
    import torch

    exe = _build_dropin()
    if torch.cuda.is_available():
        pytest.skip("a device is present; the no-device behaviour is checked on the CPU box")
    r = subprocess.run([exe, "cpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "dropin sample: OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_dropin_sample_defaults_on_device():
    exe = os.path.join(ROOT, "tests", "cpp", "test_dropin_sample")
    if not os.path.exists(exe):
        exe = _build_dropin()
    r = subprocess.run([exe, "gpu"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "dropin sample: OK" in r.stdout, r.stdout + r.stderr


def test_work_pool_many_callers():
    """prlib_amd/csrc/prl/work_pool.h (copy threads of the host-list entries): five callers inside parallel_for at once, every
    index run exactly once (plain C++, no device; tools/sanitize_cpu.sh runs the same program under ThreadSanitizer)."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "-s", "test_work_pool"], check=True)
    r = subprocess.run([os.path.join(os.path.join(ROOT, "tests", "cpp"), "test_work_pool")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "work_pool: OK" in r.stdout, r.stdout + r.stderr


def test_cpp_boundary_compiles_against_opencv_signatures():
    """The OpenCV-present branch of the C++ boundary (prl.h: PRL_HAVE_OPENCV).  No box of this pool has OpenCV, so that
    branch is compiled - syntax only - against tests/cpp/opencv_api/: declaration-only headers carrying OpenCV's real
    signatures for the members the host layer and its callers use (cv::Exception(int, const String&, const String&, const
    String&, int), MatStep / MatSize, _InputArray / _OutputArray, cv::cvtColor).  A use of anything only cvmat_shim.h offers
    fails here.  Also the converse: the shim's own Exception has no other public constructor."""
    cpp = os.path.join(ROOT, "tests", "cpp")
    r = subprocess.run(["make", "-C", cpp, "-s", "conformance"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    # the check discriminates: the one-string constructor the old shim offered is rejected by the same compile
    probe = ('#include "%s"\nvoid f() { throw cv::Exception("one string"); }\n'
             % os.path.join(ROOT, "prlib_amd", "csrc", "prl", "prl.h"))
    for inc in (["-I" + os.path.join(cpp, "opencv_api"), "-DPRL_REQUIRE_OPENCV"], []):   # OpenCV's API, then the shim
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-x", "c++", "-"] + inc, input=probe, capture_output=True, text=True)
        assert r.returncode != 0 and "Exception" in r.stderr, r.stderr
