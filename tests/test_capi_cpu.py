"""CPU-side checks of the drop-in boundary: libprlib_hip.so loads, exports every symbol the public
header declares, validates arguments like the reference, and fails loudly without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported(prl):
    from prlib_amd import _capi

    header = open(os.path.join(ROOT, "include", "prl_hip.h")).read()
    declared = set(re.findall(r"\b(prl_hip_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found in include/prl_hip.h"
    assert declared == set(_capi.EXPORTED_SYMBOLS), declared ^ set(_capi.EXPORTED_SYMBOLS)
    L = _capi.lib()
    for name in declared:
        assert hasattr(L, name), f"{name} is declared in the header but not exported"
    assert L.prl_hip_abi_version() == 4


def test_struct_layout_matches_header(prl):
    from prlib_amd import _capi

    assert C.sizeof(_capi.BinarizeParams) == 56
    assert _capi.BinarizeParams.k.offset == 8 and _capi.BinarizeParams.feng_alpha1.offset == 24
    assert C.sizeof(_capi.BinarizeGeometry) == 24
    assert C.sizeof(_capi.BinarizeStats) == 64
    from prlib_amd.deskew import DeskewStats   # prl_deskew_stats: 8 x 64-bit

    assert C.sizeof(DeskewStats) == 64 and DeskewStats.min_page_headroom.offset == 48


def test_defaults_are_the_reference_headers(prl):
    # binarizeSauvola.h:43-47, binarizeNiblack.h:43-47, binarizeWolfJolion.h:43-47, binarizeNICK.h:43-47, binarizeFeng.h:46-53
    for m in (prl.SAUVOLA, prl.NIBLACK, prl.WOLFJOLION):
        p = prl.default_params(m)
        assert (p.window_size, p.k, p.morph_iterations) == (101, 0.01, 2)
    p = prl.default_params(prl.NICK)
    assert (p.window_size, p.k, p.morph_iterations) == (21, -0.01, 0)
    p = prl.default_params(prl.FENG)
    assert (p.window_size, p.feng_alpha1, p.feng_k1, p.feng_k2, p.feng_gamma, p.morph_iterations) == (21, 0.75, 0.2, 0.03, 2.0, 2)


@pytest.mark.parametrize("method", range(5))
@pytest.mark.parametrize("w,h,win", [(4096, 4096, 31), (2480, 3508, 101), (100, 100, 101), (60, 80, 101), (33, 200, 15)])
def test_geometry_agrees_with_oracle(prl, oracle, method, w, h, win):
    po = oracle.make_params(method, win, 0.1, 0)
    st_o, go = oracle.geometry(po, w, h)
    pp = prl.make_params(method, win, 0.1, 0)
    from prlib_amd import _capi

    g = _capi.BinarizeGeometry()
    st_p = _capi.lib().prl_hip_binarize_geometry(C.byref(pp), w, h, C.byref(g))
    assert st_p == st_o
    assert (g.w, g.half, g.padded_w, g.padded_h, g.out_w, g.out_h) == (go.w, go.half, go.padded_w, go.padded_h, go.out_w, go.out_h)


def test_reference_quirks_in_geometry(prl):
    # Sauvola/Niblack output (W-1)x(H-1); Wolf/NICK/Feng (W-w)x(H-w)
    g = prl.geometry(prl.make_params(prl.SAUVOLA, 31), 4096, 4096)
    assert (g.out_w, g.out_h, g.padded_w) == (4095, 4095, 4126)
    g = prl.geometry(prl.make_params(prl.NICK, 21), 2480, 3508)
    assert (g.out_w, g.out_h) == (2459, 3487)
    # clamped even window keeps the page size (Appendix D.7)
    g = prl.geometry(prl.make_params(prl.NIBLACK, 101), 100, 100)
    assert (g.w, g.half, g.out_w, g.out_h) == (100, 50, 100, 100)


def test_errors_without_touching_a_device(prl):
    from prlib_amd import _capi

    page = np.zeros((50, 60), np.uint8)
    with pytest.raises(ValueError, match="Window size must satisfy"):
        prl.binarizeSauvola(page, 30)
    with pytest.raises(ValueError, match="empty"):
        prl.binarizeNiblack(np.zeros((0, 0), np.uint8))
    with pytest.raises(_capi.PrlError) as e:
        prl.binarizeFeng(page, 51)
    assert e.value.status == _capi.PRL_ERR_EMPTY_RECT


def test_no_device_means_loud_failure_not_cpu_fallback(prl):
    import torch
    from prlib_amd import _capi

    if torch.cuda.is_available():
        pytest.skip("device present")
    page = np.full((64, 64), 200, np.uint8)
    with pytest.raises(_capi.PrlError) as e:
        prl.binarizeSauvola(page, 15, 0.34, 0)
    assert e.value.status == _capi.PRL_ERR_NO_DEVICE
    with pytest.raises(_capi.PrlError) as e:
        prl.denoise(np.zeros((32, 32, 3), np.uint8))
    assert e.value.status == _capi.PRL_ERR_NO_DEVICE


def test_product_does_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing under prlib_amd/ may import, include or link it."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "prlib_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"#include\s+[\"<][^\">]*oracle|from oracle|import oracle|lprl_oracle|prl_oracle_", txt):
                    bad.append(os.path.join(base, f))
    assert not bad, bad


def test_library_exports_exactly_the_header():
    """libprlib_hip.so exports the functions include/prl_hip.h declares and nothing else (the reference marks exactly its public
    functions CV_EXPORTS: binarizeSauvola.h:43); the test-hooks build adds the four prl_hip_internal_* entries; both carry the ABI
    version in their SONAME.  The linker maps are generated from the header (tools/gen_export_map.py)."""
    import shutil
    import subprocess
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_export_map as gem

    assert subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_export_map.py"), "--check"]).returncode == 0
    declared = gem.header_functions()
    assert len(declared) == len(set(declared)) >= 56
    from prlib_amd import _capi

    assert sorted(_capi.EXPORTED_SYMBOLS) == sorted(declared)
    if not shutil.which("nm") or not shutil.which("readelf"):
        pytest.skip("binutils not installed")
    abi = re.search(r"#define PRL_HIP_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "prl_hip.h")).read()).group(1)
    for path, extra, soname in ((_capi.LIB_PATH, [], "libprlib_hip.so." + abi),
                                (_capi.HOOKS_LIB_PATH, gem.INTERNAL, "libprlib_hip_testhooks.so." + abi)):
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        got = sorted(ln.split()[-1].split("@")[0] for ln in out.splitlines() if ln.split() and ln.split()[-2] in "TtWwDdBbRr")
        got = [g for g in got if g != "PRLIB_HIP_" + abi]   # (the version node itself is an absolute symbol)
        assert got == sorted(declared + extra), (path, sorted(set(got) ^ set(declared + extra)))
        dyn = subprocess.run(["readelf", "-d", path], capture_output=True, text=True, check=True).stdout
        assert f"[{soname}]" in dyn, dyn


def test_cmake_install_tree_builds_the_dropin_sample(tmp_path):
    """The CMake route (what a PRLib maintainer uses: the reference builds with CMake, CMakeLists.txt:23-33 there): configure, build,
    install into a scratch prefix; the installed library exports the header's functions only and carries its SONAME; a caller that
    keeps the reference's #include lines (tests/cpp/test_dropin_sample.cpp) builds against the INSTALLED tree alone."""
    import shutil
    import subprocess

    if not shutil.which("cmake") or not os.path.exists("/opt/rocm/bin/hipcc") or not shutil.which("g++"):
        pytest.skip("cmake / hipcc / g++ not installed")
    b, prefix = str(tmp_path / "b"), str(tmp_path / "prefix")
    for cmd in (["cmake", "-S", ROOT, "-B", b, "-DCMAKE_INSTALL_PREFIX=" + prefix], ["cmake", "--build", b, "-j", str(min(8, os.cpu_count() or 1))],
                ["cmake", "--install", b]):
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, " ".join(cmd) + "\n" + r.stdout[-3000:] + r.stderr[-3000:]
    abi = re.search(r"#define PRL_HIP_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "prl_hip.h")).read()).group(1)
    so = os.path.join(prefix, "lib", "libprlib_hip.so." + abi)
    assert os.path.exists(so) and os.path.exists(os.path.join(prefix, "lib", "libprlib_hip_host.a"))
    assert os.path.exists(os.path.join(prefix, "lib", "cmake", "prlib_hip", "prlib_hipConfig.cmake"))
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    names = sorted(ln.split()[-1].split("@")[0] for ln in out.splitlines() if ln.split() and ln.split()[-2] in "TtWw")
    from prlib_amd import _capi

    assert names == sorted(_capi.EXPORTED_SYMBOLS)
    from oracle import capi as oc

    oc.build()   # (the sample checks itself against the oracle when it runs on a GPU box; here it only has to build)
    exe = str(tmp_path / "dropin")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-I" + os.path.join(prefix, "include", "prl"),
                        os.path.join(ROOT, "tests", "cpp", "test_dropin_sample.cpp"), "-L" + os.path.join(prefix, "lib"), "-lprlib_hip_host", "-lprlib_hip",
                        "-L" + os.path.join(ROOT, "oracle"), "-lprl_oracle", "-Wl,-rpath," + os.path.join(prefix, "lib"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
                        "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0 and os.path.exists(exe), r.stderr[-3000:]
