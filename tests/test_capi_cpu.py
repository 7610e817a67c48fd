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
