"""Host logic: the float32 decision margin E1 the fused kernel relies on (fused_bounds() in
prlib_amd/csrc/binarize_fused.hip) really bounds the float32 evaluation error.

The kernel's float32 sequence is emulated with numpy float32 (fma through float64, which is exact for
24-bit operands up to one final rounding) on window sums of random real windows, and compared with the
exact-arithmetic threshold computed in float64 from the same integers (error ~1e-13, negligible here)."""
import ctypes as C

import numpy as np
import pytest

SAUVOLA, NIBLACK, WOLFJOLION, NICK, FENG = range(5)
_HOOKS = None


def _hooks():
    """libprlib_hip_testhooks.so: the product library exports only what include/prl_hip.h declares; the prl_hip_internal_* entries
    (host-side helpers of the fused pipeline, no device needed) live in the test-hooks build."""
    global _HOOKS
    if _HOOKS is None:
        from prlib_amd import _capi

        _HOOKS = C.CDLL(_capi.HOOKS_LIB_PATH)
    return _HOOKS


def _bounds(prl, method, w, k, size=4096):
    from prlib_amd import _capi

    L = _hooks()
    L.prl_hip_internal_fused_bounds.argtypes = [C.POINTER(_capi.BinarizeParams), C.c_int, C.c_int, C.POINTER(C.c_double)]
    out = (C.c_double * 8)()
    p = prl.make_params(method, w, k, 0)
    assert L.prl_hip_internal_fused_bounds(C.byref(p), size, size, out) == 0
    return dict(Em=out[0], Eq=out[1], vthr=out[2], E1=out[3], Elit=out[4], eps1=out[5], kappa=out[6])


def _window_sums(rng, w, n):
    npx = (w - 1) ** 2
    kinds = rng.integers(0, 4, n)
    S = np.empty(n, np.int64)
    Q = np.empty(n, np.int64)
    for i, kind in enumerate(kinds):
        if kind == 0:
            px = rng.integers(0, 256, npx)
        elif kind == 1:
            px = np.clip(rng.normal(rng.integers(20, 240), rng.integers(1, 40), npx), 0, 255).round().astype(np.int64)
        elif kind == 2:
            px = np.where(rng.random(npx) < rng.random(), 255, 0)
        else:
            px = np.full(npx, rng.integers(1, 256))
        S[i], Q[i] = px.sum(), (px * px).sum()
    return S, Q


def _f32_fma(a, b, c):
    return (np.float64(a) * np.float64(b) + np.float64(c)).astype(np.float32)


@pytest.mark.parametrize("method,w,k", [(SAUVOLA, 31, 0.34), (SAUVOLA, 15, 0.34), (SAUVOLA, 101, 0.01), (SAUVOLA, 31, -0.5),
                                       (NIBLACK, 31, 0.01), (NIBLACK, 15, -0.2), (NICK, 21, -0.01), (NICK, 31, -0.1)])
def test_float32_margin_bounds_the_evaluation_error(prl, method, w, k):
    b = _bounds(prl, method, w, k)
    rng = np.random.default_rng(w * 7 + method)
    S, Q = _window_sums(rng, w, 4000)
    f = 1.0 / float(w * w)
    # exact-arithmetic reference (float64 on exact integers)
    m, q = S * f, Q * f
    v = q - m * m
    keep = v > b["vthr"] * 1.05
    s = np.sqrt(np.where(keep, v, 1.0))
    if method == SAUVOLA:
        T = m * (s * (k / 128.0) + (1.0 - k))
    elif method == NIBLACK:
        T = s * k + m
    else:
        T = m + k * np.sqrt(q)
    # the kernel's float32 sequence (eval32): K~ = fma(w^2, Q~, -S^2~), constants carry f and Z = 2^30
    Z = 2.0 ** 30
    Sf, Qf = S.astype(np.float32), Q.astype(np.float32)
    K32 = _f32_fma(np.float32(w * w), Qf, -(Sf * Sf))
    P2 = np.float32(0.0)  # T only: compare Z*T~ with Z*T*
    one_ulp = np.float32(1 + 2 ** -23)                      # v_sqrt_f32 is 1 ulp, not correctly rounded
    with np.errstate(invalid="ignore"):
        sK = np.sqrt(K32) * one_ulp
        if method == SAUVOLA:
            d = _f32_fma(sK, np.float32(Z * (k / 128.0) * f * f), np.float32(Z * (1.0 - k) * f))
            t2 = _f32_fma(-Sf, d, P2)
        elif method == NIBLACK:
            t2 = _f32_fma(-Sf, np.float32(Z * f), _f32_fma(-sK, np.float32(Z * k * f), P2))
        else:
            t2 = _f32_fma(-Sf, np.float32(Z * f), _f32_fma(-(np.sqrt(Qf) * one_ulp), np.float32(Z * k * f ** 0.5), P2))
    T32 = -t2.astype(np.float64) / Z
    err = np.abs(T32 - T)[keep]
    assert keep.sum() > 3000
    assert err.max() <= b["E1"], (err.max(), b["E1"])
    # relative error of the float32 variance surrogate K~ f^2 (bound rho used by the Wolf sweeps and the K floor)
    rel = np.abs(K32.astype(np.float64) * f * f - v)[keep] / v[keep]
    R = (w - 1) ** 2 / (2.0 * w - 1.0)
    assert rel.max() <= 1.1 * (2 + 2 * R) * 2.0 ** -24


def test_margins_scale_with_page_and_stay_small(prl):
    b4k = _bounds(prl, SAUVOLA, 31, 0.34, 4096)
    b1k = _bounds(prl, SAUVOLA, 31, 0.34, 1024)
    assert b1k["Em"] < b4k["Em"] and b1k["Eq"] < b4k["Eq"]
    assert 1e-6 <= b4k["eps1"] < 5e-3          # the mask packing needs eps1 >= 1e-6; the band must stay narrow
    assert b4k["vthr"] >= 64 * (b4k["Eq"] + 510 * b4k["Em"]) or b4k["vthr"] == 1e-2


@pytest.mark.parametrize("w", [9, 15, 21, 23, 25, 29, 31])
def test_float32_pipeline_q_error_bound(prl, w):
    """flt_usable()'s bound on the float32 pipeline's window sum of squares (cq): the lane chain of
    strip_loop_f is emulated in numpy float32 on adversarial column sums (0 / maximal / random columns) and compared
    with the exact integer window sums: |Q~ - Q| <= delta, and no error at all below Qmin."""
    from prlib_amd import _capi

    L = _hooks()
    L.prl_hip_internal_flt_q_error.argtypes = [C.c_int, C.POINTER(C.c_double)]
    out = (C.c_double * 3)()
    assert L.prl_hip_internal_flt_q_error(w, out) == 1
    delta, qmin, cq = out[0], out[1], out[2]
    n1, lane_off, sh = w - 1, (w - 1) // 8, (w - 1) % 8
    colmax = n1 * 65025
    rng = np.random.default_rng(w)
    worst = 0.0
    for trial in range(300):
        kind = trial % 4
        if kind == 0:
            vq = np.full(512, colmax, np.int64)
        elif kind == 1:
            vq = rng.integers(0, colmax + 1, 512)
        elif kind == 2:   # dark windows beside saturated columns
            vq = np.where((np.arange(512) // int(rng.integers(5, 40))) % 2 == 0, colmax, rng.integers(0, 2000, 512))
        else:
            vq = np.where(rng.random(512) < rng.random(), colmax, rng.integers(0, colmax // 50 + 1, 512))
        v = vq.reshape(64, 8)
        f = v.astype(np.float32)
        # in-lane exclusive prefixes and totals (float32, as the kernel forms them)
        eq = np.zeros((64, 8), np.float32)
        acc = f[:, 0].copy()
        for c in range(1, 8):
            eq[:, c] = acc
            acc = (acc + f[:, c]).astype(np.float32)
        tot = acc
        up = lambda x: np.concatenate([x[1:], np.zeros(1, np.float32)])  # lane + 1 (0 beyond the wavefront)
        w0, w1 = np.zeros(64, np.float32), tot.copy()
        for _ in range(lane_off):
            w0 = w1
            w1 = (tot + up(w1)).astype(np.float32)
        exact_prefix = np.concatenate([[0], np.cumsum(vq)])
        for lane in range(0, 64 - lane_off - 1):
            for c in range(8):
                far1 = (c + sh) >= 8
                fl = lane + lane_off + (1 if far1 else 0)
                far = eq[fl, (c + sh) & 7]
                q32 = np.float32(np.float32(far - eq[lane, c]) + (w1[lane] if far1 else w0[lane]))
                col = lane * 8 + c
                exact = int(exact_prefix[col + n1] - exact_prefix[col])   # columns col .. col + w - 2
                err = abs(float(q32) - exact)
                assert err <= delta, (w, trial, lane, c, err, delta)
                if exact < qmin:
                    assert err == 0.0, (w, trial, lane, c, exact, qmin, err)
                worst = max(worst, err)
    assert cq >= 1.0 and worst <= delta


@pytest.mark.parametrize("w", [33, 41, 51, 63, 81, 101, 121, 129])
def test_float32_sweep_a_wide_window_q_error_bound(prl, w):
    """flt_a_usable()'s ABSOLUTE bound on the float32 window sum of squares of Wolf-Jolion's sweep A with wide windows
    (strip_loop_f, LO == 4): the in-lane prefixes and the doubling form of the lane sums W (sums of 2, 4, 8, 16 consecutive lanes,
    pieces picked by the binary digits of the lane offset) are emulated in numpy float32 on adversarial column sums and compared
    with the exact integer window sums.  The S sums of the same shape must be exact."""
    from prlib_amd import _capi

    L = _hooks()
    L.prl_hip_internal_flt_a_q_error.argtypes = [C.c_int, C.POINTER(C.c_double)]
    dq = C.c_double(0)
    assert L.prl_hip_internal_flt_a_q_error(w, C.byref(dq)) == 1
    n1, loff, sh = w - 1, (w - 1) // 8, (w - 1) % 8
    assert 4 <= loff <= 16
    rng = np.random.default_rng(w)
    shl = lambda x, k: np.concatenate([x[k:], np.zeros(k, np.float32)])   # value of lane + k (garbage lanes are never valid outputs)
    f32 = lambda x: x.astype(np.float32)
    worst = 0.0
    for colmax, exact_required in ((n1 * 65025, False), (n1 * 255, True)):   # column sums of P*P, and of P (must come out exact)
        for trial in range(120):
            kind = trial % 4
            if kind == 0:
                vq = np.full(512, colmax, np.int64)
            elif kind == 1:
                vq = rng.integers(0, colmax + 1, 512)
            elif kind == 2:
                vq = np.where((np.arange(512) // int(rng.integers(5, 60))) % 2 == 0, colmax, rng.integers(0, 2000, 512))
            else:
                vq = np.where(rng.random(512) < rng.random(), colmax, rng.integers(0, colmax // 50 + 1, 512))
            f = f32(vq.reshape(64, 8))
            eq = np.zeros((64, 8), np.float32)
            acc = f[:, 0].copy()
            for c in range(1, 8):
                eq[:, c] = acc
                acc = f32(acc + f[:, c])
            tot = acc
            s2 = f32(tot + shl(tot, 1))
            s4 = f32(s2 + shl(s2, 2))
            s8 = f32(s4 + shl(s4, 4))
            w0, off = np.zeros(64, np.float32), 0
            if loff & 16:
                w0, off = f32(s8 + shl(s8, 8)), 16
            for bit, piece in ((8, s8), (4, s4), (2, s2), (1, tot)):
                if loff & bit:
                    w0 = f32(w0 + (shl(piece, off) if off else piece))
                    off += bit
            w1 = f32(w0 + shl(tot, off))
            exact_prefix = np.concatenate([[0], np.cumsum(vq)])
            for lane in range(0, 64 - loff - 1):
                for c in range(8):
                    far1 = (c + sh) >= 8
                    fl = lane + loff + (1 if far1 else 0)
                    q32 = np.float32(np.float32(eq[fl, (c + sh) & 7] - eq[lane, c]) + (w1[lane] if far1 else w0[lane]))
                    col = lane * 8 + c
                    exact = int(exact_prefix[col + n1] - exact_prefix[col])
                    err = abs(float(q32) - exact)
                    if exact_required:
                        assert err == 0.0, (w, trial, lane, c, err)
                    else:
                        assert err <= dq.value, (w, trial, lane, c, err, dq.value)
                        worst = max(worst, err)
    assert worst <= dq.value


def test_strip_layout(prl):
    """Strips per row (binarize_fused.hip strip_layout).  The extended last strip saves the seventh strip of an A4 row and
    the eleventh of a 4096-column row at the default w = 101; a ragged uo saves the sixth strip of an A4 row at NICK's
    default w = 21; rows without a small remainder, bit-plane output and Wolf-Jolion keep uo a multiple of 8; an extended
    strip never has more than 64 lanes of outputs and its last lane is padding."""
    from prlib_amd import _capi

    SAUVOLA, NIBLACK, WOLFJOLION, NICK, FENG = range(5)
_HOOKS = None


def _hooks():
    """libprlib_hip_testhooks.so: the product library exports only what include/prl_hip.h declares; the prl_hip_internal_* entries
    (host-side helpers of the fused pipeline, no device needed) live in the test-hooks build."""
    global _HOOKS
    if _HOOKS is None:
        from prlib_amd import _capi

        _HOOKS = C.CDLL(_capi.HOOKS_LIB_PATH)
    return _HOOKS
    L = _hooks()
    L.prl_hip_internal_strip_layout.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_int)]
    out = (C.c_int * 2)()

    def layout(method, w, width, bit_out=0):
        ow = width - 1 if method in (SAUVOLA, NIBLACK) else width - w
        n = L.prl_hip_internal_strip_layout(method, w, width, ow, bit_out, out)
        return n, out[0], out[1]

    assert layout(NIBLACK, 101, 2480) == (6, 408, 1)     # A4 at 300 dpi, the header defaults
    assert layout(SAUVOLA, 101, 4096) == (10, 408, 1)
    assert layout(NICK, 101, 2480) == (6, 408, 0)        # 2379 outputs: six plain strips
    assert layout(SAUVOLA, 31, 4096) == (9, 480, 0)      # the headline: 8.5 strips of work either way
    assert layout(SAUVOLA, 31, 2480) == (6, 480, 0)
    assert layout(NICK, 21, 2480) == (5, 492, 0)         # 2459 outputs = 5 x 492 - 1
    assert layout(NICK, 21, 2480, bit_out=1) == (6, 488, 0)
    assert layout(WOLFJOLION, 21, 2480) == (6, 488, 0)
    assert layout(SAUVOLA, 41, 486) == (1, 472, 1) and layout(SAUVOLA, 41, 487) == (2, 472, 0)
    assert layout(SAUVOLA, 101, 456) == (1, 408, 1) and layout(SAUVOLA, 101, 457) == (2, 408, 0)
    for w in (9, 15, 21, 31, 41, 51, 101, 151, 201, 257):
        uo8, h = ((512 - (w - 1)) // 8) * 8, w // 2
        for width in range(w + 2, 2200, 3):
            ow = width - 1
            n, uo, e = layout(SAUVOLA, w, width)
            assert uo in (uo8, 512 - (w - 1))
            plain = -(-ow // uo)
            assert n == plain - e and n <= -(-ow // uo8)
            if uo != uo8:
                assert w <= 31 and not e and n < -(-ow // uo8)
            if e:
                xs = (n - 1) * uo
                assert ow - xs <= 512 and xs + 1 - h + 504 >= width - 1
            else:
                assert ow - (n - 1) * uo <= uo
