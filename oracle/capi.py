"""ctypes binding of oracle/libprl_oracle.so (the CPU restatement).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Nothing under prlib_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libprl_oracle.so")

SAUVOLA, NIBLACK, WOLFJOLION, NICK, FENG = range(5)
METHOD_NAMES = {SAUVOLA: "sauvola", NIBLACK: "niblack", WOLFJOLION: "wolfjolion", NICK: "nick", FENG: "feng"}

PRL_OK, PRL_ERR_EMPTY, PRL_ERR_BAD_WINDOW, PRL_ERR_BAD_CHANNELS, PRL_ERR_EMPTY_RECT, PRL_ERR_BAD_ARG = range(6)


class Params(C.Structure):
    """struct prl_binarize_params (include/prl_hip.h)."""

    _fields_ = [
        ("method", C.c_int32),
        ("window_size", C.c_int32),
        ("k", C.c_double),
        ("morph_iterations", C.c_int32),
        ("reserved0", C.c_int32),
        ("feng_alpha1", C.c_double),
        ("feng_k1", C.c_double),
        ("feng_k2", C.c_double),
        ("feng_gamma", C.c_double),
    ]


class Geometry(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("w", "half", "padded_w", "padded_h", "out_w", "out_h")]


# the reference's default arguments (binarizeSauvola.h:43-47, binarizeNICK.h:43-47, binarizeFeng.h:46-53)
_DEFAULTS = {
    SAUVOLA: dict(window_size=101, k=0.01, morph_iterations=2),
    NIBLACK: dict(window_size=101, k=0.01, morph_iterations=2),
    WOLFJOLION: dict(window_size=101, k=0.01, morph_iterations=2),
    NICK: dict(window_size=21, k=-0.01, morph_iterations=0),
    FENG: dict(window_size=21, k=0.0, morph_iterations=2),
}


def make_params(method: int, window_size=None, k=None, morph_iterations=None,
                alpha1=0.75, k1=0.2, k2=0.03, gamma=2.0) -> Params:
    d = dict(_DEFAULTS[method])
    if window_size is not None:
        d["window_size"] = window_size
    if k is not None:
        d["k"] = k
    if morph_iterations is not None:
        d["morph_iterations"] = morph_iterations
    return Params(method, d["window_size"], d["k"], d["morph_iterations"], 0, alpha1, k1, k2, gamma)


def build(force: bool = False) -> str:
    """Compile the oracle with the committed Makefile (gcc only)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        u8p = C.c_void_p
        L.prl_oracle_binarize_geometry.argtypes = [C.POINTER(Params), C.c_int, C.c_int, C.POINTER(Geometry)]
        L.prl_oracle_binarize.argtypes = [C.POINTER(Params), u8p, C.c_size_t, C.c_int, C.c_int, u8p, C.c_size_t]
        L.prl_oracle_binarize_batch.argtypes = [C.POINTER(Params), C.c_int, u8p, C.c_size_t, C.c_size_t,
                                                C.c_int, C.c_int, u8p, C.c_size_t, C.c_size_t, C.c_int]
        L.prl_oracle_threshold_plane.argtypes = [C.POINTER(Params), u8p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
        L.prl_oracle_mean_dev.argtypes = [C.POINTER(Params), u8p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.prl_oracle_morph.argtypes = [C.c_int, u8p, C.c_size_t, C.c_int, C.c_int, u8p, C.c_size_t]
        L.prl_oracle_morph.restype = None
        L.prl_oracle_pad_replicate.argtypes = [u8p, C.c_size_t, C.c_int, C.c_int, C.c_int, u8p, C.c_size_t]
        L.prl_oracle_pad_replicate.restype = None
        L.prl_oracle_sat_u8.argtypes = [C.c_double]
        L.prl_oracle_sat_u8.restype = C.c_uint8
        L.prl_oracle_bgr2gray.argtypes = [u8p, C.c_size_t, C.c_int, C.c_int, C.c_int, u8p, C.c_size_t]
        L.prl_oracle_bgr2gray.restype = None
        L.prl_oracle_otsu.argtypes = [u8p, C.c_size_t, C.c_int, C.c_int, u8p, C.c_size_t]
        if hasattr(L, "prl_oracle_nlm_planes"):
            L.prl_oracle_nlm_weights.argtypes = [C.c_int, C.c_float, C.c_void_p, C.c_int]
            L.prl_oracle_nlm_planes.argtypes = [C.c_int, C.c_float, u8p, C.c_size_t, C.c_int, C.c_int,
                                                u8p, C.c_size_t, C.c_int]
            L.prl_oracle_lbgr2lab.argtypes = [u8p, C.c_size_t, C.c_int, C.c_int, C.c_int, u8p, C.c_size_t]
            L.prl_oracle_lbgr2lab.restype = None
            L.prl_oracle_lab2lbgr.argtypes = [u8p, C.c_size_t, C.c_int, C.c_int, u8p, C.c_size_t, C.c_int]
            L.prl_oracle_lab2lbgr.restype = None
            L.prl_oracle_denoise.argtypes = [C.c_int, C.c_float, u8p, C.c_size_t, C.c_int, C.c_int,
                                             u8p, C.c_size_t, C.c_int]
        L.prl_oracle_thin.argtypes = [C.c_int, u8p, C.c_size_t, C.c_int, C.c_int, u8p, C.c_size_t, C.POINTER(C.c_int)]
        L.prl_oracle_bgnorm_map_size.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.prl_oracle_bgnorm_map_size.restype = None
        L.prl_oracle_bgnorm_fgmask.argtypes = [C.c_int, u8p, C.c_size_t, C.c_int, C.c_int, u8p]
        L.prl_oracle_bgnorm_fgmask.restype = None
        L.prl_oracle_bgnorm_bgmap.argtypes = [C.c_int, C.c_int, u8p, C.c_size_t, C.c_int, C.c_int, u8p, u8p]
        L.prl_oracle_bgnorm_blockconv.argtypes = [u8p, C.c_int, C.c_int, u8p]
        L.prl_oracle_bgnorm_blockconv.restype = None
        L.prl_oracle_bgnorm_invmap.argtypes = [u8p, C.c_int, C.c_int, C.c_void_p]
        L.prl_oracle_bgnorm.argtypes = [C.c_int, u8p, C.c_size_t, C.c_int, C.c_int, u8p, C.c_size_t]
        L.prl_oracle_houghp.argtypes = [u8p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.prl_oracle_vote_angle.argtypes = [C.c_void_p, C.c_int]
        L.prl_oracle_vote_angle.restype = C.c_double
        L.prl_oracle_find_angle.argtypes = [u8p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.prl_oracle_find_angle.restype = C.c_double
        L.prl_oracle_rotate_kind.argtypes = [C.c_double]
        L.prl_oracle_rotate_size.argtypes = [C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.prl_oracle_rotate_size.restype = None
        L.prl_oracle_rotate_matrix.argtypes = [C.c_int, C.c_int, C.c_double, C.POINTER(C.c_double * 6)]
        L.prl_oracle_rotate_matrix.restype = None
        L.prl_oracle_rotate.argtypes = [C.c_int, u8p, C.c_size_t, C.c_int, C.c_int, C.c_double, u8p, C.c_size_t]
        L.prl_oracle_deskew.argtypes = [C.c_int, u8p, C.c_size_t, C.c_int, C.c_int, u8p, C.c_size_t, C.POINTER(C.c_int),
                                        C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.prl_oracle_local_variance_map.argtypes = [u8p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
        L.prl_oracle_local_variance_map.restype = None
        L.prl_oracle_binarize_lv.argtypes = [u8p, C.c_size_t, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, u8p, C.c_size_t]
        L.prl_oracle_binarize_lv_nofilters.argtypes = [u8p, C.c_size_t, C.c_int, C.c_int, C.c_double, C.c_int, u8p, C.c_size_t]
        _lib = L
    return _lib


class OracleError(RuntimeError):
    def __init__(self, status: int):
        super().__init__(f"oracle status {status}")
        self.status = status


def _ptr(a: np.ndarray) -> int:
    return a.ctypes.data


def geometry(params: Params, width: int, height: int):
    g = Geometry()
    st = lib().prl_oracle_binarize_geometry(C.byref(params), width, height, C.byref(g))
    return st, g


def binarize(img: np.ndarray, params: Params) -> np.ndarray:
    """One page (H x W uint8, any row stride) -> mask."""
    assert img.dtype == np.uint8 and img.ndim == 2 and img.strides[1] == 1
    h, w = img.shape
    st, g = geometry(params, w, h)
    if st != PRL_OK:
        raise OracleError(st)
    out = np.empty((g.out_h, g.out_w), dtype=np.uint8)
    st = lib().prl_oracle_binarize(C.byref(params), _ptr(img), img.strides[0], w, h, _ptr(out), out.strides[0])
    if st != PRL_OK:
        raise OracleError(st)
    return out


def binarize_batch(pages: np.ndarray, params: Params, threads: int = 1) -> np.ndarray:
    """pages: N x H x W uint8 contiguous."""
    assert pages.dtype == np.uint8 and pages.ndim == 3 and pages.flags.c_contiguous
    n, h, w = pages.shape
    st, g = geometry(params, w, h)
    if st != PRL_OK:
        raise OracleError(st)
    out = np.empty((n, g.out_h, g.out_w), dtype=np.uint8)
    st = lib().prl_oracle_binarize_batch(C.byref(params), n, _ptr(pages), pages.strides[0], pages.strides[1],
                                         w, h, _ptr(out), out.strides[0], out.strides[1], threads)
    if st != PRL_OK:
        raise OracleError(st)
    return out


def threshold_plane(img: np.ndarray, params: Params) -> np.ndarray:
    h, w = img.shape
    st, g = geometry(params, w, h)
    if st != PRL_OK:
        raise OracleError(st)
    t = np.empty((g.out_h, g.out_w), dtype=np.float64)
    st = lib().prl_oracle_threshold_plane(C.byref(params), _ptr(img), img.strides[0], w, h, _ptr(t))
    if st != PRL_OK:
        raise OracleError(st)
    return t


def mean_dev(img: np.ndarray, params: Params):
    h, w = img.shape
    st, g = geometry(params, w, h)
    if st != PRL_OK:
        raise OracleError(st)
    m = np.empty((g.out_h, g.out_w), dtype=np.float64)
    s = np.empty((g.out_h, g.out_w), dtype=np.float64)
    st = lib().prl_oracle_mean_dev(C.byref(params), _ptr(img), img.strides[0], w, h, _ptr(m), _ptr(s))
    if st != PRL_OK:
        raise OracleError(st)
    return m, s


def morph(mask: np.ndarray, iterations: int) -> np.ndarray:
    assert mask.dtype == np.uint8 and mask.ndim == 2 and mask.strides[1] == 1
    h, w = mask.shape
    out = np.empty((h, w), dtype=np.uint8)
    lib().prl_oracle_morph(iterations, _ptr(mask), mask.strides[0], w, h, _ptr(out), out.strides[0])
    return out


def pad_replicate(img: np.ndarray, half: int) -> np.ndarray:
    h, w = img.shape
    out = np.empty((h + 2 * half, w + 2 * half), dtype=np.uint8)
    lib().prl_oracle_pad_replicate(_ptr(img), img.strides[0], w, h, half, _ptr(out), out.strides[0])
    return out


def sat_u8(v: float) -> int:
    return int(lib().prl_oracle_sat_u8(float(v)))


def bgr2gray(img: np.ndarray) -> np.ndarray:
    h, w, c = img.shape
    out = np.empty((h, w), dtype=np.uint8)
    lib().prl_oracle_bgr2gray(_ptr(img), img.strides[0], w, h, c, _ptr(out), out.strides[0])
    return out


def otsu(img: np.ndarray):
    h, w = img.shape
    out = np.empty((h, w), dtype=np.uint8)
    thr = lib().prl_oracle_otsu(_ptr(img), img.strides[0], w, h, _ptr(out), out.strides[0])
    return thr, out


# ---- NL-means ---------------------------------------------------------------------------------

def nlm_weights(channels: int, h: float, cap: int = 1 << 18) -> np.ndarray:
    lut = np.zeros(cap, dtype=np.int32)
    n = lib().prl_oracle_nlm_weights(channels, h, _ptr(lut), cap)
    return lut[:n].copy()


def nlm_planes(img: np.ndarray, h: float, threads: int = 1) -> np.ndarray:
    """img: H x W (1 plane) or H x W x C (C interleaved planes), uint8."""
    a = img if img.ndim == 3 else img[:, :, None]
    a = np.ascontiguousarray(a)
    hh, ww, c = a.shape
    out = np.empty_like(a)
    st = lib().prl_oracle_nlm_planes(c, h, _ptr(a), a.strides[0], ww, hh, _ptr(out), out.strides[0], threads)
    if st != PRL_OK:
        raise OracleError(st)
    return out if img.ndim == 3 else out[:, :, 0]


def lbgr2lab(img: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(img)
    h, w, c = a.shape
    out = np.empty((h, w, 3), dtype=np.uint8)
    lib().prl_oracle_lbgr2lab(_ptr(a), a.strides[0], w, h, c, _ptr(out), out.strides[0])
    return out


def lab2lbgr(lab: np.ndarray, channels: int = 3) -> np.ndarray:
    a = np.ascontiguousarray(lab)
    h, w, _ = a.shape
    out = np.empty((h, w, channels), dtype=np.uint8)
    lib().prl_oracle_lab2lbgr(_ptr(a), a.strides[0], w, h, _ptr(out), out.strides[0], channels)
    return out


def denoise(img: np.ndarray, strength: float, threads: int = 1) -> np.ndarray:
    a = np.ascontiguousarray(img)
    h, w, c = a.shape
    out = np.empty_like(a)
    st = lib().prl_oracle_denoise(c, strength, _ptr(a), a.strides[0], w, h, _ptr(out), out.strides[0], threads)
    if st != PRL_OK:
        raise OracleError(st)
    return out


# ---- thinning -----------------------------------------------------------------------------------

ZHANGSUEN, GUOHALL = 0, 1


def thin(img: np.ndarray, method: int = ZHANGSUEN, return_passes: bool = False):
    """prl::thinZhangSuen / prl::thinGuoHall on a 1-channel u8 image (after cvtColor)."""
    assert img.dtype == np.uint8 and img.ndim == 2 and img.strides[1] == 1
    h, w = img.shape
    out = np.empty((h, w), dtype=np.uint8)
    passes = C.c_int(0)
    st = lib().prl_oracle_thin(method, _ptr(img), img.strides[0], w, h, _ptr(out), out.strides[0], C.byref(passes))
    if st != PRL_OK:
        raise OracleError(st)
    return (out, passes.value) if return_passes else out


# ---- background normalisation (prl_oracle_bgnorm.c) ---------------------------------------------

def _as3(img: np.ndarray) -> np.ndarray:
    a = img if img.ndim == 3 else img[:, :, None]
    assert a.dtype == np.uint8 and (a.shape[2] == 1 or a.strides[2] == 1) and a.strides[1] == a.shape[2]
    return a


def bgnorm_map_size(width: int, height: int):
    mw, mh = C.c_int(0), C.c_int(0)
    lib().prl_oracle_bgnorm_map_size(width, height, C.byref(mw), C.byref(mh))
    return mw.value, mh.value


def bgnorm_fgmask(img: np.ndarray) -> np.ndarray:
    a = _as3(img)
    h, w, c = a.shape
    fg = np.empty((h, w), dtype=np.uint8)
    lib().prl_oracle_bgnorm_fgmask(c, _ptr(a), a.strides[0], w, h, _ptr(fg))
    return fg


def bgnorm_bgmap(img: np.ndarray, channel: int = 0):
    """(failed, map) for one channel: tile averages + pixFillMapHoles."""
    a = _as3(img)
    h, w, c = a.shape
    fg = bgnorm_fgmask(img)
    mw, mh = bgnorm_map_size(w, h)
    m = np.empty((mh, mw), dtype=np.uint8)
    failed = lib().prl_oracle_bgnorm_bgmap(c, channel, _ptr(a), a.strides[0], w, h, _ptr(fg), _ptr(m))
    return bool(failed), m


def bgnorm_blockconv(m: np.ndarray) -> np.ndarray:
    m = np.ascontiguousarray(m)
    out = np.empty_like(m)
    lib().prl_oracle_bgnorm_blockconv(_ptr(m), m.shape[1], m.shape[0], _ptr(out))
    return out


def bgnorm_invmap(m: np.ndarray):
    m = np.ascontiguousarray(m)
    inv = np.empty(m.shape, dtype=np.uint16)
    failed = lib().prl_oracle_bgnorm_invmap(_ptr(m), m.shape[1], m.shape[0], _ptr(inv))
    return bool(failed), inv


def bgnorm(img: np.ndarray) -> np.ndarray:
    """prl::backgroundNormalization: H x W (1 channel) or H x W x {3,4} uint8; 4 channels come back as 3."""
    a = _as3(img)
    h, w, c = a.shape
    och = 1 if c == 1 else 3
    out = np.empty((h, w, och), dtype=np.uint8)
    st = lib().prl_oracle_bgnorm(c, _ptr(a), a.strides[0], w, h, _ptr(out), out.strides[0])
    if st != PRL_OK:
        raise OracleError(st)
    return out if img.ndim == 3 else out[:, :, 0]


# ---- deskew / rotate (prl_oracle_deskew.c) --------------------------------------------------------

def houghp(img: np.ndarray, threshold: int, line_length: int, line_gap: int) -> np.ndarray:
    """cv::HoughLinesP(img, lines, 1, CV_PI/180, threshold, line_length, line_gap) -> n x 4 int32."""
    assert img.dtype == np.uint8 and img.ndim == 2 and img.strides[1] == 1
    h, w = img.shape
    cap = 4096
    while True:
        lines = np.empty((cap, 4), dtype=np.int32)
        n = lib().prl_oracle_houghp(_ptr(img), img.strides[0], w, h, threshold, line_length, line_gap, _ptr(lines), cap)
        if n <= cap:
            return lines[:n].copy()
        cap = n


def vote_angle(lines: np.ndarray) -> float:
    lines = np.ascontiguousarray(lines, dtype=np.int32)
    return float(lib().prl_oracle_vote_angle(_ptr(lines), len(lines)))


def find_angle(binary: np.ndarray):
    assert binary.dtype == np.uint8 and binary.ndim == 2 and binary.strides[1] == 1
    h, w = binary.shape
    n = C.c_int(0)
    ang = lib().prl_oracle_find_angle(_ptr(binary), binary.strides[0], w, h, C.byref(n))
    return float(ang), n.value


def rotate_size(width: int, height: int, angle: float):
    ow, oh = C.c_int(0), C.c_int(0)
    lib().prl_oracle_rotate_size(width, height, angle, C.byref(ow), C.byref(oh))
    return ow.value, oh.value


def rotate_matrix(width: int, height: int, angle: float) -> np.ndarray:
    m = (C.c_double * 6)()
    lib().prl_oracle_rotate_matrix(width, height, angle, C.byref(m))
    return np.array(list(m), dtype=np.float64)


def rotate(img: np.ndarray, angle: float) -> np.ndarray:
    a = _as3(img)
    h, w, c = a.shape
    ow, oh = rotate_size(w, h, angle)
    out = np.empty((oh, ow, c), dtype=np.uint8)
    st = lib().prl_oracle_rotate(c, _ptr(a), a.strides[0], w, h, angle, _ptr(out), out.strides[0])
    if st != PRL_OK:
        raise OracleError(st)
    return out if img.ndim == 3 else out[:, :, 0]


def deskew(img: np.ndarray):
    """prl::deskew -> (image, info) with info = dict(angle, otsu, n_lines)."""
    a = _as3(img)
    h, w, c = a.shape
    ln = max(w, h)
    buf = np.empty((ln, ln, c), dtype=np.uint8)
    ow, oh, ang, thr, nl = C.c_int(0), C.c_int(0), C.c_double(0), C.c_int(0), C.c_int(0)
    st = lib().prl_oracle_deskew(c, _ptr(a), a.strides[0], w, h, _ptr(buf), buf.strides[0], C.byref(ow), C.byref(oh),
                                 C.byref(ang), C.byref(thr), C.byref(nl))
    if st != PRL_OK:
        raise OracleError(st)
    out = buf[: oh.value, : ow.value].copy()
    info = dict(angle=ang.value, otsu=thr.value, n_lines=nl.value)
    return (out if img.ndim == 3 else out[:, :, 0]), info


# ---- binarizeByLocalVariances (prl_oracle_lv.c) ----------------------------------------------------

def local_variance_map(img: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(img)
    h, w, c = a.shape
    assert c == 3 and a.dtype == np.uint8
    out = np.empty((h, w, 3), dtype=np.float32)
    lib().prl_oracle_local_variance_map(_ptr(a), a.strides[0], w, h, _ptr(out))
    return out


def binarize_lv(img: np.ndarray, coeff: float = 0.125, min_result_variance: int = 25, gamma: float = 2.0) -> np.ndarray:
    a = np.ascontiguousarray(img)
    h, w, c = a.shape
    if c != 3:
        raise OracleError(PRL_ERR_BAD_CHANNELS)
    out = np.empty((h, w), dtype=np.uint8)
    st = lib().prl_oracle_binarize_lv(_ptr(a), a.strides[0], w, h, coeff, min_result_variance, gamma, _ptr(out), out.strides[0])
    if st != PRL_OK:
        raise OracleError(st)
    return out


def binarize_lv_nofilters(img: np.ndarray, coeff: float = 0.125, min_result_variance: int = 10) -> np.ndarray:
    a = np.ascontiguousarray(img)
    h, w, c = a.shape
    if c != 3:
        raise OracleError(PRL_ERR_BAD_CHANNELS)
    out = np.empty((h, w), dtype=np.uint8)
    st = lib().prl_oracle_binarize_lv_nofilters(_ptr(a), a.strides[0], w, h, coeff, min_result_variance, _ptr(out), out.strides[0])
    if st != PRL_OK:
        raise OracleError(st)
    return out
