"""Channel adapters and the device-resident chain over the C ABI.

`cvtColorBGR2GRAY` is the `cv::cvtColor(..., COLOR_BGR2GRAY)` every reference binarizer runs first on a colour input
(src/binarizations/binarizeSauvola.cpp:51); `cvtColorGRAY2BGR` is what `prl::denoise` needs in front of it for a gray
scan; `bitwise_not` turns a binarizer's mask (white = paper) into what `prl::thinZhangSuen` thins (white = strokes,
src/thinning/thinZhangSuen.cpp:85).  `process_pages` strings denoise -> gray -> binarize -> invert -> thinning together
without leaving the device (BASELINE config 5 minus deskew / background normalisation, which need Leptonica).
"""
from __future__ import annotations

import ctypes as C

from . import _capi
from .binarizations import make_params

NO_THINNING, ZHANGSUEN, GUOHALL = -1, 0, 1


def _pages(t, channels_last: bool):
    import torch

    if t.dtype != torch.uint8 or not t.is_cuda:
        raise TypeError("expected a uint8 CUDA tensor")
    want = 3 if channels_last else 2
    squeeze = t.dim() == want
    if squeeze:
        t = t.unsqueeze(0)
    if t.dim() != want + 1:
        raise TypeError("expected [N,] H x W" + (" x C" if channels_last else ""))
    if channels_last:
        if t.stride(3) != 1 or t.stride(2) != t.shape[3]:
            raise TypeError("pixels must be interleaved and rows dense")
    elif t.stride(2) != 1:
        raise TypeError("rows must be dense")
    if t.numel() == 0:
        raise ValueError("Input image is empty")
    return t, squeeze


def _stream(t):
    import torch

    _capi.check(_capi.lib().prl_hip_set_device(t.device.index or 0))
    return torch.cuda.current_stream(t.device).cuda_stream


def cvtColorBGR2GRAY(image, out=None):
    """[N,] H x W x 3|4 (BGR / BGRA) -> [N,] H x W."""
    import torch

    t, squeeze = _pages(image, True)
    n, h, w, c = t.shape
    if c not in (3, 4):
        raise ValueError("expected 3 or 4 channels")
    o = torch.empty((n, h, w), dtype=torch.uint8, device=t.device) if out is None else (out.unsqueeze(0) if out.dim() == 2 else out)
    _capi.check(_capi.lib().prl_hip_bgr2gray_batch_device(n, c, t.data_ptr(), t.stride(0), t.stride(1), w, h,
                                                         o.data_ptr(), o.stride(0), o.stride(1), _stream(t)))
    return o[0] if squeeze else o


def cvtColorGRAY2BGR(image, channels: int = 3, out=None):
    """[N,] H x W -> [N,] H x W x channels (alpha = 255 for 4)."""
    import torch

    t, squeeze = _pages(image, False)
    n, h, w = t.shape
    if channels not in (3, 4):
        raise ValueError("expected 3 or 4 channels")
    o = torch.empty((n, h, w, channels), dtype=torch.uint8, device=t.device) if out is None else (out.unsqueeze(0) if out.dim() == 3 else out)
    _capi.check(_capi.lib().prl_hip_gray2bgr_batch_device(n, channels, t.data_ptr(), t.stride(0), t.stride(1), w, h,
                                                         o.data_ptr(), o.stride(0), o.stride(1), _stream(t)))
    return o[0] if squeeze else o


def bitwise_not(image, out=None):
    """[N,] H x W -> 255 - image (in place when out is image)."""
    import torch

    t, squeeze = _pages(image, False)
    n, h, w = t.shape
    o = torch.empty_like(t) if out is None else (out.unsqueeze(0) if out.dim() == 2 else out)
    _capi.check(_capi.lib().prl_hip_invert_batch_device(n, t.data_ptr(), t.stride(0), t.stride(1), w, h,
                                                       o.data_ptr(), o.stride(0), o.stride(1), _stream(t)))
    return o[0] if squeeze else o


def process_pages(pages, channels: int, method: int = 0, windowSize: int = 101, k: float = 0.01,
                  morphIterationCount: int = 2, denoise_strength=None, thin: int = NO_THINNING, out=None,
                  deskew: bool = False, background_normalization: bool = False, **feng):
    """[deskew] -> [denoise] (if `denoise_strength` is given; colour pages only) -> [backgroundNormalization] -> gray ->
    binarize -> thinning of the inverted mask: BASELINE config 5's chain with device-resident intermediates.

    pages: uint8 CUDA tensor [N,] H x W (channels = 1) or [N,] H x W x channels (BGR / BGRA).
    Returns [N,] out_h x out_w: the mask, or with `thin` the skeleton of the dark strokes.  With `deskew` every page has
    its own result size: returns (list of per-page tensors, angles in degrees) - (tensor, angle) for a single un-batched
    page - and `out` is not accepted."""
    import numpy as np
    import torch

    if deskew:
        if out is not None:
            raise ValueError("`out` cannot be given with deskew: every page's result has its own size")
        t, squeeze = _pages(pages, channels != 1)
        n, h, w = t.shape[0], t.shape[1], t.shape[2]
        if channels != 1 and t.shape[3] != channels:
            raise ValueError("last dimension does not match `channels`")
        L = _capi.lib()
        cp = _capi.ChainParams()
        L.prl_hip_default_chain_params(C.byref(cp))
        cp.denoise = 0 if denoise_strength is None else 1
        cp.denoise_strength = 5.5 if denoise_strength is None else float(denoise_strength)
        cp.binarize = make_params(method, windowSize, k, morphIterationCount, **feng)
        cp.thin = int(thin)
        cp.deskew = 1
        cp.background_normalization = 1 if background_normalization else 0
        mw, mh = C.c_int(0), C.c_int(0)
        st = L.prl_hip_chain_max_out_size(C.byref(cp), w, h, C.byref(mw), C.byref(mh))
        if st in (_capi.PRL_ERR_EMPTY, _capi.PRL_ERR_BAD_WINDOW):
            raise ValueError(L.prl_hip_strerror(st).decode())
        _capi.check(st)
        o = torch.empty((n, mh.value, mw.value), dtype=torch.uint8, device=t.device)
        wh = np.zeros((n, 2), dtype=np.int32)
        ang = np.zeros(n, dtype=np.float64)
        _capi.check(L.prl_hip_chain_pages_device(C.byref(cp), n, channels, t.data_ptr(), t.stride(0), t.stride(1), w, h,
                                                o.data_ptr(), o.stride(0), o.stride(1), wh.ctypes.data, ang.ctypes.data, _stream(t)))
        res = [o[i, : wh[i, 1], : wh[i, 0]] for i in range(n)]
        return (res[0], ang[0]) if squeeze else (res, ang)

    t, squeeze = _pages(pages, channels != 1)
    n, h, w = t.shape[0], t.shape[1], t.shape[2]
    if channels != 1 and t.shape[3] != channels:
        raise ValueError("last dimension does not match `channels`")
    L = _capi.lib()
    cp = _capi.ChainParams()
    L.prl_hip_default_chain_params(C.byref(cp))
    cp.denoise = 0 if denoise_strength is None else 1
    cp.denoise_strength = 5.5 if denoise_strength is None else float(denoise_strength)
    cp.binarize = make_params(method, windowSize, k, morphIterationCount, **feng)
    cp.thin = int(thin)
    cp.background_normalization = 1 if background_normalization else 0
    g = _capi.BinarizeGeometry()
    st = L.prl_hip_binarize_geometry(C.byref(cp.binarize), w, h, C.byref(g))
    if st in (_capi.PRL_ERR_EMPTY, _capi.PRL_ERR_BAD_WINDOW):
        raise ValueError(L.prl_hip_strerror(st).decode())   # std::invalid_argument in the reference
    _capi.check(st)
    o = torch.empty((n, g.out_h, g.out_w), dtype=torch.uint8, device=t.device) if out is None else (out.unsqueeze(0) if out.dim() == 2 else out)
    if tuple(o.shape) != (n, g.out_h, g.out_w) or o.stride(2) != 1:
        raise ValueError("output tensor has the wrong shape")
    _capi.check(L.prl_hip_chain_batch_device(C.byref(cp), n, channels, t.data_ptr(), t.stride(0), t.stride(1), w, h,
                                            o.data_ptr(), o.stride(0), o.stride(1), _stream(t)))
    return o[0] if squeeze else o


def process_pages_host(pages, method: int = 0, windowSize: int = 101, k: float = 0.01, morphIterationCount: int = 2,
                       denoise_strength=None, thin: int = NO_THINNING, deskew: bool = False,
                       background_normalization: bool = False, n_devices: int = 0, **feng):
    """prl_hip_chain_batch_host: the chain of process_pages on a list (or array) of equal-size uint8 HOST pages, H x W or
    H x W x 3|4, sharded over the node's GPUs by the library.  Returns (list of per-page numpy results, angles)."""
    import numpy as np

    pages = [np.ascontiguousarray(p) for p in pages]
    n = len(pages)
    if n == 0:
        return [], np.zeros(0)
    shape = pages[0].shape
    if any(p.shape != shape or p.dtype != np.uint8 for p in pages) or len(shape) not in (2, 3):
        raise TypeError("expected equal-size uint8 pages")
    h, w = shape[0], shape[1]
    channels = 1 if len(shape) == 2 else shape[2]
    L = _capi.lib()
    cp = _capi.ChainParams()
    L.prl_hip_default_chain_params(C.byref(cp))
    cp.denoise = 0 if denoise_strength is None else 1
    cp.denoise_strength = 5.5 if denoise_strength is None else float(denoise_strength)
    cp.binarize = make_params(method, windowSize, k, morphIterationCount, **feng)
    cp.thin = int(thin)
    cp.deskew = 1 if deskew else 0
    cp.background_normalization = 1 if background_normalization else 0
    mw, mh = C.c_int(0), C.c_int(0)
    st = L.prl_hip_chain_max_out_size(C.byref(cp), w, h, C.byref(mw), C.byref(mh))
    if st in (_capi.PRL_ERR_EMPTY, _capi.PRL_ERR_BAD_WINDOW):
        raise ValueError(L.prl_hip_strerror(st).decode())
    _capi.check(st)
    out = np.empty((n, mh.value, mw.value), dtype=np.uint8)
    wh = np.zeros((n, 2), dtype=np.int32)
    angles = np.zeros(n, dtype=np.float64)
    src = (C.c_void_p * n)(*[p.ctypes.data for p in pages])
    dst = (C.c_void_p * n)(*[out[i].ctypes.data for i in range(n)])
    _capi.check(L.prl_hip_chain_batch_host(C.byref(cp), n, channels, src, pages[0].strides[0], w, h, dst, out.strides[1],
                                           wh.ctypes.data, angles.ctypes.data, n_devices))
    return [out[i, : wh[i, 1], : wh[i, 0]] for i in range(n)], angles
