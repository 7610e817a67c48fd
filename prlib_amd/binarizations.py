"""Host-side mirror of PRLib's local-adaptive binarizers over the C ABI (include/prl_hip.h).

Function names, argument order, defaults and error behaviour follow the reference headers:
  prl::binarizeSauvola     src/binarizations/binarizeSauvola.h:43-47
  prl::binarizeNiblack     src/binarizations/binarizeNiblack.h:43-47
  prl::binarizeWolfJolion  src/binarizations/binarizeWolfJolion.h:43-47
  prl::binarizeNICK        src/binarizations/binarizeNICK.h:43-47
  prl::binarizeFeng        src/binarizations/binarizeFeng.h:46-53
The reference throws std::invalid_argument for an empty image or a bad window; here that is a
ValueError.  Inputs are either a numpy uint8 page (H x W, staged through the device by the library)
or a torch uint8 CUDA tensor (H x W or N x H x W, processed in place on the device: torch is only
the owner of the device memory and the stream).  Compute always happens in libprlib_hip.so.
"""
from __future__ import annotations

import ctypes as C
import weakref

import numpy as np

from . import _capi
from ._capi import FENG, NICK, NIBLACK, SAUVOLA, WOLFJOLION, BinarizeGeometry, BinarizeParams, BinarizeStats

METHODS = {"sauvola": SAUVOLA, "niblack": NIBLACK, "wolfjolion": WOLFJOLION, "nick": NICK, "feng": FENG}


def default_params(method: int) -> BinarizeParams:
    p = BinarizeParams()
    _capi.check(_capi.lib().prl_hip_default_params(method, C.byref(p)))
    return p


def make_params(method: int, window_size=None, k=None, morph_iterations=None,
                alpha1=None, k1=None, k2=None, gamma=None) -> BinarizeParams:
    p = default_params(method)
    if window_size is not None:
        p.window_size = int(window_size)
    if k is not None:
        p.k = float(k)
    if morph_iterations is not None:
        p.morph_iterations = int(morph_iterations)
    if alpha1 is not None:
        p.feng_alpha1 = float(alpha1)
    if k1 is not None:
        p.feng_k1 = float(k1)
    if k2 is not None:
        p.feng_k2 = float(k2)
    if gamma is not None:
        p.feng_gamma = float(gamma)
    return p


def _raise_like_reference(status: int):
    L = _capi.lib()
    msg = L.prl_hip_strerror(status).decode()
    if status in (_capi.PRL_ERR_EMPTY, _capi.PRL_ERR_BAD_WINDOW):
        raise ValueError(msg)  # std::invalid_argument in the reference (binarizeSauvola.cpp:38-47)
    _capi.check(status)


def geometry(params: BinarizeParams, width: int, height: int) -> BinarizeGeometry:
    g = BinarizeGeometry()
    st = _capi.lib().prl_hip_binarize_geometry(C.byref(params), int(width), int(height), C.byref(g))
    if st != _capi.PRL_OK:
        _raise_like_reference(st)
    return g


def last_stats() -> BinarizeStats:
    s = BinarizeStats()
    _capi.check(_capi.lib().prl_hip_last_stats(C.byref(s)))
    return s


def _free_pinned(address: int) -> None:
    try:
        _capi.lib().prl_hip_free_host(C.c_void_p(address))
    except Exception:   # interpreter shutdown
        pass


class PinnedPages:
    """N x H x W uint8 pages in pinned host memory (prl_hip_alloc_host): `array` is a numpy view.  Pages in such memory are
    moved by DMA directly (no bounce copies).

    Lifetime: the block belongs to the BUFFER the views are made of, not to this wrapper - it is freed when the last numpy view
    of it (`array`, slices, `list(pin.array)`) is gone, so `PinnedPages(n, h, w).array` or pages that outlive the wrapper stay
    valid (and valid DMA targets).  close() / the context manager free it at once; the caller then must not touch views it
    still holds."""

    def __init__(self, n: int, h: int, w: int):
        ptr = C.c_void_p()
        _capi.check(_capi.lib().prl_hip_alloc_host(max(1, n * h * w), C.byref(ptr)))
        buf = (C.c_uint8 * (n * h * w)).from_address(ptr.value)
        # every view keeps `buf` alive through its .base chain; the finalizer runs when the last one dies (or at close())
        self._finalizer = weakref.finalize(buf, _free_pinned, ptr.value)
        self.array = np.frombuffer(buf, dtype=np.uint8).reshape(n, h, w)

    def close(self):
        self.array = None
        self._finalizer()   # idempotent

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def binarize_pages_host(pages, params: BinarizeParams, n_devices: int = 0, out=None):
    """prl_hip_binarize_batch_host: a list (or N x H x W array) of equal-size uint8 host pages, sharded over the node's
    GPUs by the library (upload / kernels / download pipelined per device).  Returns an N x out_h x out_w array (`out`, if
    given - e.g. a PinnedPages array, which the DMA engines then write directly)."""
    pages = [np.ascontiguousarray(p) if p.strides[1] != 1 else p for p in pages]
    n = len(pages)
    h, w = pages[0].shape
    if any(p.shape != (h, w) or p.dtype != np.uint8 for p in pages):
        raise TypeError("expected equal-size uint8 pages")
    g = geometry(params, w, h)
    if out is None:
        out = np.empty((n, g.out_h, g.out_w), dtype=np.uint8)
    elif out.shape != (n, g.out_h, g.out_w) or out.dtype != np.uint8 or out.strides[2] != 1:
        raise ValueError("output array has the wrong shape")
    if any(p.strides[0] != pages[0].strides[0] for p in pages):
        pages = [np.ascontiguousarray(p) for p in pages]
    src = (C.c_void_p * n)(*[p.ctypes.data for p in pages])
    dst = (C.c_void_p * n)(*[out[i].ctypes.data for i in range(n)])
    st = _capi.lib().prl_hip_binarize_batch_host(C.byref(params), n, src, pages[0].strides[0], w, h, dst, out.strides[1], n_devices)
    if st != _capi.PRL_OK:
        _raise_like_reference(st)
    return out


def set_deferred_completion(enabled: bool) -> None:
    """prl_hip_set_deferred_completion: binarize calls return right after enqueuing; call finish() before using the masks."""
    _capi.check(_capi.lib().prl_hip_set_deferred_completion(1 if enabled else 0))


def finish(device=None) -> None:
    """prl_hip_finish on torch's current stream of `device` (default: the current device)."""
    import torch

    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    L = _capi.lib()
    _capi.check(L.prl_hip_set_device(dev.index or 0))
    _capi.check(L.prl_hip_finish(torch.cuda.current_stream(dev).cuda_stream))


def set_literal_page_budget(max_pages: int) -> None:
    """prl_hip_set_literal_page_budget: a call that would redo more than `max_pages` pages literally fails with
    PRL_ERR_LITERAL_BUDGET instead (-1: no limit).  The cost bound a service puts on hostile input."""
    _capi.check(_capi.lib().prl_hip_set_literal_page_budget(int(max_pages)))


def set_exec_mode(mode: int) -> None:
    _capi.check(_capi.lib().prl_hip_set_exec_mode(mode))


def binarize(image, params: BinarizeParams, out=None, return_padded: bool = False):
    """Run one of the five binarizers.

    numpy H x W uint8       -> numpy mask (out_h x out_w); with return_padded also the replicate-padded
                               page the reference leaves in the caller's input Mat.
    torch CUDA uint8 tensor -> torch mask on the same device ([N,] out_h x out_w view of a
                               256-byte-pitched buffer), enqueued on torch's current stream.
    """
    if isinstance(image, np.ndarray):
        return _binarize_numpy(image, params, return_padded, out)
    return _binarize_torch(image, params, out)


def _binarize_numpy(img: np.ndarray, params: BinarizeParams, return_padded: bool, out=None):
    if img.ndim != 2 or img.dtype != np.uint8:
        raise TypeError("expected a 2-D uint8 page (convert colour pages with cvtColor first)")
    h, w = img.shape
    g = geometry(params, w, h)
    if img.size and img.strides[1] != 1:
        img = np.ascontiguousarray(img)
    if out is None:
        out = np.empty((g.out_h, g.out_w), dtype=np.uint8)
    elif out.shape != (g.out_h, g.out_w) or out.dtype != np.uint8 or out.strides[1] != 1:
        raise ValueError("output array has the wrong shape")
    padded = np.empty((g.padded_h, g.padded_w), dtype=np.uint8) if return_padded else None
    st = _capi.lib().prl_hip_binarize_host(
        C.byref(params), img.ctypes.data, img.strides[0], w, h, out.ctypes.data, out.strides[0],
        padded.ctypes.data if return_padded else None, padded.strides[0] if return_padded else 0)
    if st != _capi.PRL_OK:
        _raise_like_reference(st)
    return (out, padded) if return_padded else out


def alloc_output(n_pages: int, out_w: int, out_h: int, device, pitch_align: int = 256):
    """Pitched output buffer: returns (N x out_h x pitch tensor, pitch)."""
    import torch

    pitch = (out_w + pitch_align - 1) // pitch_align * pitch_align
    return torch.empty((n_pages, out_h, pitch), dtype=torch.uint8, device=device), pitch


def _binarize_torch(pages, params: BinarizeParams, out=None):
    import torch

    if pages.dtype != torch.uint8 or not pages.is_cuda:
        raise TypeError("expected a uint8 CUDA tensor")
    squeeze = pages.dim() == 2
    if squeeze:
        pages = pages.unsqueeze(0)
    if pages.dim() != 3 or pages.stride(2) != 1:
        raise TypeError("expected N x H x W with unit pixel stride")
    n, h, w = pages.shape
    g = geometry(params, w, h)
    if out is None:
        out, _ = alloc_output(n, g.out_w, g.out_h, pages.device)
    if out.dim() != 3 or out.shape[0] != n or out.shape[1] != g.out_h or out.shape[2] < g.out_w or out.stride(2) != 1:
        raise ValueError("output buffer has the wrong shape")
    L = _capi.lib()
    _capi.check(L.prl_hip_set_device(pages.device.index or 0))
    stream = torch.cuda.current_stream(pages.device).cuda_stream
    st = L.prl_hip_binarize_batch_device(
        C.byref(params), n, pages.data_ptr(), pages.stride(0), pages.stride(1), w, h,
        out.data_ptr(), out.stride(0), out.stride(1), stream)
    if st != _capi.PRL_OK:
        _raise_like_reference(st)
    res = out[:, :, : g.out_w]
    return res[0] if squeeze else res


# ---- the reference's five entry points, same names / argument order / defaults -------------------

def binarizeSauvola(inputImage, windowSize: int = 101, thresholdCoefficient: float = 0.01,
                    morphIterationCount: int = 2, **kw):
    """prl::binarizeSauvola (binarizeSauvola.h:43-47)."""
    return binarize(inputImage, make_params(SAUVOLA, windowSize, thresholdCoefficient, morphIterationCount), **kw)


def binarizeNiblack(inputImage, windowSize: int = 101, thresholdCoefficient: float = 0.01,
                    morphIterationCount: int = 2, **kw):
    """prl::binarizeNiblack (binarizeNiblack.h:43-47)."""
    return binarize(inputImage, make_params(NIBLACK, windowSize, thresholdCoefficient, morphIterationCount), **kw)


def binarizeWolfJolion(inputImage, windowSize: int = 101, thresholdCoefficient: float = 0.01,
                       morphIterationCount: int = 2, **kw):
    """prl::binarizeWolfJolion (binarizeWolfJolion.h:43-47)."""
    return binarize(inputImage, make_params(WOLFJOLION, windowSize, thresholdCoefficient, morphIterationCount), **kw)


def binarizeNICK(inputImage, windowSize: int = 21, thresholdCoefficient: float = -0.01,
                 morphIterationCount: int = 0, **kw):
    """prl::binarizeNICK (binarizeNICK.h:43-47)."""
    return binarize(inputImage, make_params(NICK, windowSize, thresholdCoefficient, morphIterationCount), **kw)


def binarizeFeng(inputImage, windowSize: int = 21, thresholdCoefficient_alpha1: float = 0.75,
                 thresholdCoefficient_k1: float = 0.2, thresholdCoefficient_k2: float = 0.03,
                 thresholdCoefficient_gamma: float = 2.0, morphIterationCount: int = 2, **kw):
    """prl::binarizeFeng (binarizeFeng.h:46-53)."""
    p = make_params(FENG, windowSize, None, morphIterationCount, thresholdCoefficient_alpha1,
                    thresholdCoefficient_k1, thresholdCoefficient_k2, thresholdCoefficient_gamma)
    return binarize(inputImage, p, **kw)


def morph(mask, iterations: int, out=None):
    """The dilate/erode pair of binarizeSauvola.cpp:125-134 on a CUDA uint8 mask ([N,] H x W)."""
    import torch

    squeeze = mask.dim() == 2
    if squeeze:
        mask = mask.unsqueeze(0)
    n, h, w = mask.shape
    if out is None:
        out, _ = alloc_output(n, w, h, mask.device)
    L = _capi.lib()
    _capi.check(L.prl_hip_set_device(mask.device.index or 0))
    stream = torch.cuda.current_stream(mask.device).cuda_stream
    _capi.check(L.prl_hip_morph_batch_device(int(iterations), n, mask.data_ptr(), mask.stride(0), mask.stride(1),
                                             w, h, out.data_ptr(), out.stride(0), out.stride(1), stream))
    res = out[:, :, :w]
    return res[0] if squeeze else res


def _lv(inputImage, with_filters, coeff, min_result_variance, gamma, out=None):
    L = _capi.lib()
    if isinstance(inputImage, np.ndarray):
        img = np.ascontiguousarray(inputImage)
        if img.size == 0:
            raise ValueError("binarizeByLocalVariances: Input inputImage for binarization is empty")
        if img.ndim != 3 or img.shape[2] != 3 or img.dtype != np.uint8:
            raise TypeError("expected an H x W x 3 uint8 image")
        h, w = img.shape[:2]
        res = np.empty((h, w), dtype=np.uint8)
        _capi.check(L.prl_hip_binarize_lv_host(with_filters, coeff, min_result_variance, gamma, img.ctypes.data, img.strides[0],
                                               w, h, res.ctypes.data, res.strides[0]))
        return res
    import torch

    t = inputImage
    squeeze = t.dim() == 3
    if squeeze:
        t = t.unsqueeze(0)
    if t.dtype != torch.uint8 or not t.is_cuda or t.dim() != 4 or t.shape[3] != 3 or t.stride(3) != 1 or t.stride(2) != 3:
        raise TypeError("expected a uint8 CUDA tensor [N,] H x W x 3")
    n, h, w = t.shape[:3]
    o = torch.empty((n, h, w), dtype=torch.uint8, device=t.device) if out is None else (out.unsqueeze(0) if out.dim() == 2 else out)
    _capi.check(L.prl_hip_set_device(t.device.index or 0))
    stream = torch.cuda.current_stream(t.device).cuda_stream
    _capi.check(L.prl_hip_binarize_lv_batch_device(n, with_filters, coeff, min_result_variance, gamma, t.data_ptr(), t.stride(0),
                                                   t.stride(1), w, h, o.data_ptr(), o.stride(0), o.stride(1), stream))
    return o[0] if squeeze else o


def binarizeByLocalVariances(inputImage, varianceThresholdCoeff: float = 0.125, minResultVariance: int = 25,
                             gamma: float = 2.0, out=None):
    """src/binarizations/binarizeByLocalVariances.h:8-9"""
    return _lv(inputImage, 1, float(varianceThresholdCoeff), int(minResultVariance), float(gamma), out)


def binarizeByLocalVariancesWithoutFilters(inputImage, varianceThresholdCoeff: float = 0.125, minResultVariance: int = 10, out=None):
    """src/binarizations/binarizeByLocalVariances.h:11-12"""
    return _lv(inputImage, 0, float(varianceThresholdCoeff), int(minResultVariance), 2.0, out)
