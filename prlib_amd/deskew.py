"""Host-side mirror of prl::deskew / prl::rotate (src/deskew/deskew.cpp:208-251, src/rotate.cpp:35-72) over the C ABI.

    bool prl::deskew(const cv::Mat& inputImage, cv::Mat& outputImage)
    void prl::rotate(const cv::Mat& inputImage, cv::Mat& outputImage, double angle)

Pages are torch CUDA uint8 tensors N x H x W x C (C in 1, 3, 4; a 3-d tensor is N x H x W gray).  Every deskewed page
has its own size (max(W,H)^2 when an angle was found, W x H otherwise), so `deskew` returns a list of views into one
N x L x L x C buffer together with the angles findAngle produced.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi


def _pages4(t):
    import torch

    if t.dtype != torch.uint8 or not t.is_cuda or not t.is_contiguous() or t.dim() not in (3, 4):
        raise TypeError("expected a contiguous uint8 CUDA tensor N x H x W [x C]")
    return t if t.dim() == 4 else t[:, :, :, None]


def rotate_out_size(width: int, height: int, angle: float):
    ow, oh = C.c_int(0), C.c_int(0)
    _capi.check(_capi.lib().prl_hip_rotate_out_size(width, height, float(angle), C.byref(ow), C.byref(oh)))
    return ow.value, oh.value


def rotate(pages, angles):
    """prl::rotate per page -> list of tensors (views of one buffer)."""
    import torch

    t4 = _pages4(pages)
    n, h, w, c = t4.shape
    ang = np.ascontiguousarray(np.broadcast_to(np.asarray(angles, dtype=np.float64), (n,)))
    sizes = [rotate_out_size(w, h, a) for a in ang]
    mw, mh = max(s[0] for s in sizes), max(s[1] for s in sizes)
    buf = torch.empty((n, mh, mw, c), dtype=torch.uint8, device=t4.device)
    L = _capi.lib()
    _capi.check(L.prl_hip_set_device(t4.device.index or 0))
    stream = torch.cuda.current_stream(t4.device).cuda_stream
    _capi.check(L.prl_hip_rotate_batch_device(n, c, ang.ctypes.data, t4.data_ptr(), t4.stride(0), t4.stride(1), w, h,
                                              buf.data_ptr(), buf.stride(0), buf.stride(1), stream))
    outs = [buf[i, :oh, :ow] for i, (ow, oh) in enumerate(sizes)]
    return outs if pages.dim() == 4 else [o[:, :, 0] for o in outs]


def deskew(pages):
    """prl::deskew per page -> (list of tensors, angles in degrees as float64 numpy)."""
    import torch

    t4 = _pages4(pages)
    n, h, w, c = t4.shape
    ln = max(w, h)
    buf = torch.empty((n, ln, ln, c), dtype=torch.uint8, device=t4.device)
    wh = np.zeros((n, 2), dtype=np.int32)
    ang = np.zeros(n, dtype=np.float64)
    L = _capi.lib()
    _capi.check(L.prl_hip_set_device(t4.device.index or 0))
    stream = torch.cuda.current_stream(t4.device).cuda_stream
    _capi.check(L.prl_hip_deskew_batch_device(n, c, t4.data_ptr(), t4.stride(0), t4.stride(1), w, h, buf.data_ptr(),
                                              buf.stride(0), buf.stride(1), wh.ctypes.data, ang.ctypes.data, stream))
    outs = [buf[i, : wh[i, 1], : wh[i, 0]] for i in range(n)]
    return (outs if pages.dim() == 4 else [o[:, :, 0] for o in outs]), ang


def houghp(image, threshold: int, line_length: int, line_gap: int) -> np.ndarray:
    """cv::HoughLinesP(image, lines, 1, CV_PI/180, threshold, line_length, line_gap) on one H x W CUDA page."""
    import torch

    if image.dtype != torch.uint8 or not image.is_cuda or image.dim() != 2 or image.stride(1) != 1:
        raise TypeError("expected an H x W uint8 CUDA tensor")
    h, w = image.shape
    L = _capi.lib()
    _capi.check(L.prl_hip_set_device(image.device.index or 0))
    stream = torch.cuda.current_stream(image.device).cuda_stream
    cap = 4096
    while True:
        lines = np.empty((cap, 4), dtype=np.int32)
        n = C.c_int(0)
        _capi.check(L.prl_hip_houghp_device(image.data_ptr(), image.stride(0), w, h, threshold, line_length, line_gap,
                                            lines.ctypes.data, cap, C.byref(n), stream))
        if n.value <= cap:
            return lines[: n.value].copy()
        cap = n.value
