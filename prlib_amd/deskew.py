"""Host-side mirror of prl::deskew / prl::rotate (src/deskew/deskew.cpp:208-251, src/rotate.cpp:35-72) over the C ABI.

    bool prl::deskew(const cv::Mat& inputImage, cv::Mat& outputImage)
    double prl::findAngle(const cv::Mat& inputImage)                       (src/deskew/deskew.h:62, deskew.cpp:139-205)
    double prl::findOrientation(const cv::Mat& inputImage)                 (src/deskew/deskew.h:52, deskew.cpp:70-136)
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


def find_angle(pages, return_segments: bool = False):
    """prl::findAngle per 1-channel page (N x H x W, or one H x W page) -> angles in degrees (float64 numpy; a float for a
    single page).  The points of the Hough transform are the pixels != 255, as after the reference's bitwise_not."""
    import torch

    single = pages.dim() == 2
    t = pages[None] if single else pages
    if t.dtype != torch.uint8 or not t.is_cuda or t.dim() != 3 or t.stride(2) != 1:
        raise TypeError("expected an [N x] H x W uint8 CUDA tensor (cv::HoughLinesP takes CV_8UC1)")
    n, h, w = t.shape
    ang = np.zeros(n, dtype=np.float64)
    nseg = np.zeros(n, dtype=np.int32)
    L = _capi.lib()
    _capi.check(L.prl_hip_set_device(t.device.index or 0))
    stream = torch.cuda.current_stream(t.device).cuda_stream
    _capi.check(L.prl_hip_find_angle_batch_device(n, t.data_ptr(), t.stride(0), t.stride(1), w, h, ang.ctypes.data,
                                                  nseg.ctypes.data, stream))
    a = float(ang[0]) if single else ang
    return (a, (int(nseg[0]) if single else nseg)) if return_segments else a


def find_orientation(pages):
    """prl::findOrientation.  The reference builds its gray page only for 3-channel input (deskew.cpp:73-76) and then needs
    Leptonica's pixOrientDetectDwa (:91); for the 1-channel page prl::deskew hands it (:238) the gray page is empty and the
    function returns 0 through its NULL-pix exit (:81-84) or throws, depending on the OpenCV version.  Canonical result: 0.0
    (SURVEY.md Appendix D).  `pages`: H x W or N x H x W (1 channel).  A 3-channel page (H x W x 3 / N x H x W x 3) is where the
    reference really runs Leptonica's detector: not provided here, NotImplementedError instead of a silent 0."""
    if pages.dim() == 4 or (pages.dim() == 3 and pages.shape[-1] == 3):
        raise NotImplementedError("prl::findOrientation on 3-channel pages needs Leptonica's pixOrientDetectDwa (deskew.cpp:91)")
    return 0.0 if pages.dim() == 2 else np.zeros(pages.shape[0], dtype=np.float64)


class DeskewStats(C.Structure):
    """prl_deskew_stats (include/prl_hip.h): totals of the HoughLinesP searches since the last reset."""
    _fields_ = [("pages", C.c_uint64), ("points", C.c_uint64), ("segments", C.c_uint64), ("segment_capacity", C.c_uint64),
                ("max_page_points", C.c_uint64), ("max_page_segments", C.c_uint64), ("min_page_headroom", C.c_int64),
                ("reserved", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_ if k != "reserved"}


def deskew_stats(reset: bool = False) -> DeskewStats:
    st = DeskewStats()
    _capi.check(_capi.lib().prl_hip_last_deskew_stats(C.byref(st)))
    if reset:
        _capi.check(_capi.lib().prl_hip_reset_deskew_stats())
    return st


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
