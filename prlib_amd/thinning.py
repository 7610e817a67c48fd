"""Host-side mirror of prl::thinZhangSuen (src/thinning/thinZhangSuen.h) and prl::thinGuoHall
(src/thinning/thinGuoHall.h) over the C ABI.  1-channel u8 images; foreground = odd pixel values (the reference's
`&= 1`), i.e. WHITE (255) is thinned, as in the reference.  The reference throws std::invalid_argument for an empty
image (thinZhangSuen.cpp:59-62): ValueError here.
"""
from __future__ import annotations

import numpy as np

from . import _capi

ZHANGSUEN, GUOHALL = 0, 1


def _thin(image, method: int, out=None):
    L = _capi.lib()
    if isinstance(image, np.ndarray):
        if image.ndim != 2 or image.dtype != np.uint8:
            raise TypeError("expected a 2-D uint8 image (convert colour images with cvtColor first)")
        if image.size == 0:
            raise ValueError("Input image for thinning is empty")
        img = np.ascontiguousarray(image)
        res = np.empty_like(img)
        h, w = img.shape
        _capi.check(L.prl_hip_thin_host(method, img.ctypes.data, img.strides[0], w, h, res.ctypes.data, res.strides[0]))
        return res
    import torch

    t = image
    squeeze = t.dim() == 2
    if squeeze:
        t = t.unsqueeze(0)
    if t.dtype != torch.uint8 or not t.is_cuda or t.dim() != 3 or t.stride(2) != 1:
        raise TypeError("expected a uint8 CUDA tensor [N,] H x W")
    if t.numel() == 0:
        raise ValueError("Input image for thinning is empty")
    n, h, w = t.shape
    if out is None:
        out = torch.empty_like(t)
    elif out.dim() == 2:
        out = out.unsqueeze(0)
    if out.shape != t.shape or out.stride(2) != 1:
        raise ValueError("output tensor has the wrong shape")
    _capi.check(L.prl_hip_set_device(t.device.index or 0))
    stream = torch.cuda.current_stream(t.device).cuda_stream
    _capi.check(L.prl_hip_thin_batch_device(method, n, t.data_ptr(), t.stride(0), t.stride(1), w, h,
                                            out.data_ptr(), out.stride(0), out.stride(1), stream))
    return out[0] if squeeze else out


def thinZhangSuen(inputImage, out=None):
    """prl::thinZhangSuen(cv::Mat& inputImage, cv::Mat& outputImage)."""
    return _thin(inputImage, ZHANGSUEN, out)


def thinGuoHall(inputImage, out=None):
    """prl::thinGuoHall(cv::Mat& inputImage, cv::Mat& outputImage)."""
    return _thin(inputImage, GUOHALL, out)
