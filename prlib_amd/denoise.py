"""Host-side mirror of prl::denoise (src/denoise/denoiseNLM.h:32, denoiseNLM.cpp:29-32) over the C ABI.

    void prl::denoise(const cv::Mat& inputImage, cv::Mat& outputImage, double strength = 5.5)
      = cv::fastNlMeansDenoisingColored(inputImage, outputImage, strength)     [hColor 3, template 7, search 21]

numpy H x W x {3,4} uint8 -> numpy (staged through the device by the library);
torch CUDA uint8 [N,] H x W x {3,4} -> torch tensor on the same device.  Other channel counts raise, as
the reference's OpenCV call does ("Type of input image should be CV_8UC3 or CV_8UC4!").
"""
from __future__ import annotations

import numpy as np

from . import _capi


def denoise(inputImage, strength: float = 5.5, out=None):
    if isinstance(inputImage, np.ndarray):
        img = np.ascontiguousarray(inputImage)
        if img.ndim != 3 or img.dtype != np.uint8:
            raise TypeError("expected an H x W x C uint8 image")
        h, w, c = img.shape
        res = np.empty_like(img)
        _capi.check(_capi.lib().prl_hip_denoise_host(c, float(strength), img.ctypes.data, img.strides[0], w, h,
                                                     res.ctypes.data, res.strides[0]))
        return res
    import torch

    t = inputImage
    squeeze = t.dim() == 3
    if squeeze:
        t = t.unsqueeze(0)
    if t.dtype != torch.uint8 or not t.is_cuda or t.dim() != 4 or not t.is_contiguous():
        raise TypeError("expected a contiguous uint8 CUDA tensor [N,] H x W x C")
    n, h, w, c = t.shape
    if out is None:
        out = torch.empty_like(t)
    L = _capi.lib()
    _capi.check(L.prl_hip_set_device(t.device.index or 0))
    stream = torch.cuda.current_stream(t.device).cuda_stream
    _capi.check(L.prl_hip_denoise_batch_device(n, c, float(strength), t.data_ptr(), t.stride(0), t.stride(1), w, h,
                                               out.data_ptr(), out.stride(0), out.stride(1), stream))
    return out[0] if squeeze else out


def nlm_planes(planes, h: float, out=None):
    """cv::fastNlMeansDenoising core on 1/2/3 interleaved u8 planes (CUDA tensor [N,] H x W [x C])."""
    import torch

    t = planes
    if t.dim() == 2:
        t4 = t[None, :, :, None]
    elif t.dim() == 3:
        t4 = t[None]
    else:
        t4 = t
    t4 = t4.contiguous()
    n, hh, w, c = t4.shape
    res = torch.empty_like(t4) if out is None else out.view(t4.shape)
    L = _capi.lib()
    _capi.check(L.prl_hip_set_device(t4.device.index or 0))
    stream = torch.cuda.current_stream(t4.device).cuda_stream
    _capi.check(L.prl_hip_nlm_planes_device(n, c, float(h), t4.data_ptr(), t4.stride(0), t4.stride(1), w, hh,
                                            res.data_ptr(), res.stride(0), res.stride(1), stream))
    return res.view(t.shape)
