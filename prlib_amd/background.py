"""Host-side mirror of prl::backgroundNormalization (src/backgroundNormalization.cpp:36-61) over the C ABI.

    void prl::backgroundNormalization(const cv::Mat& inputImage, cv::Mat& outputImage)
      = leptonicaToOpenCV(pixBackgroundNormSimple(opencvToLeptonica(input), NULL, NULL))

torch CUDA uint8 [N,] H x W (1 channel) or [N,] H x W x {3,4}; a 4-channel input comes back with 3 channels, as the
reference's converter does (src/formatConvert.cpp:193-206).  numpy arrays go through the library's host entry.
"""
from __future__ import annotations

import numpy as np

from . import _capi


def backgroundNormalization(inputImage, out=None):
    L = _capi.lib()
    if isinstance(inputImage, np.ndarray):
        img = np.ascontiguousarray(inputImage)
        if img.dtype != np.uint8 or img.ndim not in (2, 3):
            raise TypeError("expected an H x W [x C] uint8 image")
        h, w = img.shape[:2]
        c = 1 if img.ndim == 2 else img.shape[2]
        och = 1 if c == 1 else 3
        res = np.empty((h, w) if img.ndim == 2 else (h, w, och), dtype=np.uint8)
        _capi.check(L.prl_hip_bgnorm_host(c, img.ctypes.data, img.strides[0], w, h, res.ctypes.data, res.strides[0]))
        return res
    import torch

    t = inputImage
    if t.dtype != torch.uint8 or not t.is_cuda or not t.is_contiguous():
        raise TypeError("expected a contiguous uint8 CUDA tensor")
    gray = t.dim() == 2 or (t.dim() == 3 and t.shape[-1] > 4)
    if t.dim() == 2:
        t4 = t[None, :, :, None]
    elif t.dim() == 3:
        t4 = t[:, :, :, None] if gray else t[None]
    else:
        t4 = t
    n, h, w, c = t4.shape
    och = 1 if c == 1 else 3
    res = torch.empty((n, h, w, och), dtype=torch.uint8, device=t.device) if out is None else out.view(n, h, w, och)
    _capi.check(L.prl_hip_set_device(t.device.index or 0))
    stream = torch.cuda.current_stream(t.device).cuda_stream
    _capi.check(L.prl_hip_bgnorm_batch_device(n, c, t4.data_ptr(), t4.stride(0), t4.stride(1), w, h, res.data_ptr(),
                                              res.stride(0), res.stride(1), stream))
    if t.dim() == 2:
        return res[0, :, :, 0]
    if t.dim() == 3:
        return res[:, :, :, 0] if gray else res[0]
    return res
