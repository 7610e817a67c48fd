#!/usr/bin/env python3
"""Calibrates rocprofv3 FETCH_SIZE / WRITE_SIZE for 8 B/lane streaming on this GPU: streams a known
4 GiB through k_calib_stream8.  Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prlib_amd import _capi
L = _capi.lib()
L.prl_hip_internal_calib_stream8.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
n = 4 << 30
a = torch.randint(0, 255, (n,), dtype=torch.uint8, device="cuda")
b = torch.empty_like(a)
for _ in range(3):
    assert L.prl_hip_internal_calib_stream8(a.data_ptr(), b.data_ptr(), n, torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()
print("streamed", n, "bytes x3")
