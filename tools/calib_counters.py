#!/usr/bin/env python3
"""Calibrates rocprofv3 FETCH_SIZE / WRITE_SIZE for 8 B/lane streaming on this GPU: streams a known
4 GiB through k_calib_stream8 (tools/ubench/stream_probe.hip).  Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench", "libstream_probe.so"))   # bench tooling, not the product library
L.prl_probe_calib_stream8.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
n = 4 << 30
a = torch.randint(0, 255, (n,), dtype=torch.uint8, device="cuda")
b = torch.empty_like(a)
for _ in range(3):
    assert L.prl_probe_calib_stream8(a.data_ptr(), b.data_ptr(), n, torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()
print("streamed", n, "bytes x3")
