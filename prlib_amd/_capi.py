"""ctypes binding of libprlib_hip.so — the C ABI declared in include/prl_hip.h.

This module only marshals pointers, sizes and strides.  It never computes: if the shared library
is missing it raises, and if no gfx950 device is usable the library itself returns
PRL_ERR_NO_DEVICE.  There is no CPU fallback anywhere in the product path.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libprlib_hip.so")
# the same library built with -DPRL_TEST_HOOKS (`make -C prlib_amd/csrc hooks`): reads the PRL_HIP_* tuning knobs; loaded only
# by tests / tools that call use_library(HOOKS_LIB_PATH) before their first call
HOOKS_LIB_PATH = os.path.join(_HERE, "libprlib_hip_testhooks.so")


def use_library(path: str) -> None:
    """Load another build of the library (tests: the hooks build; tools: A/B of kernel experiments).  Before the first call."""
    global LIB_PATH
    if _lib is not None:
        raise RuntimeError("the library is already loaded")
    LIB_PATH = os.path.abspath(path)

PRL_OK = 0
PRL_ERR_EMPTY = 1
PRL_ERR_BAD_WINDOW = 2
PRL_ERR_BAD_CHANNELS = 3
PRL_ERR_EMPTY_RECT = 4
PRL_ERR_BAD_ARG = 5
PRL_ERR_NO_DEVICE = 6
PRL_ERR_HIP = 7
PRL_ERR_NOMEM = 8
PRL_ERR_LITERAL_BUDGET = 9

SAUVOLA, NIBLACK, WOLFJOLION, NICK, FENG = range(5)
MODE_AUTO, MODE_LITERAL = 0, 1

# every symbol include/prl_hip.h declares (tests check the library exports all of them)
EXPORTED_SYMBOLS = [
    "prl_hip_abi_version", "prl_hip_strerror", "prl_hip_last_error_detail", "prl_hip_device_count",
    "prl_hip_set_device", "prl_hip_set_exec_mode", "prl_hip_get_exec_mode", "prl_hip_last_stats",
    "prl_hip_release_workspace", "prl_hip_set_profiling", "prl_hip_last_kernel_ms", "prl_hip_last_call_ms", "prl_hip_set_deferred_completion", "prl_hip_finish", "prl_hip_default_params", "prl_hip_binarize_geometry",
    "prl_hip_binarize_batch_device", "prl_hip_binarize_pages_device", "prl_hip_binarize_host",
    "prl_hip_morph_batch_device", "prl_hip_nlm_planes_device", "prl_hip_denoise_batch_device",
    "prl_hip_denoise_host", "prl_hip_thin_batch_device", "prl_hip_thin_host",
    "prl_hip_bgr2gray_batch_device", "prl_hip_gray2bgr_batch_device", "prl_hip_invert_batch_device",
    "prl_hip_default_chain_params", "prl_hip_chain_batch_device",
    "prl_hip_bgnorm_out_channels", "prl_hip_bgnorm_batch_device", "prl_hip_bgnorm_host",
    "prl_hip_rotate_out_size", "prl_hip_rotate_batch_device", "prl_hip_houghp_device", "prl_hip_deskew_batch_device",
    "prl_hip_rotate_host", "prl_hip_deskew_host", "prl_hip_find_angle_batch_device", "prl_hip_find_angle_host", "prl_hip_last_deskew_stats", "prl_hip_reset_deskew_stats", "prl_hip_set_literal_page_budget", "prl_hip_get_literal_page_budget", "prl_hip_chain_max_out_size", "prl_hip_chain_pages_device",
    "prl_hip_binarize_batch_host", "prl_hip_page_range", "prl_hip_binarize_lv_batch_device", "prl_hip_binarize_lv_host",
    "prl_hip_chain_batch_host", "prl_hip_alloc_host", "prl_hip_free_host", "prl_hip_host_register", "prl_hip_host_unregister",
]


class BinarizeParams(C.Structure):
    """struct prl_binarize_params."""

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


class ChainParams(C.Structure):
    """struct prl_chain_params."""

    _fields_ = [("denoise", C.c_int32), ("denoise_strength", C.c_float), ("binarize", BinarizeParams), ("thin", C.c_int32),
                ("deskew", C.c_int32), ("background_normalization", C.c_int32)]


class BinarizeGeometry(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("w", "half", "padded_w", "padded_h", "out_w", "out_h")]


class BinarizeStats(C.Structure):
    _fields_ = [
        ("pixels", C.c_uint64),
        ("refined_pixels", C.c_uint64),
        ("exact_pixels", C.c_uint64),
        ("literal_pages", C.c_uint64),
        ("wolf_candidates", C.c_uint64),
        ("exact_sweep_pages", C.c_uint64),
        ("reserved", C.c_uint64 * 2),
    ]


class PrlError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(message)
        self.status = status


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). prlib_amd has no CPU fallback.")
        # One HIP runtime per process: PyTorch ships its own libamdhip64.so.7.  When torch is going to be used (it owns
        # the device tensors of the Python layer), it must be loaded FIRST so that this library binds to the same
        # runtime; loaded the other way round, two runtimes open the device and the second finds none.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        vp, sz, i = C.c_void_p, C.c_size_t, C.c_int
        P = C.POINTER
        L.prl_hip_strerror.restype = C.c_char_p
        L.prl_hip_strerror.argtypes = [i]
        L.prl_hip_last_error_detail.restype = C.c_char_p
        L.prl_hip_device_count.argtypes = [P(C.c_int)]
        L.prl_hip_set_device.argtypes = [i]
        L.prl_hip_set_exec_mode.argtypes = [i]
        L.prl_hip_last_stats.argtypes = [P(BinarizeStats)]
        L.prl_hip_set_profiling.argtypes = [i]
        L.prl_hip_last_kernel_ms.argtypes = [P(C.c_float)]
        L.prl_hip_last_call_ms.argtypes = [P(C.c_float)]
        L.prl_hip_set_deferred_completion.argtypes = [i]
        L.prl_hip_alloc_host.argtypes = [sz, P(vp)]
        L.prl_hip_free_host.argtypes = [vp]
        L.prl_hip_host_register.argtypes = [vp, sz]
        L.prl_hip_host_unregister.argtypes = [vp]
        L.prl_hip_finish.argtypes = [vp]
        L.prl_hip_default_params.argtypes = [i, P(BinarizeParams)]
        L.prl_hip_binarize_geometry.argtypes = [P(BinarizeParams), i, i, P(BinarizeGeometry)]
        L.prl_hip_binarize_batch_device.argtypes = [P(BinarizeParams), i, vp, sz, sz, i, i, vp, sz, sz, vp]
        L.prl_hip_binarize_pages_device.argtypes = [P(BinarizeParams), i, P(vp), sz, i, i, P(vp), sz, vp]
        L.prl_hip_binarize_host.argtypes = [P(BinarizeParams), vp, sz, i, i, vp, sz, vp, sz]
        L.prl_hip_morph_batch_device.argtypes = [i, i, vp, sz, sz, i, i, vp, sz, sz, vp]
        L.prl_hip_nlm_planes_device.argtypes = [i, i, C.c_float, vp, sz, sz, i, i, vp, sz, sz, vp]
        L.prl_hip_denoise_batch_device.argtypes = [i, i, C.c_float, vp, sz, sz, i, i, vp, sz, sz, vp]
        L.prl_hip_denoise_host.argtypes = [i, C.c_float, vp, sz, i, i, vp, sz]
        L.prl_hip_thin_batch_device.argtypes = [i, i, vp, sz, sz, i, i, vp, sz, sz, vp]
        L.prl_hip_thin_host.argtypes = [i, vp, sz, i, i, vp, sz]
        L.prl_hip_bgr2gray_batch_device.argtypes = [i, i, vp, sz, sz, i, i, vp, sz, sz, vp]
        L.prl_hip_gray2bgr_batch_device.argtypes = [i, i, vp, sz, sz, i, i, vp, sz, sz, vp]
        L.prl_hip_invert_batch_device.argtypes = [i, vp, sz, sz, i, i, vp, sz, sz, vp]
        L.prl_hip_default_chain_params.argtypes = [P(ChainParams)]
        L.prl_hip_default_chain_params.restype = None
        L.prl_hip_chain_batch_device.argtypes = [P(ChainParams), i, i, vp, sz, sz, i, i, vp, sz, sz, vp]
        L.prl_hip_bgnorm_out_channels.argtypes = [i]
        L.prl_hip_bgnorm_batch_device.argtypes = [i, i, vp, sz, sz, i, i, vp, sz, sz, vp]
        L.prl_hip_bgnorm_host.argtypes = [i, vp, sz, i, i, vp, sz]
        L.prl_hip_rotate_out_size.argtypes = [i, i, C.c_double, P(C.c_int), P(C.c_int)]
        L.prl_hip_rotate_batch_device.argtypes = [i, i, vp, vp, sz, sz, i, i, vp, sz, sz, vp]
        L.prl_hip_houghp_device.argtypes = [vp, sz, i, i, i, i, i, vp, i, P(C.c_int), vp]
        L.prl_hip_deskew_batch_device.argtypes = [i, i, vp, sz, sz, i, i, vp, sz, sz, vp, vp, vp]
        L.prl_hip_rotate_host.argtypes = [i, C.c_double, vp, sz, i, i, vp, sz]
        L.prl_hip_deskew_host.argtypes = [i, vp, sz, i, i, vp, sz, P(C.c_int), P(C.c_int), P(C.c_double)]
        L.prl_hip_find_angle_batch_device.argtypes = [i, vp, sz, sz, i, i, vp, vp, vp]
        L.prl_hip_find_angle_host.argtypes = [vp, sz, i, i, P(C.c_double), P(C.c_int32)]
        L.prl_hip_chain_max_out_size.argtypes = [P(ChainParams), i, i, P(C.c_int), P(C.c_int)]
        L.prl_hip_chain_pages_device.argtypes = [P(ChainParams), i, i, vp, sz, sz, i, i, vp, sz, sz, vp, vp, vp]
        L.prl_hip_binarize_batch_host.argtypes = [P(BinarizeParams), i, P(vp), sz, i, i, P(vp), sz, i]
        L.prl_hip_chain_batch_host.argtypes = [P(ChainParams), i, i, P(vp), sz, i, i, P(vp), sz, vp, vp, i]
        L.prl_hip_page_range.argtypes = [i, i, i, P(C.c_int), P(C.c_int)]
        L.prl_hip_binarize_lv_batch_device.argtypes = [i, i, C.c_double, i, C.c_double, vp, sz, sz, i, i, vp, sz, sz, vp]
        L.prl_hip_binarize_lv_host.argtypes = [i, C.c_double, i, C.c_double, vp, sz, i, i, vp, sz]
        _lib = L
    return _lib


def check(status: int) -> None:
    if status != PRL_OK:
        L = lib()
        msg = L.prl_hip_strerror(status).decode()
        detail = L.prl_hip_last_error_detail().decode()
        raise PrlError(status, f"{msg}" + (f" [{detail}]" if detail and status in (PRL_ERR_HIP, PRL_ERR_NO_DEVICE, PRL_ERR_NOMEM, PRL_ERR_BAD_ARG, PRL_ERR_LITERAL_BUDGET) else ""))
