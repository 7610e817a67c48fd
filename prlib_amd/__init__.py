"""prlib_amd — MI355X (gfx950) implementation of PRLib's local-adaptive binarization hot path.

The product is libprlib_hip.so (hand-written HIP kernels behind the C ABI of include/prl_hip.h).
This package is the thin Python host layer used by the tests and the benchmark; the C++ host layer
with the reference's prl::binarize*(cv::Mat&, cv::Mat&, ...) signatures lives in csrc/prl/.
"""
from . import _capi, binarizations  # noqa: F401
from .denoise import denoise, nlm_planes  # noqa: F401
from .thinning import thinGuoHall, thinZhangSuen  # noqa: F401
from .background import backgroundNormalization  # noqa: F401
from .deskew import deskew, deskew_stats, find_angle as findAngle, find_orientation as findOrientation, houghp, rotate  # noqa: F401
from .chain import bitwise_not, cvtColorBGR2GRAY, cvtColorGRAY2BGR, process_pages, process_pages_host  # noqa: F401
from .binarizations import (  # noqa: F401
    FENG, NICK, NIBLACK, SAUVOLA, WOLFJOLION, binarize, binarizeFeng, binarizeNICK, binarizeNiblack,
    binarizeSauvola, binarizeWolfJolion, default_params, geometry, last_stats, make_params, morph,
    set_exec_mode, set_literal_page_budget, set_deferred_completion, finish, binarize_pages_host, PinnedPages, binarizeByLocalVariances, binarizeByLocalVariancesWithoutFilters,
)

__all__ = [
    "binarize", "binarizeSauvola", "binarizeNiblack", "binarizeWolfJolion", "binarizeNICK", "binarizeFeng", "binarizeByLocalVariances", "binarizeByLocalVariancesWithoutFilters",
    "denoise", "nlm_planes", "backgroundNormalization", "deskew", "rotate", "houghp", "findAngle", "findOrientation", "deskew_stats", "thinZhangSuen", "thinGuoHall", "cvtColorBGR2GRAY", "cvtColorGRAY2BGR", "bitwise_not", "process_pages", "process_pages_host", "make_params", "default_params", "geometry", "last_stats", "morph", "set_exec_mode", "set_literal_page_budget", "set_deferred_completion", "finish", "binarize_pages_host", "PinnedPages",
    "SAUVOLA", "NIBLACK", "WOLFJOLION", "NICK", "FENG",
]
