"""GPU parity of prl::binarizeByLocalVariances / ...WithoutFilters against the CPU oracle.

WithoutFilters is float32 arithmetic on integer-valued sums with a fixed operation order: bit-exact.  The filtered
variant goes through float32 log / exp (cv::log / cv::exp in the reference, logf / expf in the oracle, the device's logf
/ expf here): last-bit differences move a pixel of the 8-bit maps across a rounding boundary now and then, so the
stated tolerance is <= 1e-3 of the pixels differing (measured: ~1e-5)."""
import numpy as np
import pytest

from prlib_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _colour(h, w, seed, skew=0.0, shading=0.2):
    g = synth.text_page_numpy(h, w, seed, skew_deg=skew, shading=shading)
    rng = np.random.default_rng(seed)
    return np.clip(g[..., None].astype(np.int32) + rng.normal(0, 5, (h, w, 3)), 0, 255).round().astype(np.uint8)


@pytest.mark.parametrize("shape", [(200, 260), (97, 131), (33, 40), (8, 32), (1, 1), (257, 65), (530, 777), (18, 3000)])
def test_without_filters_is_bit_exact(prl, oracle, cuda_device, shape):
    import torch

    pages = np.stack([_colour(shape[0], shape[1], s) for s in (1, 2)])
    for coeff, mv in ((0.125, 10), (0.5, 25), (0.0, 0)):
        got = prl.binarizeByLocalVariancesWithoutFilters(torch.from_numpy(pages).to(cuda_device), coeff, mv).cpu().numpy()
        for i in range(2):
            want = oracle.binarize_lv_nofilters(pages[i], coeff, mv)
            assert np.array_equal(got[i], want), (shape, coeff, mv, int((got[i] != want).sum()))


@pytest.mark.parametrize("shape", [(200, 260), (97, 131), (64, 48), (300, 411), (140, 700)])
def test_with_filters_within_tolerance(prl, oracle, cuda_device, shape):
    import torch

    pages = np.stack([_colour(shape[0], shape[1], s, skew=1.0) for s in (3, 4)])
    for coeff, mv, gamma in ((0.125, 25, 2.0), (0.3, 10, 2.0), (0.125, 25, 1.5)):
        got = prl.binarizeByLocalVariances(torch.from_numpy(pages).to(cuda_device), coeff, mv, gamma).cpu().numpy()
        for i in range(2):
            want = oracle.binarize_lv(pages[i], coeff, mv, gamma)
            bad = int((got[i] != want).sum())
            assert bad <= max(2, TOL * want.size), (shape, coeff, mv, gamma, bad, want.size)
            assert set(np.unique(got[i])) <= {0, 255}


def test_lv_flat_noise_host_entry_and_errors(prl, oracle, cuda_device):
    import torch

    flat = np.full((50, 60, 3), 120, np.uint8)          # variance 0.01 everywhere: nothing exceeds 10 -> all black
    assert prl.binarizeByLocalVariances(torch.from_numpy(flat).to(cuda_device)).max().item() == 0
    assert prl.binarizeByLocalVariancesWithoutFilters(torch.from_numpy(flat).to(cuda_device)).max().item() == 0
    rng = np.random.default_rng(9)
    noise = rng.integers(0, 256, (80, 90, 3), dtype=np.uint8)
    assert np.array_equal(prl.binarizeByLocalVariancesWithoutFilters(torch.from_numpy(noise).to(cuda_device)).cpu().numpy(),
                          oracle.binarize_lv_nofilters(noise))
    page = _colour(120, 150, 5)
    assert np.array_equal(prl.binarizeByLocalVariancesWithoutFilters(page), oracle.binarize_lv_nofilters(page))   # host entry
    got = prl.binarizeByLocalVariances(page)
    assert int((got != oracle.binarize_lv(page)).sum()) <= max(2, TOL * got.size)
    with pytest.raises(TypeError):
        prl.binarizeByLocalVariances(torch.zeros((10, 10), dtype=torch.uint8, device=cuda_device))
    with pytest.raises(ValueError):
        prl.binarizeByLocalVariances(np.zeros((0, 0, 3), np.uint8))
