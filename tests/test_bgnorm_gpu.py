"""GPU parity of prl::backgroundNormalization against the CPU oracle: bit-exact."""
import glob
import os

import numpy as np
import pytest

from prlib_amd import synth

pytestmark = pytest.mark.gpu

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "0*.npz")))


def _gray_cases():
    yield "synthetic", synth.page_numpy(203, 317, index=11)
    yield "text_shaded", synth.text_page_numpy(330, 250, 3, skew_deg=1.5, shading=0.6)
    p = synth.page_numpy(150, 200, index=5)
    p[30:110, 40:160] = 20
    yield "holes", p
    p = synth.page_numpy(160, 120, index=6)
    p[:, :35] = 10
    p[:, 95:] = 0
    yield "missing_columns", p
    yield "flat", np.full((90, 70), 100, np.uint8)
    yield "all_dark", np.full((90, 80), 30, np.uint8)                # map not made -> copy
    yield "too_small_map", synth.page_numpy(60, 49, index=1)         # 4 x 4 tiles -> copy
    yield "no_complete_tile", synth.page_numpy(14, 9, index=1)
    yield "exact_multiple", synth.page_numpy(150, 100, index=2)      # no incomplete tile row / column
    yield "one_px", np.array([[200]], np.uint8)
    yield "wide", synth.text_page_numpy(80, 1300, 4, shading=0.3)    # several 250-column workgroups per tile row
    yield "minimal", synth.page_numpy(75, 50, index=3)               # exactly 5 x 5 tiles


@pytest.mark.parametrize("name,page", list(_gray_cases()), ids=[n for n, _ in _gray_cases()])
def test_bgnorm_gray_matches_oracle(prl, oracle, cuda_device, name, page):
    import torch

    got = prl.backgroundNormalization(torch.from_numpy(page).to(cuda_device)).cpu().numpy()
    want = oracle.bgnorm(page)
    assert got.shape == want.shape and np.array_equal(got, want), f"{name}: {(got != want).sum()} bytes differ"


@pytest.mark.parametrize("c", [3, 4])
def test_bgnorm_colour_batch_matches_oracle(prl, oracle, cuda_device, c):
    import torch

    rng = np.random.default_rng(c)
    pages = []
    for i in range(3):
        g = synth.text_page_numpy(211, 167, 10 + i, skew_deg=i, shading=0.5)
        col = np.stack([g, np.clip(g.astype(int) + 10, 0, 255).astype(np.uint8), g // 2 + 100] +
                       ([rng.integers(0, 256, g.shape, dtype=np.uint8)] if c == 4 else []), axis=-1)
        pages.append(col)
    pages.append(rng.integers(0, 256, pages[0].shape, dtype=np.uint8))   # noise: many tiles below mincount
    batch = np.stack(pages)
    got = prl.backgroundNormalization(torch.from_numpy(batch).to(cuda_device)).cpu().numpy()
    assert got.shape == batch.shape[:3] + (3,)
    for i in range(len(pages)):
        want = oracle.bgnorm(batch[i])
        assert np.array_equal(got[i], want), f"page {i}: {(got[i] != want).sum()} bytes differ"


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_bgnorm_reference_test_images(prl, oracle, cuda_device, path):
    """The reference's own test_data/binarize images (inputs held in tests/golden/*.npz)."""
    import torch

    img = np.load(path)["gray"]
    got = prl.backgroundNormalization(torch.from_numpy(img).to(cuda_device)).cpu().numpy()
    assert np.array_equal(got, oracle.bgnorm(img))


def test_bgnorm_views_and_host_entry(prl, oracle, cuda_device):
    import torch

    # a pitched batch (row step > width) and an output with its own pitch, guard bytes untouched
    pages = np.stack([synth.text_page_numpy(95, 131, i, shading=0.4) for i in range(3)])
    big = torch.full((3, 95, 160), 7, dtype=torch.uint8, device=cuda_device)
    big[:, :, 3:134] = torch.from_numpy(pages).to(cuda_device)
    from prlib_amd import _capi
    L = _capi.lib()
    out = torch.full((3, 95, 144), 9, dtype=torch.uint8, device=cuda_device)
    src = big[:, :, 3:134]
    dst = out[:, :, 5:136]
    _capi.check(L.prl_hip_bgnorm_batch_device(3, 1, src.data_ptr(), src.stride(0), src.stride(1), 131, 95, dst.data_ptr(),
                                              dst.stride(0), dst.stride(1), torch.cuda.current_stream().cuda_stream))
    o = out.cpu().numpy()
    for i in range(3):
        assert np.array_equal(o[i, :, 5:136], oracle.bgnorm(pages[i]))
    o[:, :, 5:136] = 9
    assert (o == 9).all()
    # host entry (what the cv::Mat wrapper calls), gray and colour
    assert np.array_equal(prl.backgroundNormalization(pages[0]), oracle.bgnorm(pages[0]))
    col = np.stack([pages[1], pages[2], pages[0]], axis=-1)
    assert np.array_equal(prl.backgroundNormalization(col), oracle.bgnorm(col))


def test_bgnorm_argument_errors(prl, cuda_device):
    import torch

    from prlib_amd import _capi
    L = _capi.lib()
    t = torch.zeros((10, 10), dtype=torch.uint8, device=cuda_device)
    assert L.prl_hip_bgnorm_batch_device(1, 1, t.data_ptr(), 100, 10, 0, 10, t.data_ptr(), 100, 10, None) == _capi.PRL_ERR_EMPTY
    assert L.prl_hip_bgnorm_batch_device(1, 2, t.data_ptr(), 100, 10, 5, 10, t.data_ptr(), 100, 10, None) == _capi.PRL_ERR_BAD_CHANNELS
    assert L.prl_hip_bgnorm_batch_device(1, 1, t.data_ptr(), 100, 4, 5, 10, t.data_ptr(), 100, 10, None) == _capi.PRL_ERR_BAD_ARG
    assert L.prl_hip_bgnorm_out_channels(1) == 1 and L.prl_hip_bgnorm_out_channels(3) == 3 and L.prl_hip_bgnorm_out_channels(4) == 3
