"""Parity at BASELINE.json's full sizes through size-independent properties (the CPU oracle takes ~1.5 s per
4K page, so only a few pages go through it; the rest is checked against the GPU's own literal pipeline, which is
an independent implementation of the same specification, and through batch/crop invariances)."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SAUVOLA, NIBLACK, WOLFJOLION, NICK, FENG = range(5)


def _digest(t):
    return hashlib.sha256(t.contiguous().cpu().numpy().tobytes()).hexdigest()


def test_headline_batch_fused_equals_literal_and_is_batch_independent(prl, oracle, cuda_device):
    """Sauvola k=0.34 w=31 on 4096x4096 pages (the headline workload, 24 pages here)."""
    import torch
    from prlib_amd import synth

    pages = synth.pages_torch(24, 4096, 4096, cuda_device, seed=5000, pitch=4096)
    p = prl.make_params(SAUVOLA, 31, 0.34, 0)
    fused = prl.binarize(pages, p).clone()
    st = prl.last_stats()
    assert st.literal_pages == 0 and st.pixels == 24 * 4095 * 4095
    prl.set_exec_mode(1)
    try:
        lit = prl.binarize(pages[:6], p).clone()
    finally:
        prl.set_exec_mode(0)
    assert torch.equal(fused[:6], lit)                                   # two independent device paths agree
    alone = prl.binarize(pages[17:18], p)
    assert torch.equal(alone[0], fused[17])                              # a page's mask does not depend on its batch
    dup = torch.cat([pages[3:4], pages[3:4]])
    d = prl.binarize(dup, p)
    assert torch.equal(d[0], d[1]) and torch.equal(d[0], fused[3])
    vals = torch.unique(fused[:2])
    assert set(vals.tolist()) <= {0, 255}
    want = oracle.binarize(pages[0].cpu().numpy(), oracle.make_params(SAUVOLA, 31, 0.34, 0))
    assert np.array_equal(fused[0].cpu().numpy(), want)                  # and the CPU oracle on one full page


@pytest.mark.parametrize("method,win,k,morph", [(NIBLACK, 101, 0.01, 2), (WOLFJOLION, 101, 0.01, 2), (NICK, 21, -0.01, 0),
                                               (FENG, 21, 0.0, 2), (SAUVOLA, 101, 0.01, 2)])
def test_a4_defaults_fused_equals_literal(prl, oracle, cuda_device, method, win, k, morph):
    """BASELINE config 3: A4@300dpi pages (2480 x 3508) at the reference's header defaults."""
    import torch
    from prlib_amd import synth

    pages = synth.pages_torch(6, 3508, 2480, cuda_device, seed=7000 + method, pitch=2560)
    p = prl.make_params(method, win, k, morph)
    fused = prl.binarize(pages, p).clone()
    assert prl.last_stats().literal_pages == 0
    prl.set_exec_mode(1)
    try:
        lit = prl.binarize(pages[:3], p).clone()
    finally:
        prl.set_exec_mode(0)
    assert torch.equal(fused[:3], lit)
    want = oracle.binarize(pages[5].cpu().numpy(), oracle.make_params(method, win, k, morph))
    assert np.array_equal(fused[5].cpu().numpy(), want)


def test_crop_invariance_of_the_interior(prl, cuda_device):
    """A pixel's mask value depends only on its window: the interior of a crop equals the interior of the page."""
    import torch
    from prlib_amd import synth

    page = synth.pages_torch(1, 2048, 3000, cuda_device, seed=9000)[0]
    p = prl.make_params(SAUVOLA, 31, 0.34, 0)
    full = prl.binarize(page, p)
    y0, x0, hh, ww = 500, 777, 600, 901
    crop = prl.binarize(page[y0:y0 + hh, x0:x0 + ww].contiguous(), p)
    m = 16  # > w/2: away from the crop's replicate padding
    assert torch.equal(crop[m:hh - 1 - m, m:ww - 1 - m], full[y0 + m:y0 + hh - 1 - m, x0 + m:x0 + ww - 1 - m])


def test_nlm_4k_interior_equals_oracle_on_crop(prl, oracle, cuda_device):
    """BASELINE config 4 size: NL-means is local (27x27 support), so a crop with a 13-pixel margin reproduces it."""
    import torch
    from prlib_amd import synth

    gray = synth.pages_torch(1, 4096, 4096, cuda_device, seed=11000)[0]
    gen = torch.Generator(device=cuda_device)
    gen.manual_seed(3)
    img = (gray[:, :, None].float() + torch.randn((4096, 4096, 3), device=cuda_device, generator=gen) * 15).round().clamp(0, 255).to(torch.uint8)
    den = prl.denoise(img, 10.0)
    for (y0, x0) in [(0, 0), (2000, 1500), (4096 - 160, 4096 - 200)]:
        hh, ww = 160, 200
        sub = img[y0:y0 + hh, x0:x0 + ww].cpu().numpy().copy()
        want = oracle.denoise(sub, 10.0, threads=8)
        got = den[y0:y0 + hh, x0:x0 + ww].cpu().numpy()
        ys = slice(0 if y0 == 0 else 13, hh if y0 + hh == 4096 else hh - 13)
        xs = slice(0 if x0 == 0 else 13, ww if x0 + ww == 4096 else ww - 13)
        assert np.array_equal(got[ys, xs], want[ys, xs])
    flat = torch.full((300, 300, 3), 77, dtype=torch.uint8, device=cuda_device)
    assert torch.equal(prl.denoise(flat, 10.0), flat)                   # constant images are fixed points


def test_config5_full_size_a4_chain(prl, oracle, cuda_device):
    """BASELINE config 5 at its page size: deskew -> NL-means -> backgroundNormalization -> Sauvola -> Zhang-Suen on A4@300dpi
    colour scans (6 pages through the device chain, one of them through the composed oracle - host NL-means on a 3508^2 page
    takes tens of seconds), plus batch independence: a page's result does not depend on its neighbours."""
    import torch
    from prlib_amd import synth

    pages, skews = synth.text_pages_torch(6, 3508, 2480, cuda_device, seed=7100, channels=3)
    outs, angles = prl.process_pages(pages, 3, prl.SAUVOLA, 31, 0.34, 0, denoise_strength=10.0, thin=0, deskew=True,
                                     background_normalization=True)
    assert sum(1 for o in outs if o.shape[0] == o.shape[1]) >= 4          # most pages are skewed -> square canvases
    assert float(np.abs(angles - skews).max()) < 1.0                       # the vote recovers the drawn skew
    alone, a1 = prl.process_pages(pages[4:5].contiguous(), 3, prl.SAUVOLA, 31, 0.34, 0, denoise_strength=10.0, thin=0, deskew=True,
                                  background_normalization=True)
    assert a1[0] == angles[4] and torch.equal(alone[0], outs[4])
    i = 1
    cur, info = oracle.deskew(np.ascontiguousarray(pages[i].cpu().numpy()))
    cur = oracle.denoise(np.ascontiguousarray(cur), 10.0, threads=64)
    cur = oracle.bgr2gray(np.ascontiguousarray(oracle.bgnorm(np.ascontiguousarray(cur))))
    mask = oracle.binarize(np.ascontiguousarray(cur), oracle.make_params(SAUVOLA, 31, 0.34, 0))
    want = oracle.thin(255 - mask, 0)
    assert info["angle"] == angles[i]
    got = outs[i].cpu().numpy()
    assert got.shape == want.shape and np.array_equal(got, want)


def test_round2_stages_on_the_reference_test_images(prl, oracle, cuda_device):
    """deskew and backgroundNormalization on the reference's own test_data/binarize images (inputs in tests/golden/*.npz)."""
    import glob
    import os

    import torch

    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    n = 0
    for path in sorted(glob.glob(os.path.join(gdir, "0*.npz"))):
        gray = np.load(path)["gray"]
        t = torch.from_numpy(gray).to(cuda_device)
        outs, angles = prl.deskew(t[None])
        want, info = oracle.deskew(gray)
        assert angles[0] == info["angle"] and np.array_equal(outs[0].cpu().numpy(), want), path
        assert np.array_equal(prl.backgroundNormalization(t).cpu().numpy(), oracle.bgnorm(gray)), path
        n += 1
    assert n >= 6


@pytest.mark.parametrize("method,win,k,morph", [(SAUVOLA, 31, 0.34, 0), (SAUVOLA, 101, 0.01, 2), (WOLFJOLION, 101, 0.01, 2), (FENG, 21, 0.0, 2)])
def test_real_scans_tiled_to_4k_pages_fused_equals_literal(prl, oracle, cuda_device, method, win, k, morph):
    """The reference's own scans (tests/golden/scans) tiled to 4096 x 4096 pages - tools/bench_real.py's batch, 12 pages of it:
    real paper queues two orders of magnitude more near-threshold pixels than the synthetic pages, repeats its deviation maximum
    at every repetition of a tile (hundreds of Wolf-Jolion candidates) and holds Feng's exact ties; fused == literal on every
    page, the CPU oracle on one."""
    import glob
    import os

    import torch

    scans = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scans", "*.npz")))
    assert len(scans) >= 24
    pages_np = []
    for i in range(12):
        g = np.load(scans[(5 * i) % len(scans)])["gray"]
        big = np.tile(g, (-(-4096 // g.shape[0]) + 1, -(-4096 // g.shape[1]) + 1))
        oy, ox = (i * 97) % g.shape[0], (i * 211) % g.shape[1]
        pages_np.append(np.ascontiguousarray(big[oy:oy + 4096, ox:ox + 4096]))
    pages = torch.from_numpy(np.stack(pages_np)).to(cuda_device)
    p = prl.make_params(method, win, k, morph)
    fused = prl.binarize(pages, p).clone()
    st = prl.last_stats()
    assert st.literal_pages == 0
    prl.set_exec_mode(1)
    try:
        lit = prl.binarize(pages, p).clone()
    finally:
        prl.set_exec_mode(0)
    assert torch.equal(fused, lit)
    want = oracle.binarize(pages_np[7], oracle.make_params(method, win, k, morph))
    assert np.array_equal(fused[7].cpu().numpy(), want)
