"""BASELINE.json configs 3, 4 and 5 AT THEIR STATED BATCH SIZES ("next" item 5): 256 A4 pages through Niblack /
Wolf-Jolion / NICK, 64 x 4096^2 x 3 through prl::denoise, a 256-page A4 colour batch through the five-stage chain.  Per config at
least three spread-out pages go through the CPU oracle; every other page is checked through an independent device path (the
literal pipeline for the binarizers, the stages called one by one for the chain) or through batch-independence properties."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SAUVOLA, NIBLACK, WOLFJOLION, NICK, FENG = range(5)
CORES = os.cpu_count() or 1


def _spread(n, k):
    return sorted({int(round(i * (n - 1) / max(1, k - 1))) for i in range(k)})


@pytest.mark.parametrize("method,win,k,morph", [(NIBLACK, 101, 0.01, 2), (WOLFJOLION, 101, 0.01, 2), (NICK, 21, -0.01, 0)])
def test_config3_256_a4_pages(prl, oracle, cuda_device, method, win, k, morph):
    """256 x 2480x3508 pages at the reference's header defaults: fused == literal pipeline on every page, == oracle on 3."""
    import torch
    from prlib_amd import synth

    n = 256
    pages = synth.pages_torch(n, 3508, 2480, cuda_device, seed=8000 + method, pitch=2560)
    p = prl.make_params(method, win, k, morph)
    fused = prl.binarize(pages, p).clone()
    st = prl.last_stats()
    assert st.literal_pages == 0 and st.pixels == n * fused.shape[1] * fused.shape[2]
    prl.set_exec_mode(1)
    try:
        for a in range(0, n, 64):     # (the literal pipeline keeps two float64 integral planes per page: in chunks)
            lit = prl.binarize(pages[a:a + 64], p)
            assert torch.equal(fused[a:a + 64], lit), f"pages {a}..{a + 63}: fused and literal pipelines differ"
            del lit
    finally:
        prl.set_exec_mode(0)
    idx = _spread(n, 3)
    host = pages[idx].cpu().numpy()[:, :, :2480].copy()
    want = oracle.binarize_batch(host, oracle.make_params(method, win, k, morph), threads=CORES)
    got = fused[idx].cpu().numpy()
    for j, i in enumerate(idx):
        assert np.array_equal(got[j], want[j]), f"page {i}: {int((got[j] != want[j]).sum())} pixels differ from the oracle"


def test_config4_64_noisy_4k_scans(prl, oracle, cuda_device):
    """prl::denoise(10) on 64 x 4096^2 x 3: 256^2 crops of 3 spread-out pages against the oracle (NL-means is local: 13-pixel
    margin), duplicated pages give duplicated results, a page denoised alone equals the page denoised in the batch."""
    import torch
    from prlib_amd import synth

    n = 64
    gray = synth.pages_torch(8, 4096, 4096, cuda_device, seed=12000)
    gen = torch.Generator(device=cuda_device)
    gen.manual_seed(5)
    img = torch.empty((n, 4096, 4096, 3), dtype=torch.uint8, device=cuda_device)
    for i in range(n):   # 8 base pages, fresh noise per scan except the last one, which repeats scan 5 exactly
        noise = torch.randn((4096, 4096, 3), device=cuda_device, generator=gen) * 15.0
        img[i] = (gray[i % 8][:, :, None].float() + noise).round_().clamp_(0, 255).to(torch.uint8)
    img[n - 1] = img[5]
    out = prl.denoise(img, 10.0)
    assert torch.equal(out[n - 1], out[5])
    alone = prl.denoise(img[40:41].contiguous(), 10.0)
    assert torch.equal(alone[0], out[40])
    for i, (y0, x0) in zip(_spread(n, 3), [(0, 0), (1900, 2100), (4096 - 256, 4096 - 256)]):
        sub = img[i, y0:y0 + 256, x0:x0 + 256].cpu().numpy().copy()
        want = oracle.denoise(sub, 10.0, threads=CORES)
        got = out[i, y0:y0 + 256, x0:x0 + 256].cpu().numpy()
        ys = slice(0 if y0 == 0 else 13, 256 if y0 + 256 == 4096 else 256 - 13)
        xs = slice(0 if x0 == 0 else 13, 256 if x0 + 256 == 4096 else 256 - 13)
        assert np.array_equal(got[ys, xs], want[ys, xs]), f"scan {i}"


def test_config5_chain_on_256_a4_colour_scans(prl, oracle, cuda_device):
    """deskew -> NL-means -> backgroundNormalization -> Sauvola -> Zhang-Suen on 256 A4@300dpi colour scans in one call (several
    passes, the angle search of a pass beside the NL-means kernels of the one before): 3 spread-out pages against the composed
    oracle, 24 more against the same stages called one by one through the public entry points (no chain glue, no overlap)."""
    import torch
    from prlib_amd import synth

    n = 256
    pages, skews = synth.text_pages_torch(n, 3508, 2480, cuda_device, seed=7300, channels=3)
    outs, angles = prl.process_pages(pages, 3, prl.SAUVOLA, 31, 0.34, 0, denoise_strength=10.0, thin=0, deskew=True,
                                     background_normalization=True)
    assert len(outs) == n and float(np.abs(angles - skews).max()) < 1.0
    # the stages one by one on a spread-out subset
    idx = _spread(n, 24)
    sub = pages[idx].contiguous()
    d_outs, d_ang = prl.deskew(sub)
    for j, i in enumerate(idx):
        assert d_ang[j] == angles[i]
        cur = prl.denoise(d_outs[j].contiguous(), 10.0)
        cur = prl.cvtColorBGR2GRAY(prl.backgroundNormalization(cur))
        mask = prl.binarizeSauvola(cur, 31, 0.34, 0)
        sk = prl.thinZhangSuen(prl.bitwise_not(mask))
        assert sk.shape == outs[i].shape and torch.equal(sk, outs[i]), f"page {i}: chain and staged results differ"
    for i in _spread(n, 3):
        cur, info = oracle.deskew(np.ascontiguousarray(pages[i].cpu().numpy()))
        cur = oracle.denoise(np.ascontiguousarray(cur), 10.0, threads=CORES)
        cur = oracle.bgr2gray(np.ascontiguousarray(oracle.bgnorm(np.ascontiguousarray(cur))))
        mask = oracle.binarize(np.ascontiguousarray(cur), oracle.make_params(SAUVOLA, 31, 0.34, 0))
        want = oracle.thin(255 - mask, 0)
        got = outs[i].cpu().numpy()
        assert info["angle"] == angles[i]
        assert got.shape == want.shape and np.array_equal(got, want), f"page {i} differs from the composed oracle"
