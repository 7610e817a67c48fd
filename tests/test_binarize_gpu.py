"""GPU parity: libprlib_hip.so (through the C ABI) against the CPU oracle, bit for bit.

Integer/byte output => the bar is 0 mismatching pixels.  Cases follow SURVEY.md §4/§8c: seeded
synthetic pages, odd/even sizes, ragged (non-multiple-of-tile) sizes, window clamping, all five
methods, morphology in both directions, flat/black/white/binary pages, and both execution modes.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SAUVOLA, NIBLACK, WOLFJOLION, NICK, FENG = range(5)


def _oracle_batch(oracle, pages_np, method, win, k, morph, **feng):
    p = oracle.make_params(method, win, k, morph, **feng)
    return [oracle.binarize(pg, p) for pg in pages_np]


def _check(prl, oracle, dev, pages_np, method, win, k, morph, feng=None, mode=None):
    import torch

    feng = feng or {}
    pages = torch.from_numpy(np.stack(pages_np)).to(dev)
    if mode is not None:
        prl.set_exec_mode(mode)
    try:
        got = prl.binarize(pages, prl.make_params(method, win, k, morph, **feng)).cpu().numpy()
    finally:
        if mode is not None:
            prl.set_exec_mode(0)
    want = _oracle_batch(oracle, pages_np, method, win, k, morph, **feng)
    for i, wnt in enumerate(want):
        assert got[i].shape == wnt.shape
        bad = int((got[i] != wnt).sum())
        assert bad == 0, f"page {i}: {bad} mismatching pixels (method {method}, w {win}, k {k}, morph {morph})"
    return prl.last_stats()


def _pages(shape, kinds, seed=0):
    from prlib_amd import synth

    rng = np.random.default_rng(seed)
    h, w = shape
    out = []
    for i, kind in enumerate(kinds):
        if kind == "doc":
            out.append(synth.page_numpy(h, w, index=seed * 100 + i))
        elif kind == "noise":
            out.append(rng.integers(0, 256, (h, w), dtype=np.uint8))
        elif kind == "binary":
            out.append((rng.integers(0, 2, (h, w)) * 255).astype(np.uint8))
        elif kind == "flat":
            out.append(np.full((h, w), int(rng.integers(1, 256)), np.uint8))
        elif kind == "black":
            out.append(np.zeros((h, w), np.uint8))
        elif kind == "white":
            out.append(np.full((h, w), 255, np.uint8))
        elif kind == "ramp":
            out.append((np.add.outer(np.arange(h), np.arange(w)) % 256).astype(np.uint8))
        elif kind == "dark_corner":  # bright page, black bottom-right: small window sums, large absolute integrals
            a = np.full((h, w), 250, np.uint8)
            a[h // 2:, w // 2:] = rng.integers(0, 3, (h - h // 2, w - w // 2), dtype=np.uint8)
            out.append(a)
    return out


@pytest.mark.parametrize("mode", [0, 1], ids=["auto", "literal"])
@pytest.mark.parametrize("method,win,k", [
    (SAUVOLA, 31, 0.34), (SAUVOLA, 15, 0.34), (SAUVOLA, 101, 0.01),
    (NIBLACK, 31, 0.01), (NIBLACK, 15, -0.2),
    (WOLFJOLION, 31, 0.3), (WOLFJOLION, 15, 0.01),
    (NICK, 21, -0.01), (NICK, 31, -0.1),
    (FENG, 21, 0.0), (FENG, 31, 0.0),
])
def test_methods_on_mixed_pages(prl, oracle, cuda_device, method, win, k, mode):
    kinds = ["doc", "noise", "binary", "flat", "black", "white", "ramp", "dark_corner"]
    _check(prl, oracle, cuda_device, _pages((203, 331), kinds, seed=method + 1), method, win, k, 0, mode=mode)


@pytest.mark.parametrize("shape", [(64, 64), (65, 130), (129, 67), (40, 512), (300, 33), (257, 1031)])
@pytest.mark.parametrize("method", [SAUVOLA, NIBLACK, WOLFJOLION, NICK, FENG])
def test_ragged_sizes(prl, oracle, cuda_device, shape, method):
    win = 15 if min(shape) > 15 else 7
    _check(prl, oracle, cuda_device, _pages(shape, ["doc", "noise"], seed=3), method, win, 0.2, 0)


@pytest.mark.parametrize("morph", [1, 2, -1, -2, 3])
@pytest.mark.parametrize("method", [SAUVOLA, NICK])
def test_morphology(prl, oracle, cuda_device, method, morph):
    _check(prl, oracle, cuda_device, _pages((150, 211), ["doc", "noise", "binary"], seed=5), method, 15, 0.2, morph)


def test_window_clamped_to_page(prl, oracle, cuda_device):
    # windowSize=101 on a 60x80 page: w = 60 (even), output = page size
    st_pages = _pages((60, 80), ["doc", "noise"], seed=7)
    _check(prl, oracle, cuda_device, st_pages, SAUVOLA, 101, 0.2, 0)
    _check(prl, oracle, cuda_device, st_pages, NIBLACK, 101, 0.2, 2)


def test_reference_defaults(prl, oracle, cuda_device):
    # header defaults: Sauvola/Niblack/Wolf (101, 0.01, 2), NICK (21, -0.01, 0), Feng (21, ..., 2)
    pages = _pages((260, 300), ["doc", "noise"], seed=11)
    import torch

    dev_pages = torch.from_numpy(np.stack(pages)).to(cuda_device)
    for fn, method in [(prl.binarizeSauvola, SAUVOLA), (prl.binarizeNiblack, NIBLACK),
                       (prl.binarizeWolfJolion, WOLFJOLION), (prl.binarizeNICK, NICK), (prl.binarizeFeng, FENG)]:
        got = fn(dev_pages).cpu().numpy()
        p = oracle.make_params(method)
        for i, pg in enumerate(pages):
            assert np.array_equal(got[i], oracle.binarize(pg, p)), f"defaults, method {method}, page {i}"


def test_feng_parameters(prl, oracle, cuda_device):
    pages = _pages((120, 140), ["doc", "flat", "black"], seed=13)
    for feng in [dict(alpha1=0.12, k1=0.25, k2=0.04, gamma=2.0), dict(alpha1=0.5, k1=0.1, k2=0.01, gamma=3.0)]:
        _check(prl, oracle, cuda_device, pages, FENG, 21, 0.0, 0, feng=feng)


def test_host_entry_point_and_padded_side_effect(prl, oracle, cuda_device):
    page = _pages((90, 111), ["doc"], seed=17)[0]
    mask, padded = prl.binarize(page, prl.make_params(SAUVOLA, 15, 0.34, 2), return_padded=True)
    assert np.array_equal(mask, oracle.binarize(page, oracle.make_params(SAUVOLA, 15, 0.34, 2)))
    assert np.array_equal(padded, oracle.pad_replicate(page, 7))
    # non-contiguous rows (cv::Mat ROI): step > width
    big = np.zeros((90, 160), np.uint8)
    big[:, :111] = page
    assert np.array_equal(prl.binarize(big[:, :111], prl.make_params(NICK, 21, -0.1, 0)),
                          oracle.binarize(page, oracle.make_params(NICK, 21, -0.1, 0)))


def test_pages_table_entry_point(prl, oracle, cuda_device):
    import ctypes as C

    import torch
    from prlib_amd import _capi

    pages = _pages((100, 120), ["doc", "noise", "ramp"], seed=19)
    dev = [torch.from_numpy(p).to(cuda_device) for p in pages]
    params = prl.make_params(SAUVOLA, 15, 0.3, 0)
    g = prl.geometry(params, 120, 100)
    outs = [torch.zeros((g.out_h, 128), dtype=torch.uint8, device=cuda_device) for _ in pages]
    src_tab = (C.c_void_p * 3)(*[t.data_ptr() for t in dev])
    dst_tab = (C.c_void_p * 3)(*[t.data_ptr() for t in outs])
    _capi.check(_capi.lib().prl_hip_binarize_pages_device(C.byref(params), 3, src_tab, 120, 120, 100, dst_tab, 128,
                                                          torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    p = oracle.make_params(SAUVOLA, 15, 0.3, 0)
    for i, pg in enumerate(pages):
        assert np.array_equal(outs[i][:, :g.out_w].cpu().numpy(), oracle.binarize(pg, p))


@pytest.mark.parametrize("morph", [0, 2, -3, 9])
def test_flagged_pages_are_redone_literally_as_one_batch(prl, oracle, cuda_device, morph):
    """Several pages of a call overflow the fix-up list (flat pages on the tie, each with more undecidable pixels than the list
    holds): the library redoes them through the literal pipeline in ONE batch - page-pointer tables on the device, chunks by the
    scratch budget, the morphology pass behind it (radius 9: the chained large-radius path) - here through the page-table entry
    point, so the flagged pages come from scattered allocations."""
    import ctypes as C

    import torch
    from prlib_amd import _capi

    w, c = 15, 200
    k = _flat_boundary_k(c, w, c)
    h, wd = 400, 360                    # 143 000 undecidable pixels a flat page: more than the 2^17 entries of the list
    pages = _pages((h, wd), ["doc"] * 6, seed=97)
    for i in (1, 2, 4):
        pages[i][:, :] = c
    dev = [torch.from_numpy(p).to(cuda_device) for p in pages]
    params = prl.make_params(SAUVOLA, w, k, morph)
    g = prl.geometry(params, wd, h)
    outs = [torch.full((g.out_h, 384), 7, dtype=torch.uint8, device=cuda_device) for _ in pages]
    src_tab = (C.c_void_p * 6)(*[t.data_ptr() for t in dev])
    dst_tab = (C.c_void_p * 6)(*[t.data_ptr() for t in outs])
    _capi.check(_capi.lib().prl_hip_binarize_pages_device(C.byref(params), 6, src_tab, wd, wd, h, dst_tab, 384,
                                                          torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    st = prl.last_stats()
    assert st.literal_pages == 3
    p = oracle.make_params(SAUVOLA, w, k, morph)
    for i, pg in enumerate(pages):
        assert np.array_equal(outs[i][:, :g.out_w].cpu().numpy(), oracle.binarize(pg, p)), i
        assert (outs[i][:, g.out_w:] == 7).all()


def test_errors_match_reference(prl, cuda_device):
    import torch

    page = torch.zeros((50, 50), dtype=torch.uint8, device=cuda_device)
    with pytest.raises(ValueError):
        prl.binarizeSauvola(page, 30)          # even window
    with pytest.raises(ValueError):
        prl.binarizeSauvola(page, 1)           # window <= 1
    with pytest.raises(ValueError):
        prl.binarizeNICK(torch.zeros((0, 0), dtype=torch.uint8, device=cuda_device))  # empty
    from prlib_amd import _capi

    with pytest.raises(_capi.PrlError) as e:
        prl.binarizeWolfJolion(page, 51)       # W == w after clamping: empty rect (cv::Exception upstream)
    assert e.value.status == _capi.PRL_ERR_EMPTY_RECT


def test_c2_full_page_sauvola_w15(prl, oracle, cuda_device):
    """BASELINE config 2: Sauvola k=0.34 w=15 on one 4096x4096 page, bit-exact vs CPU."""
    from prlib_amd import synth

    page = synth.page_numpy(4096, 4096, index=0)
    st = _check(prl, oracle, cuda_device, [page], SAUVOLA, 15, 0.34, 0)
    assert st.pixels == 4095 * 4095


def _flat_boundary_k(c, w, p):
    """k for which Sauvola's exact-arithmetic threshold on a flat page of value c equals p - 0.5."""
    n = (w - 1) ** 2
    f = 1.0 / (w * w)
    m = f * c * n
    v = f * c * c * n - m * m
    s = v ** 0.5
    return ((p - 0.5) / m - 1.0) / (s / 128.0 - 1.0)


def test_undecidable_pixels_go_through_refine_and_fixup(prl, oracle, cuda_device):
    """Flat page whose threshold sits within ~1e-13 of p - 0.5: every pixel defeats the float32 test
    and the float64 interval test, so the absolute-integral fix-up decides all of them."""
    w, c = 15, 200
    k = _flat_boundary_k(c, w, c)
    page = np.full((96, 120), c, np.uint8)
    st = _check(prl, oracle, cuda_device, [page], SAUVOLA, w, k, 0)
    assert st.exact_pixels > 0.9 * st.pixels
    assert st.literal_pages == 0
    # a few boundary pixels inside an ordinary page: only those take the slow stages
    doc = _pages((128, 160), ["doc"], seed=23)[0]
    doc[40:70, 50:100] = c
    st = _check(prl, oracle, cuda_device, [doc], SAUVOLA, w, k, 0)
    assert 0 < st.exact_pixels + st.refined_pixels < 0.2 * st.pixels


def test_fixup_list_overflow_falls_back_to_literal_pipeline(prl, oracle, cuda_device):
    w, c = 15, 200
    k = _flat_boundary_k(c, w, c)
    pages = [np.full((640, 700), c, np.uint8), _pages((640, 700), ["doc"], seed=29)[0]]
    st = _check(prl, oracle, cuda_device, pages, SAUVOLA, w, k, 0)
    assert st.literal_pages == 1
    st = _check(prl, oracle, cuda_device, pages, SAUVOLA, w, k, 2)  # with morphology after the rerun
    assert st.literal_pages == 1


def test_auto_mode_uses_the_fused_kernel(prl, oracle, cuda_device):
    pages = _pages((300, 600), ["doc", "doc"], seed=31)
    st = _check(prl, oracle, cuda_device, pages, SAUVOLA, 31, 0.34, 0)
    assert st.literal_pages == 0 and st.exact_pixels < 50


def test_golden_fixtures_on_device(prl, cuda_device):
    """The committed fixtures (inputs = the reference's own test images) through the HIP path."""
    import glob
    import os

    import torch

    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    n = 0
    for path in sorted(glob.glob(os.path.join(gdir, "0*.npz"))):
        z = np.load(path)
        gray = torch.from_numpy(z["gray"]).to(cuda_device)
        for key in z.files:
            if not key.startswith("mask_"):
                continue
            _, m, w, k, mo = key.split("_")
            shape = tuple(z["shape_" + key[5:]])
            want = np.unpackbits(z[key], axis=1)[:, :shape[1]].astype(np.uint8) * 255
            got = prl.binarize(gray, prl.make_params(int(m), int(w), float(k), int(mo))).cpu().numpy()
            assert np.array_equal(got, want), (os.path.basename(path), key)
            n += 1
    assert n >= 60


def test_wolf_runs_fused(prl, oracle, cuda_device):
    pages = _pages((400, 700), ["doc", "doc", "noise"], seed=37)
    st = _check(prl, oracle, cuda_device, pages, WOLFJOLION, 31, 0.3, 0)
    assert st.literal_pages == 0
    st = _check(prl, oracle, cuda_device, pages, WOLFJOLION, 101, 0.01, 2)   # header defaults
    assert st.literal_pages == 0
    # degenerate pages (no deviation anywhere, or the same everywhere): every pixel is a maximum candidate.  Round 4: the
    # sweeps only need the EXACT maximum of K (one atomicMax per wavefront), the candidate list may overflow - the page
    # stays on the fast path (round 3 sent it through the literal pipeline)
    st = _check(prl, oracle, cuda_device, [np.zeros((700, 600), np.uint8)], WOLFJOLION, 15, 0.3, 0)
    assert st.literal_pages == 0
    st = _check(prl, oracle, cuda_device, [np.full((700, 600), 7, np.uint8), np.full((700, 600), 255, np.uint8)], WOLFJOLION, 15, 0.3, 2)
    assert st.literal_pages == 0


def _wolf_boundary_page_and_k(oracle, w, c, flat, seed, shape=(200, 260)):
    """A page with a noisy half (it holds the deviation maximum) and a flat square of value c, and the k for which the
    exact-arithmetic Wolf-Jolion threshold on the flat square equals c - 0.5: T = m + (s k / smax - k)(m - Imin)."""
    rng = np.random.default_rng(seed)
    h, wd = shape
    page = rng.integers(0, 256, (h, wd), dtype=np.uint8)
    y0, x0 = 20, wd // 2
    page[y0:y0 + flat + w, x0:x0 + flat + w] = c
    page[0, 0] = 0   # Imin = 0
    m, s = oracle.mean_dev(page, oracle.make_params(WOLFJOLION, w, 0.3, 0))
    smax = np.nanmax(s)
    half = w // 2
    yy, xx = y0 + half + 2, x0 + half + 2   # an output pixel whose window lies inside the flat square
    mf, sf = m[yy, xx], s[yy, xx]
    k = (c - 0.5 - mf) / ((sf / smax - 1.0) * (mf - 0.0))
    return page, k, (yy, xx)


def test_wolf_literal_maximum_is_computed_when_a_pixel_needs_it(prl, oracle, cuda_device):
    """Pixels within ~1e-13 of their threshold defeat the float64 interval test, so the literal fix-up decides them - and for
    Wolf-Jolion that needs the LITERAL devianceMax, which round 4 only computes then (absolute corner sums of sweep B's
    candidates of that page): the lazy path, on the page that needs it, beside pages that do not."""
    w, c = 15, 120
    page, k, _ = _wolf_boundary_page_and_k(oracle, w, c, flat=24, seed=61)
    others = _pages(page.shape, ["doc", "noise"], seed=62)
    for morph in (0, 2):
        st = _check(prl, oracle, cuda_device, [others[0], page, others[1]], WOLFJOLION, w, k, morph)
        assert st.exact_pixels >= 24 * 24 // 2 and st.literal_pages == 0
    # the same with so many such pixels that the fix-up list (2^17 entries) overflows: that page (only) goes through the literal pipeline
    big, kb, _ = _wolf_boundary_page_and_k(oracle, w, c, flat=400, seed=63, shape=(520, 900))
    st = _check(prl, oracle, cuda_device, [big, _pages(big.shape, ["doc"], seed=64)[0]], WOLFJOLION, w, kb, 0)
    assert st.literal_pages == 1


def test_many_undecidable_pixels_take_the_page_major_corner_sums(prl, oracle, cuda_device):
    """From 64 queued pixels on (calls of 8 pages or more) the absolute corner sums are built page by page: k_group_items +
    k_corner_rows instead of one pass over the page per pixel.  Flat regions on the tie (every pixel undecidable) that touch all
    four page borders and corners - the replicate padding's multiplicities - among ordinary pages; a whole flat page."""
    w, c = 15, 200
    k = _flat_boundary_k(c, w, c)
    docs = _pages((230, 310), ["doc"] * 9, seed=71)
    docs[1][:70, :90] = c            # top-left corner
    docs[1][-60:, -80:] = c          # bottom-right corner
    docs[3][:50, -100:] = c          # top-right
    docs[3][-55:, :75] = c           # bottom-left
    docs[3][100:140, 120:200] = c    # interior
    docs[6][:, :] = c                # everything (71 000 pixels, all of them through the literal evaluation)
    for morph in (0, 2):
        st = _check(prl, oracle, cuda_device, docs, SAUVOLA, w, k, morph)
        assert st.exact_pixels > 60000 and st.literal_pages == 0
    # the brute-force kernel on the same pixels (calls below 8 pages): same masks
    st = _check(prl, oracle, cuda_device, [docs[1], docs[3]], SAUVOLA, w, k, 0)
    assert st.exact_pixels > 5000 and st.literal_pages == 0


@pytest.mark.parametrize("method,k", [(SAUVOLA, 0.01), (SAUVOLA, 0.34), (NIBLACK, 0.01), (NIBLACK, -0.2), (NICK, -0.01), (FENG, 0.0)])
@pytest.mark.parametrize("win", [41, 63, 101, 129])
def test_wide_windows_on_pages_with_dark_flats(prl, oracle, cuda_device, method, k, win):
    """Windows of 41..129 columns (the wave-scan form of the horizontal sums, the extended last strip) on pages wide enough for
    interior strips, with dark flats (gray 0, 1, 3, 9), dark noise, a dark band across a strip boundary: where the variance floor
    of the float32 test bites.  (Written for the wide-window float32 loop of round 4 - tools/experiments/
    wide_window_float_threshold_loop.patch, measured slower and not kept; its dark-flat cases stay as parity cases.)"""
    rng = np.random.default_rng(win * 7 + method)
    h, wd = 420, 1400
    pages = _pages((h, wd), ["doc", "doc", "doc", "noise"], seed=90 + win)
    pages[0][150:300, 200:900] = 0                                            # black block
    pages[0][20:120, 950:1350] = 3                                            # dark flat
    pages[1][:, 380:470] = rng.integers(1, 10, (h, 90), dtype=np.uint8)       # dark noise band over the first strip boundary
    pages[1][300:, :] = 1                                                     # darkest non-black flat at the bottom
    pages[2][:60, :] = 9
    pages[2][200:260, 600:1000] = rng.integers(0, 3, (60, 400), dtype=np.uint8)
    for morph in (0, 2):
        st = _check(prl, oracle, cuda_device, pages, method, win, k, morph)
        assert st.literal_pages == 0


@pytest.mark.parametrize("width", [777, 2480, 4096, 4100, 6000, 8192, 8200])
def test_page_major_corner_sums_row_widths(prl, oracle, cuda_device, width):
    """k_corner_rows takes 64 bytes of a row per lane up to 4096 columns, 128 up to 8192 (the lane over the row's end fetches
    dwords and single bytes); wider pages stay with the per-pixel kernel.  Tie regions at both ends of the row."""
    w, c = 15, 200
    k = _flat_boundary_k(c, w, c)
    pages = _pages((70, width), ["doc"] * 8, seed=width)
    pages[2][10:50, :60] = c
    pages[2][5:45, -70:] = c
    pages[7][20:60, width // 2 - 40:width // 2 + 40] = c
    st = _check(prl, oracle, cuda_device, pages, SAUVOLA, w, k, 0)
    assert st.exact_pixels > 1000 and st.literal_pages == 0


def test_feng_rational_ties_stay_on_the_fast_path(prl, oracle, cuda_device):
    """Feng at the header defaults: (1 + (1 - alpha1)) m + c3 = 1.25 S / w^2 meets p - 0.5 EXACTLY for ~5 pixels in 10^6, which
    only the literal evaluation decides.  256 x 4K pages have 21 000 of them: round 3's fix-up list (2^14) overflowed and every page
    went through the literal pipeline (753 ms a step); now the list holds them and the page-major kernel builds their sums."""
    import torch

    from prlib_amd import synth

    pages = synth.pages_torch(24, 1536, 1536, cuda_device, seed=4000)
    p = prl.default_params(FENG)
    got = prl.binarize(pages, p).cpu().numpy()
    st = prl.last_stats()
    assert st.literal_pages == 0 and st.exact_pixels >= 64
    host = pages.cpu().numpy()
    po = oracle.make_params(FENG)
    for i in (0, 7, 23):
        assert np.array_equal(got[i], oracle.binarize(host[i].copy(), po))


def test_wolf_literal_maximum_with_many_candidates_page_major(prl, oracle, cuda_device):
    """A page tiled from a small block attains its deviation maximum at every repetition: > 64 candidates, so the lazily computed
    literal devianceMax takes the page-major kernel too (and the tie pixels of the flat patch the page-major fix-up)."""
    w, c = 15, 120
    rng = np.random.default_rng(81)
    block = rng.integers(0, 256, (32, 32), dtype=np.uint8)
    page = np.tile(block, (12, 16))[:380, :500].copy()
    page[:20, :] = 127
    page[-20:, :] = 127       # a bland frame: the windows on the replicated borders must not win
    page[:, :20] = 127
    page[:, -20:] = 127
    page[40:40 + 40 + w, 260:260 + 40 + w] = c
    page[0, 0] = 0
    m, s = oracle.mean_dev(page, oracle.make_params(WOLFJOLION, w, 0.3, 0))
    assert (s >= np.nanmax(s) * (1 - 1e-9)).sum() >= 100
    yy, xx = 40 + w // 2 + 2, 260 + w // 2 + 2
    k = (c - 0.5 - m[yy, xx]) / ((s[yy, xx] / np.nanmax(s) - 1.0) * m[yy, xx])
    pages = _pages(page.shape, ["doc"] * 8, seed=82)
    pages[5] = page
    st = _check(prl, oracle, cuda_device, pages, WOLFJOLION, w, k, 0)
    assert st.exact_pixels >= 800 and st.wolf_candidates >= 64 and st.literal_pages == 0


def test_back_to_back_calls_reuse_the_self_cleaned_state(prl, oracle, cuda_device):
    """Small batches: the last kernel of a call writes the flags into the pinned slot and re-initialises the per-page globals
    and the counter block, and the next call with the same page count skips k_init_globals.  Every transition of that state
    machine - same count twice, another count, literal mode in between, Wolf-Jolion and Feng (which need imin = 255 and the
    maxima zeroed), pages that reach the literal fix-up (the epilogue then waits for every workgroup) - against the oracle."""
    docs3 = _pages((300, 520), ["doc", "doc", "noise"], seed=51)
    docs5 = _pages((300, 520), ["doc", "noise", "doc", "binary", "doc"], seed=52)
    w, c = 15, 200
    ties = [np.full((260, 300), c, np.uint8), _pages((260, 300), ["doc"], seed=53)[0], np.full((260, 300), c, np.uint8)]
    k_tie = _flat_boundary_k(c, w, c)
    seq = [(docs3, SAUVOLA, 31, 0.34, 0, None), (docs3, SAUVOLA, 31, 0.34, 0, None), (docs5, NICK, 21, -0.1, 0, None),
           (docs3, WOLFJOLION, 31, 0.3, 0, None), (docs3, WOLFJOLION, 31, 0.3, 0, None), (docs3, SAUVOLA, 15, 0.2, 2, 1),
           (docs3, FENG, 31, 0.2, 0, None), (docs3, FENG, 31, 0.2, 0, None), (ties, SAUVOLA, w, k_tie, 0, None),
           (ties, SAUVOLA, w, k_tie, 0, None), (docs3, NIBLACK, 31, -0.2, 1, None), (docs3, WOLFJOLION, 101, 0.01, 2, None)]
    for pages, method, win, k, morph, mode in seq:
        _check(prl, oracle, cuda_device, pages, method, win, k, morph, mode=mode)


@pytest.mark.parametrize("n", [1, 2, -2, 5, 8, -8])
def test_public_morph_entry_on_gray_and_binary(prl, oracle, cuda_device, n):
    import torch

    rng = np.random.default_rng(abs(n))
    gray = rng.integers(0, 256, (77, 203), dtype=np.uint8)          # cv::dilate/erode semantics on any u8 image
    got = prl.morph(torch.from_numpy(gray).to(cuda_device), n).cpu().numpy()
    assert np.array_equal(got, oracle.morph(gray, n))


@pytest.mark.parametrize("shape", [(33, 130), (64, 128), (200, 517), (31, 40)])
@pytest.mark.parametrize("morph", [1, -1, 2, 4, -3, 8])
def test_binary_morphology_tiles(prl, oracle, cuda_device, shape, morph):
    # page sizes around the 128x32 tile of the binary kernel; window clamps when the page is small
    pages = _pages(shape, ["binary", "doc"], seed=41)
    _check(prl, oracle, cuda_device, pages, NIBLACK, 15, 0.3, morph)


@pytest.mark.parametrize("n", [9, 12, -17])
def test_morph_radius_above_eight(prl, oracle, cuda_device, n):
    """cv::dilate/erode accept any iteration count; radii above 8 are chained rectangle passes."""
    import torch

    rng = np.random.default_rng(abs(n))
    gray = rng.integers(0, 256, (90, 140), dtype=np.uint8)
    got = prl.morph(torch.from_numpy(gray).to(cuda_device), n).cpu().numpy()
    assert np.array_equal(got, oracle.morph(gray, n))
    _check(prl, oracle, cuda_device, _pages((120, 150), ["doc", "binary"], seed=43), SAUVOLA, 15, 0.3, n)


@pytest.mark.parametrize("method,morph", [(SAUVOLA, 0), (SAUVOLA, 2), (WOLFJOLION, 0), (NICK, -1)])
def test_unaligned_output_pitch(prl, oracle, cuda_device, method, morph):
    """cv::Mat outputs are continuous: the row pitch equals out_w (odd for Sauvola/Niblack) — byte-granular stores."""
    import torch

    pages = _pages((131, 1042), ["doc", "noise"], seed=47)
    dev_pages = torch.from_numpy(np.stack(pages)).to(cuda_device)
    p = prl.make_params(method, 31, 0.3, morph)
    g = prl.geometry(p, 1042, 131)
    out = torch.zeros((2, g.out_h, g.out_w), dtype=torch.uint8, device=cuda_device)   # pitch == out_w
    got = prl.binarize(dev_pages, p, out=out).cpu().numpy()
    po = oracle.make_params(method, 31, 0.3, morph)
    for i, pg in enumerate(pages):
        assert np.array_equal(got[i], oracle.binarize(pg, po))
    # and an input view with a base pointer that is not 8-byte aligned
    big = torch.zeros((2, 131, 1100), dtype=torch.uint8, device=cuda_device)
    big[:, :, 3:1045] = dev_pages
    got2 = prl.binarize(big[:, :, 3:1045], p).cpu().numpy()
    assert np.array_equal(got2, got)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("PRL_SWEEP_SEEDS", "24"))))
def test_random_sweep(prl, oracle, cuda_device, seed):
    """Seeded random configurations: shape, method, window, k, morphology radius (bit-plane and byte hand-off paths),
    page kinds, and an output view with a row step larger than the row."""
    import torch

    rng = np.random.default_rng(1000 + seed)
    method = int(rng.integers(0, 5))
    h, w = int(rng.integers(40, 330)), int(rng.integers(40, 700))
    win = int(rng.choice([3, 5, 7, 9, 15, 21, 31, 33, 35, 41, 51, 101]))
    if method in (WOLFJOLION, NICK, FENG):  # output (H-w) x (W-w) must not be empty
        win = min(win, 2 * ((min(h, w) - 2) // 2) - 1)
    k = float(rng.choice([0.34, 0.2, 0.01, -0.01, -0.2, 0.5]))
    morph = int(rng.choice([0, 0, 1, 2, 2, -1, -2, 3, 4, -4, 5, 8]))
    kinds = [str(x) for x in rng.choice(["doc", "noise", "binary", "flat", "ramp", "dark_corner", "white", "black"], 3)]
    pages = _pages((h, w), kinds, seed=seed + 50)
    params = prl.make_params(method, win, k, morph)
    g = prl.geometry(params, w, h)
    dev_pages = torch.from_numpy(np.stack(pages)).to(cuda_device)
    pad = int(rng.integers(0, 9))
    buf = torch.full((3, g.out_h, g.out_w + pad), 77, dtype=torch.uint8, device=cuda_device)
    out = buf[:, :, :g.out_w]
    prl.binarize(dev_pages, params, out=out)
    got = buf.cpu().numpy()
    p = oracle.make_params(method, win, k, morph)
    for i, pg in enumerate(pages):
        want = oracle.binarize(pg, p)
        bad = int((got[i, :, :g.out_w] != want).sum())
        assert bad == 0, f"seed {seed} page {i} ({kinds[i]}): {bad} mismatches, method {method} w {win} k {k} morph {morph} {h}x{w}"
    assert (got[:, :, g.out_w:] == 77).all()


@pytest.mark.parametrize("method,win", [(SAUVOLA, 183), (NIBLACK, 201), (WOLFJOLION, 185), (NICK, 257), (FENG, 255), (SAUVOLA, 181)])
def test_wide_windows_stay_on_the_fused_path(prl, oracle, cuda_device, method, win):
    """Windows of 183..257 columns: S can exceed 2^23, so the kernel variant without the mantissa trick runs."""
    pages = _pages((300, 340), ["doc", "noise", "white"], seed=win)
    st = _check(prl, oracle, cuda_device, pages, method, win, 0.2 if method != NICK else -0.1, 0)
    assert st.literal_pages == 0 or method == WOLFJOLION   # (a flat page makes every pixel a Wolf candidate)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("PRL_SWEEP_SEEDS_WIDE", "16"))))
def test_random_sweep_wide_pages(prl, oracle, cuda_device, seed):
    """Pages wide enough for interior strips (>= 1100 columns), windows up to 31: those strips run the float32
    pipeline (typed buffer loads, float column sums; k_refine rebuilds the window sums of queued pixels from the page).
    Hard page kinds on purpose: binary and noise pages drive the sums of squares to their largest values, where the
    float32 lane sums round."""
    import torch

    rng = np.random.default_rng(7000 + seed)
    method = int(rng.integers(0, 5))
    h, w = int(rng.integers(50, 260)), int(rng.integers(1100, 2300))
    # (windows above 31 keep the interior strips on the integer pipeline: covered here as well)
    win = int(rng.choice([3, 5, 9, 15, 17, 23, 25, 29, 31, 31, 33, 41, 51, 101]))
    if method in (WOLFJOLION, NICK, FENG):  # output (H-w) x (W-w) must not be empty
        win = min(win, 2 * ((min(h, w) - 2) // 2) - 1)
    k = float(rng.choice([0.34, 0.2, 0.01, -0.01, -0.2, 0.5]))
    morph = int(rng.choice([0, 0, 0, 2, -1, 4]))
    kinds = [str(x) for x in rng.choice(["doc", "noise", "binary", "flat", "ramp", "dark_corner", "white", "black"], 3)]
    pages = _pages((h, w), kinds, seed=seed + 900)
    params = prl.make_params(method, win, k, morph)
    g = prl.geometry(params, w, h)
    dev_pages = torch.from_numpy(np.stack(pages)).to(cuda_device)
    # an input view whose rows do not start on a multiple of 4 bytes (typed loads at every alignment)
    off = int(rng.integers(0, 4))
    big = torch.zeros((3, h, w + 8), dtype=torch.uint8, device=cuda_device)
    big[:, :, off:off + w] = dev_pages
    got = prl.binarize(big[:, :, off:off + w], params).cpu().numpy()
    p = oracle.make_params(method, win, k, morph)
    for i, pg in enumerate(pages):
        want = oracle.binarize(pg, p)
        bad = int((got[i] != want).sum())
        assert bad == 0, f"seed {seed} page {i} ({kinds[i]}): {bad} mismatches, method {method} w {win} k {k} morph {morph} {h}x{w}"


@pytest.mark.parametrize("method", [SAUVOLA, NIBLACK, WOLFJOLION, NICK, FENG])
@pytest.mark.parametrize("win", [23, 31])
def test_float_pipeline_at_the_largest_sums(prl, oracle, cuda_device, method, win):
    """Interior strips, saturated content: all-white and 0/255 patterns push the sums of squares of the lane chain past
    2^24, where the float32 pipeline rounds (cq); black/white borders put nearly-black windows next
    to saturated columns (the case that fixes Qmin)."""
    h, w = 96, 1400
    rng = np.random.default_rng(win * 10 + method)
    white = np.full((h, w), 255, np.uint8)
    checker = (((np.add.outer(np.arange(h), np.arange(w)) // 3) % 2) * 255).astype(np.uint8)
    halves = np.zeros((h, w), np.uint8)
    halves[:, : w // 2 + 5] = 255
    halves[::7, w // 2 - 40: w // 2 + 60] = rng.integers(0, 256, halves[::7, w // 2 - 40: w // 2 + 60].shape, dtype=np.uint8)
    stripes = np.where((np.arange(w) // 37) % 2 == 0, 255, rng.integers(0, 4, w)).astype(np.uint8)[None, :].repeat(h, 0)
    stripes = (stripes.astype(np.int16) - rng.integers(0, 3, (h, w))).clip(0, 255).astype(np.uint8)
    k = {SAUVOLA: 0.34, NIBLACK: -0.2, WOLFJOLION: 0.3, NICK: -0.1, FENG: 0.2}[method]
    _check(prl, oracle, cuda_device, [white, checker, halves, stripes], method, win, k, 0)


def test_wolf_flat_pages_do_not_take_the_candidate_list_from_their_neighbours(prl, oracle, cuda_device):
    """Wolf-Jolion sweep B (round 5): on a flat or blank page every pixel is a candidate for the variance maximum.  The
    candidate list is taken per wavefront-row with one atomic, a page may hold a quarter of it in a batch, and a page past its
    share only marks itself (cand_overflow) - the ordinary pages of the same call keep their candidates and stay on the fast
    path.  Two big flat pages around four document pages, several calls in a row on one stream: every mask equals the oracle's,
    no document page falls to the literal pipeline."""
    import torch

    shape = (900, 1300)
    docs = _pages(shape, ["doc", "doc", "doc", "doc"], seed=77)
    batch = [np.full(shape, 137, np.uint8)] + docs[:2] + [np.zeros(shape, np.uint8)] + docs[2:] + [np.full(shape, 255, np.uint8)]
    t = torch.from_numpy(np.stack(batch)).to(cuda_device)
    for win, k, morph in ((31, 0.3, 0), (101, 0.01, 2), (15, 0.5, 0)):
        want = _oracle_batch(oracle, batch, WOLFJOLION, win, k, morph)
        for rep in range(2):
            got = prl.binarize(t, prl.make_params(WOLFJOLION, win, k, morph)).cpu().numpy()
            st = prl.last_stats()
            for i in range(len(batch)):
                assert np.array_equal(got[i], want[i]), (win, k, morph, rep, i, int((got[i] != want[i]).sum()))
            assert st.literal_pages <= 3, (win, st.literal_pages)          # at most the three flat pages, never a document
    # a single flat page has the whole list: still exact
    one = torch.from_numpy(batch[0]).to(cuda_device)
    assert np.array_equal(prl.binarize(one, prl.make_params(WOLFJOLION, 31, 0.3, 0)).cpu().numpy(),
                          oracle.binarize(batch[0], oracle.make_params(oracle.WOLFJOLION, 31, 0.3, 0)))


def test_wolfjolion_batches_larger_than_the_per_call_wavefront_budget(cuda_device):
    """Wolf-Jolion keeps one maximum per wavefront of a call (2^20 slots); a batch with more wavefronts used
    to be rejected with PRL_ERR_BAD_ARG.  The C ABI now cuts such batches into page chunks (fused_max_pages).  The slot
    count is shrunk to 64 in a child process (PRL_HIP_SEGMAX_CAP is read once per process) so that 12 small pages need
    several chunks."""
    import subprocess
    import sys

    code = r'''
import numpy as np, torch, sys
sys.path.insert(0, %r)
import prlib_amd
prlib_amd._capi.use_library(prlib_amd._capi.HOOKS_LIB_PATH)   # the build that reads the PRL_HIP_* tuning knobs
from oracle import capi as oc
from prlib_amd import synth
pages = np.stack([synth.page_numpy(200, 1300, index=i) for i in range(12)])
p = prlib_amd.make_params(prlib_amd.WOLFJOLION, 31, 0.3, 0)
got = prlib_amd.binarize(torch.from_numpy(pages).cuda(), p).cpu().numpy()
po = oc.make_params(oc.WOLFJOLION, 31, 0.3, 0)
bad = sum(int((got[i] != oc.binarize(pages[i], po)).sum()) for i in range(12))
print("MISMATCH", bad)
''' % ROOT
    env = dict(os.environ, PRL_HIP_SEGMAX_CAP="64")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "MISMATCH 0" in r.stdout, r.stdout + r.stderr


def test_wide_windows_float_rows_kernel_equals_the_integer_loop_and_the_oracle():
    """k_fused_q (windows of 33..129 columns: float window rows, integer horizontal Q sums - VERDICT r5 "next" 7) is the default for
    Sauvola / Niblack / NICK / Feng / Wolf-Jolion's threshold sweep.  A child per setting of PRL_HIP_FUSED_QINT (hooks build: 0 = k_fused's
    integer loop, 1 = interior strips only, 2 = border strips too) binarizes the same pages - documents wide enough for interior
    strips, a ragged width, a page of stripes sitting on their threshold (the queue receives exact sums), byte masks and bit
    planes (morph != 0) - and prints a CRC per call; all three equal each other and the oracle."""
    import subprocess
    import sys

    code = r'''
import numpy as np, torch, sys, zlib
sys.path.insert(0, %r)
import prlib_amd, bench
prlib_amd._capi.use_library(prlib_amd._capi.HOOKS_LIB_PATH)
from oracle import capi as oc
from prlib_amd import synth
rng = np.random.default_rng(5)
docs = [synth.page_numpy(700, 2100, index=3), synth.page_numpy(333, 1501, index=4), synth.text_page_numpy(900, 1300, 2, skew_deg=1.0)]
noise = rng.integers(0, 256, (420, 1777)).astype(np.uint8)
cases = []
for method, w, k in ((0, 101, 0.34), (1, 101, 0.01), (3, 41, -0.1), (2, 101, 0.01), (0, 33, 0.2), (1, 129, 0.2), (0, 131, 0.34), (3, 75, -0.2), (4, 51, 0.2)):
    for morph in (0, 2):
        for img in docs + [noise]:
            cases.append((method, w, k, morph, img))
white = np.full((300, 1500), 255, np.uint8)   # the largest sums the float rows have to hold exactly: 128 x 65025 + 2^23 < 2^24, 64 x 8 x 128 x 255 < 2^24
white[::7, ::5] = 254
for method, w, k in ((1, 129, 0.2), (0, 129, 0.34), (3, 101, -0.1)):
    cases.append((method, w, k, 0, white))
adv = bench.adversarial_stripes(0, 51, 0.34, None)
stripes = np.tile(np.array([adv[0], adv[1]], np.uint8), (600, 800))
cases.append((0, 51, 0.34, 0, stripes))
bad = 0
crc = 0
for method, w, k, morph, img in cases:
    got = prlib_amd.binarize(torch.from_numpy(img).cuda(), prlib_amd.make_params(method, w, k, morph)).cpu().numpy()
    want = oc.binarize(img, oc.make_params(method, w, k, morph))
    bad += int((got != want).sum())
    crc = zlib.crc32(got.tobytes(), crc)
print("MISMATCH", bad, "CRC", crc, "CASES", len(cases))
''' % ROOT
    outs = []
    for q in ("0", "1", "2"):
        env = dict(os.environ, PRL_HIP_FUSED_QINT=q)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("MISMATCH")]
        assert line and line[0].startswith("MISMATCH 0 "), (q, r.stdout + r.stderr)
        outs.append(line[0])
    assert outs[0] == outs[1] == outs[2], outs


def test_deferred_completion_and_two_streams(prl, oracle, cuda_device):
    """prl_hip_set_deferred_completion(1): calls return after enqueuing, the per-page flags (and the literal redo of an
    overflowing page) are handled by later calls / prl_hip_finish.  Two torch streams use separate workspaces."""
    import torch

    w, c = 15, 200
    k = _flat_boundary_k(c, w, c)
    adversarial = [np.full((640, 700), c, np.uint8), _pages((640, 700), ["doc"], seed=29)[0]]   # page 0 overflows the fix-up list
    docs = _pages((300, 600), ["doc", "doc", "doc"], seed=41)
    s1, s2 = torch.cuda.Stream(device=cuda_device), torch.cuda.Stream(device=cuda_device)
    t_adv = torch.from_numpy(np.stack(adversarial)).to(cuda_device)
    t_doc = torch.from_numpy(np.stack(docs)).to(cuda_device)
    torch.cuda.synchronize(cuda_device)
    prl.set_deferred_completion(True)
    try:
        outs = []
        with torch.cuda.stream(s2):
            for method, win, kk in ((SAUVOLA, 31, 0.34), (NICK, 21, -0.1), (WOLFJOLION, 31, 0.3)):
                outs.append((docs, method, win, kk, 0, prl.binarize(t_doc, prl.make_params(method, win, kk, 0))))
        with torch.cuda.stream(s1):
            # six calls in a row on one stream: more than the four flag slots, so earlier calls get resolved on the way
            for morph in (0, 2, 0, -1, 0, 2):
                outs.append((adversarial, SAUVOLA, w, k, morph, prl.binarize(t_adv, prl.make_params(SAUVOLA, w, k, morph))))
        with torch.cuda.stream(s1):
            prl.finish(cuda_device)
            st = prl.last_stats()
        with torch.cuda.stream(s2):
            prl.finish(cuda_device)
        assert st.literal_pages == 1
        for pages, method, win, kk, morph, got in outs:
            want = _oracle_batch(oracle, pages, method, win, kk, morph)
            g = got.cpu().numpy()
            for i in range(len(pages)):
                assert np.array_equal(g[i], want[i]), (method, win, morph, i, int((g[i] != want[i]).sum()))
    finally:
        prl.set_deferred_completion(False)


def test_literal_page_budget_bounds_the_cost_of_hostile_pages(prl, oracle, cuda_device):
    """prl_hip_set_literal_page_budget: a flat page tuned to sit on its own threshold overflows the fix-up
    list and is redone literally.  With a budget of 0 the call reports that instead of paying for it (PRL_ERR_LITERAL_BUDGET,
    the statistics say how many pages needed the redo); the ordinary pages of the same call are complete; with the budget
    lifted (or large enough) the same call is bit-exact again.  Never an approximate mask."""
    import torch
    from prlib_amd import _capi

    w, c = 15, 200
    k = _flat_boundary_k(c, w, c)
    pages = [np.full((640, 700), c, np.uint8), _pages((640, 700), ["doc"], seed=29)[0], _pages((640, 700), ["doc"], seed=30)[0]]
    t = torch.from_numpy(np.stack(pages)).to(cuda_device)
    p = prl.make_params(SAUVOLA, w, k, 0)
    want = _oracle_batch(oracle, pages, SAUVOLA, w, k, 0)
    L = _capi.lib()
    assert L.prl_hip_get_literal_page_budget() == -1
    try:
        prl.set_literal_page_budget(0)
        with pytest.raises(_capi.PrlError) as e:
            prl.binarize(t, p)
        assert e.value.status == _capi.PRL_ERR_LITERAL_BUDGET and "1 of 3 pages" in str(e.value)
        assert prl.last_stats().literal_pages == 1
        got = prl.binarize(t[1:], p).cpu().numpy()            # ordinary pages: no redo needed, the budget does not bite
        assert np.array_equal(got[0], want[1]) and np.array_equal(got[1], want[2])
        prl.set_literal_page_budget(1)
        got = prl.binarize(t, p).cpu().numpy()
        assert all(np.array_equal(got[i], want[i]) for i in range(3)) and prl.last_stats().literal_pages == 1
    finally:
        prl.set_literal_page_budget(-1)
    got = prl.binarize(t, p).cpu().numpy()
    assert all(np.array_equal(got[i], want[i]) for i in range(3))


def test_host_batch_entry_shards_and_double_buffers(prl, oracle, cuda_device):
    """prl_hip_binarize_batch_host on one device: 40 pages of 2048^2 are three chunks alternating between two streams;
    results come back in the caller's order, from pages with a row stride larger than the width."""
    n, h, w = 40, 2048, 2048
    from prlib_amd import synth

    base = [synth.page_numpy(h, w, index=i % 5) for i in range(5)]
    store = np.empty((n, h, w + 64), np.uint8)           # strided pages (cv::Mat ROI-like)
    for i in range(n):
        store[i, :, :w] = np.roll(base[i % 5], i * 37, axis=1)
    pages = [store[i, :, :w] for i in range(n)]
    p = prl.make_params(SAUVOLA, 31, 0.34, 2)
    got = prl.binarize_pages_host(pages, p, n_devices=1)
    po = oracle.make_params(SAUVOLA, 31, 0.34, 2)
    for i in (0, 1, 15, 16, 17, 31, 32, 39):
        want = oracle.binarize(np.ascontiguousarray(pages[i]), po)
        assert np.array_equal(got[i], want), (i, int((got[i] != want).sum()))
    # every page = a rolled copy of one of five bases: cheap whole-batch check through the five distinct results
    assert len({got[i].tobytes() for i in range(n)}) >= 5
    # a short list (fewer pages than a chunk: single buffer) and n_devices = 0 (all visible)
    got2 = prl.binarize_pages_host(pages[:3], prl.make_params(NICK, 21, -0.1, 0), n_devices=0)
    for i in range(3):
        assert np.array_equal(got2[i], oracle.binarize(np.ascontiguousarray(pages[i]), oracle.make_params(NICK, 21, -0.1, 0)))
    with pytest.raises(ValueError):
        prl.binarize_pages_host(pages[:2], prl.make_params(SAUVOLA, 30, 0.34, 0))


def test_host_batch_entry_pinned_pages_and_repeated_calls(prl, oracle, cuda_device):
    """Pages in pinned memory (prl_hip_alloc_host) are moved by DMA directly, no bounce copies: same masks as the pageable
    path and as the oracle; mixed (pinned in / pageable out and the reverse) too.  Repeated calls re-use the per-device
    streams and chunk slots (each call used to leave two streams' workspaces behind): device memory settles."""
    import torch
    from prlib_amd import synth

    n, h, w = 24, 1024, 1536
    base = [synth.page_numpy(h, w, index=i) for i in range(4)]
    p = prl.make_params(SAUVOLA, 31, 0.34, 0)
    po = oracle.make_params(SAUVOLA, 31, 0.34, 0)
    with prl.PinnedPages(n, h, w) as pin_in, prl.PinnedPages(n, h - 1, w - 1) as pin_out:
        for i in range(n):
            pin_in.array[i] = np.roll(base[i % 4], 61 * i, axis=1)
        pageable = [pin_in.array[i].copy() for i in range(n)]
        ref = prl.binarize_pages_host(pageable, p, n_devices=1)                     # bounce path
        got = prl.binarize_pages_host(list(pin_in.array), p, n_devices=1, out=pin_out.array)   # zero-copy both ways
        assert got is pin_out.array and np.array_equal(got, ref)
        got2 = prl.binarize_pages_host(list(pin_in.array), p, n_devices=1)          # pinned in, pageable out
        got3 = prl.binarize_pages_host(pageable, p, n_devices=1, out=pin_out.array)  # pageable in, pinned out
        assert np.array_equal(got2, ref) and np.array_equal(got3, ref)
        for i in (0, 7, n - 1):
            assert np.array_equal(ref[i], oracle.binarize(pageable[i], po))
        # strided pinned pages (a cv::Mat ROI inside a pinned buffer)
        with prl.PinnedPages(4, h, w + 128) as wide:
            views = []
            for i in range(4):
                wide.array[i, :, 64:64 + w] = pageable[i]
                views.append(wide.array[i, :, 64:64 + w])
            assert np.array_equal(prl.binarize_pages_host(views, p, n_devices=1), ref[:4])
        free = []
        for it in range(12):
            prl.binarize_pages_host(pageable[: 8 + (it % 3) * 8], prl.make_params(SAUVOLA, 31, 0.34, 2 * (it % 2)), n_devices=1)
            torch.cuda.synchronize(cuda_device)
            free.append(torch.cuda.mem_get_info(cuda_device)[0] >> 20)
        assert max(free[4:]) - min(free[4:]) < 64, free


def test_single_page_host_entry_with_pinned_buffers(prl, oracle, cuda_device):
    """prl_hip_binarize_host (what the cv::Mat wrapper calls): a dense page / mask in pinned memory is one DMA each, no
    bounce copy; a strided view of pinned memory and pageable memory take the bounce path.  All equal the oracle."""
    from prlib_amd import synth

    h, w = 700, 1100
    page = synth.page_numpy(h, w, index=3)
    for method, win, k, morph in ((SAUVOLA, 31, 0.34, 2), (WOLFJOLION, 21, 0.3, 0)):
        p = prl.make_params(method, win, k, morph)
        want = oracle.binarize(page, oracle.make_params(method, win, k, morph))
        oh, ow = want.shape
        with prl.PinnedPages(1, h, w) as pin, prl.PinnedPages(1, oh, ow) as pout, prl.PinnedPages(1, h, w + 64) as wide:
            pin.array[0] = page
            got = prl.binarize(pin.array[0], p, out=pout.array[0])                      # pinned in, pinned out
            assert np.shares_memory(got, pout.array) and np.array_equal(got, want)
            assert np.array_equal(prl.binarize(pin.array[0], p), want)                   # pinned in, pageable out
            assert np.array_equal(prl.binarize(page, p, out=pout.array[0]), want)        # pageable in, pinned out
            wide.array[0, :, 32:32 + w] = page
            assert np.array_equal(prl.binarize(wide.array[0, :, 32:32 + w], p), want)    # strided pinned view: bounce path
            got, padded = prl.binarize(pin.array[0], p, return_padded=True)
            assert np.array_equal(got, want) and padded.shape == (h + 2 * (min(win, h, w) // 2), w + 2 * (min(win, h, w) // 2))


@pytest.mark.parametrize("method,k", [(SAUVOLA, 0.2), (NIBLACK, -0.2), (NICK, -0.1), (WOLFJOLION, 0.3), (FENG, 0.0)])
@pytest.mark.parametrize("win,widths", [
    (41, (473, 474, 480, 486, 487, 488, 945, 958, 959)),       # uo = 472: one strip for 473..485 outputs, two up to 957
    (101, (410, 431, 455, 456, 457, 818, 840, 864, 865, 1272)),  # uo = 408 (456: the widest page one strip takes)
    (201, (314, 360, 405, 406, 407, 718, 719)),                  # wide windows (S beyond the mantissa trick), uo = 312
    (257, (258, 300, 377, 378, 379, 634, 635)),                  # the widest fused window, uo = 256
])
def test_extended_last_strip(prl, oracle, cuda_device, method, k, win, widths):
    """The last strip of a row takes the columns of a would-be extra strip when all of those are right-hand padding
    (binarize_fused.hip strip_layout): every width around the switch points, against the oracle."""
    for wd in widths:
        if wd <= win + 1 and method in (NICK, WOLFJOLION, FENG):
            continue  # no output columns
        ht = win + 37
        kinds = ["doc", "noise", "dark_corner"]
        _check(prl, oracle, cuda_device, _pages((ht, wd), kinds, seed=wd), method, win, k, 0)
    # the bit-plane hand-off to the morphology pass on an extended strip
    _check(prl, oracle, cuda_device, _pages((win + 40, widths[1]), ["doc", "noise"], seed=3), method, win, k, 2)


@pytest.mark.parametrize("method,k", [(SAUVOLA, 0.2), (NIBLACK, -0.2), (NICK, -0.1), (WOLFJOLION, 0.3), (FENG, 0.0)])
@pytest.mark.parametrize("win,outs", [
    (21, (488, 489, 491, 492, 493, 977, 984, 985, 1470, 2459)),   # 488 / 492 outputs per strip
    (31, (481, 482, 483, 961, 964, 965)),                         # 480 / 482
    (15, (497, 498, 499, 994, 996, 997)),                         # 496 / 498
])
def test_ragged_strips(prl, oracle, cuda_device, method, k, win, outs):
    """Float32 pipeline with a strip width that is not a multiple of 8 (binarize_fused.hip strip_layout: taken when it saves
    a strip): the last lane of every strip stores 2, 4 or 6 bytes.  Output widths around every switch point."""
    for ow in outs:
        wd = ow + 1 if method in (SAUVOLA, NIBLACK) else ow + win
        _check(prl, oracle, cuda_device, _pages((win + 45, wd), ["doc", "noise", "dark_corner"], seed=ow), method, win, k, 0)
    wd = outs[1] + 1 if method in (SAUVOLA, NIBLACK) else outs[1] + win
    _check(prl, oracle, cuda_device, _pages((win + 45, wd), ["doc", "noise"], seed=5), method, win, k, 2)   # bit plane: multiple of 8
    _check(prl, oracle, cuda_device, _pages((win + 45, wd), ["doc", "noise"], seed=6), method, win, k, -1)


@pytest.mark.parametrize("method,w,k", [(SAUVOLA, 31, 0.34), (NIBLACK, 31, 0.2), (NICK, 21, -0.1), (SAUVOLA, 101, 0.34)])
def test_queue_overflow_takes_the_exact_sweep_not_the_literal_pipeline(prl, oracle, cuda_device, method, w, k):
    """Pages of two-level stripes whose levels sit inside the float32 decision band on every second pixel (bench.py's
    adversarial_stripes: the closed forms of SURVEY.md A.1 / A.2 / A.4): the threshold sweep would queue half of all pixels, the
    refine queue overflows and the page is flagged (bit 0).  Those pixels are ~1e-4 from their threshold - the float64 interval
    test settles every one of them - so the page gets a second chance through the exact sweep (k_fused_exact: integer sums for
    every strip, refine64 inline, no queue) instead of the 48-bytes-per-pixel literal pipeline; only pages that ALSO overflow
    the fix-up list (true ties by the 10^5: the tests above) still go there.  With and without the morphology pass, through
    the batch and the page-table entry, beside ordinary pages, byte for byte the oracle."""
    import os
    import sys

    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    a, b, margin = bench.adversarial_stripes(method, w, k, None)
    assert margin < 2e-3
    h, wd = 1900, 2300                                       # (a wavefront's share of a page this size overflows its queue bucket)
    stripes = np.where(np.arange(wd) % 2 == 0, a, b).astype(np.uint8)[None, :].repeat(h, 0)
    doc = _pages((h, wd), ["doc"], seed=131)[0]
    mixed = stripes.copy()
    mixed[200:420, 300:800] = doc[200:420, 300:800]          # an island of ordinary content: interior, edge and border strips all queue
    for morph in (0, 2):
        st = _check(prl, oracle, cuda_device, [doc, stripes, mixed], method, w, k, morph)
        assert st.exact_sweep_pages == 2 and st.literal_pages == 0, (st.exact_sweep_pages, st.literal_pages)
    # a budget of zero literal pages does not stand in the way of the second chance (it bounds the literal pipeline only)
    prl.set_literal_page_budget(0)
    try:
        st = _check(prl, oracle, cuda_device, [stripes], method, w, k, 0)
        assert st.exact_sweep_pages == 1 and st.literal_pages == 0
    finally:
        prl.set_literal_page_budget(-1)


def test_on_threshold_patch_that_fills_a_queue_bucket_does_not_take_the_literal_pipeline(prl, oracle, cuda_device):
    """A flat patch of 100 x 160 pixels whose level sits on its own threshold (to ~1e-13), inside one strip of an ordinary page:
    the wavefronts that own it queue more than the 8192 entries of a refine-queue bucket, so the page is flagged (bit 0) - until
    round 5 that meant the literal pipeline for the whole page.  Now the exact sweep redoes the page, its interval test leaves the
    patch's pixels open (true ties), they go to the fix-up list (10^4 entries of 2^17) and the literal sequence decides them:
    no literal page.  Beside an ordinary page, with and without morphology."""
    w, c = 15, 200
    k = _flat_boundary_k(c, w, c)
    h, wd = 1500, 1800
    doc, other = _pages((h, wd), ["doc", "doc"], seed=141)
    doc[600:700 + w, 520:680 + w] = c     # (window-sized margin: 100 x 160 output pixels see nothing but the patch)
    for morph in (0, 2):
        st = _check(prl, oracle, cuda_device, [other, doc], SAUVOLA, w, k, morph)
        assert st.literal_pages == 0 and st.exact_pixels >= 100 * 160, (st.literal_pages, st.exact_sweep_pages, st.exact_pixels)
