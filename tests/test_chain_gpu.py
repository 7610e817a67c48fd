"""GPU parity of the channel adapters and the device-resident chain against the CPU oracle."""
import os
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _colour(shape, c, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, shape + (c,), dtype=np.uint8)


@pytest.mark.parametrize("c", [3, 4])
@pytest.mark.parametrize("shape", [(40, 64), (33, 97), (17, 5), (1, 1), (50, 1026)])
def test_bgr2gray_matches_oracle(prl, oracle, cuda_device, shape, c):
    import torch

    imgs = np.stack([_colour(shape, c, s) for s in (1, 2)])
    got = prl.cvtColorBGR2GRAY(torch.from_numpy(imgs).to(cuda_device)).cpu().numpy()
    for i in range(2):
        assert np.array_equal(got[i], oracle.bgr2gray(imgs[i]))


def test_bgr2gray_extremes_and_views(prl, oracle, cuda_device):
    import torch

    # saturation corners of the fixed-point formula
    px = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [1, 1, 1], [254, 255, 253], [128, 127, 129]]], np.uint8)
    got = prl.cvtColorBGR2GRAY(torch.from_numpy(px).to(cuda_device)).cpu().numpy()
    assert np.array_equal(got, oracle.bgr2gray(px)) and got[0, 0] == 255 and got[0, 1] == 0
    # unaligned ROI view with a row step larger than the row, output into a guarded view
    h, w = 21, 37
    big = torch.from_numpy(_colour((h + 2, w + 9), 3, 7)).to(cuda_device)
    view = big[1:h + 1, 5:5 + w]                      # rows start at byte offset 15 (mod 4 = 3)
    out = torch.full((h + 2, w + 8), 9, dtype=torch.uint8, device=cuda_device)
    prl.cvtColorBGR2GRAY(view, out=out[1:h + 1, 3:3 + w])
    o = out.cpu().numpy()
    assert np.array_equal(o[1:h + 1, 3:3 + w], oracle.bgr2gray(np.ascontiguousarray(view.cpu().numpy())))
    o[1:h + 1, 3:3 + w] = 9
    assert (o == 9).all()


@pytest.mark.parametrize("shape", [(30, 64), (19, 45), (3, 2)])
def test_gray2bgr_and_invert(prl, cuda_device, shape):
    import torch

    rng = np.random.default_rng(3)
    g = rng.integers(0, 256, (2,) + shape, dtype=np.uint8)
    t = torch.from_numpy(g).to(cuda_device)
    for c in (3, 4):
        got = prl.cvtColorGRAY2BGR(t, c).cpu().numpy()
        want = np.repeat(g[..., None], c, axis=-1)
        if c == 4:
            want[..., 3] = 255
        assert np.array_equal(got, want)
    assert np.array_equal(prl.bitwise_not(t).cpu().numpy(), 255 - g)
    # in place, on an unaligned view
    big = torch.from_numpy(rng.integers(0, 256, (shape[0] + 2, shape[1] + 7), dtype=np.uint8)).to(cuda_device)
    ref = big.cpu().numpy().copy()
    v = big[1:shape[0] + 1, 3:3 + shape[1]]
    prl.bitwise_not(v, out=v)
    ref[1:shape[0] + 1, 3:3 + shape[1]] = 255 - ref[1:shape[0] + 1, 3:3 + shape[1]]
    assert np.array_equal(big.cpu().numpy(), ref)


def _oracle_chain(oracle, img, channels, method, w, k, morph, strength, thin):
    cur = img
    if strength is not None:
        cur = oracle.denoise(cur, strength, threads=8)
    if channels != 1:
        cur = oracle.bgr2gray(cur)
    mask = oracle.binarize(cur, oracle.make_params(method, w, k, morph))
    if thin < 0:
        return mask
    return oracle.thin(255 - mask, thin)


@pytest.mark.parametrize("thin", [-1, 0, 1])
@pytest.mark.parametrize("strength", [None, 8.0])
def test_chain_matches_the_composed_oracle(prl, oracle, cuda_device, strength, thin):
    import torch
    from prlib_amd import synth

    h, w = 96, 128
    pages = []
    for i in range(2):
        gray = synth.page_numpy(h, w, index=i + 11)
        rng = np.random.default_rng(i)
        pages.append(np.clip(gray[..., None].astype(np.int32) + rng.normal(0, 9, (h, w, 3)), 0, 255).round().astype(np.uint8))
    pages = np.stack(pages)
    got = prl.process_pages(torch.from_numpy(pages).to(cuda_device), 3, prl.SAUVOLA, 31, 0.34, 1,
                            denoise_strength=strength, thin=thin).cpu().numpy()
    for i in range(2):
        want = _oracle_chain(oracle, pages[i], 3, oracle.SAUVOLA, 31, 0.34, 1, strength, thin)
        assert got[i].shape == want.shape
        assert np.array_equal(got[i], want), f"page {i}: {int((got[i] != want).sum())} mismatching pixels"


def test_chain_gray_input_other_methods_and_errors(prl, oracle, cuda_device):
    import torch
    from prlib_amd import synth

    page = synth.page_numpy(120, 150, index=4)
    t = torch.from_numpy(page).to(cuda_device)
    for method, w, k in ((prl.NICK, 21, -0.1), (prl.WOLFJOLION, 31, 0.3)):
        got = prl.process_pages(t, 1, method, w, k, 0, thin=0).cpu().numpy()
        want = _oracle_chain(oracle, page, 1, method, w, k, 0, None, 0)
        assert np.array_equal(got, want)
    with pytest.raises(Exception):          # fastNlMeansDenoisingColored asserts 3/4 channels
        prl.process_pages(t, 1, prl.SAUVOLA, 31, 0.34, 0, denoise_strength=5.0)
    with pytest.raises(Exception):
        prl.process_pages(t, 1, prl.SAUVOLA, 31, 0.34, 0, thin=7)
    with pytest.raises(ValueError):         # even window: std::invalid_argument in the reference
        prl.process_pages(t, 1, prl.SAUVOLA, 30, 0.34, 0)


def test_more_pages_than_one_grid_dimension(prl, oracle, cuda_device):
    """70 000 tiny pages per call: every entry point that keeps the page index in a 65 535-limited grid dimension
    has to chunk."""
    import torch

    rng = np.random.default_rng(5)
    n, h, w = 70000, 20, 24
    gray = rng.integers(0, 256, (n, h, w), dtype=np.uint8)
    t = torch.from_numpy(gray).to(cuda_device)
    probe = (0, 1, 32767, 32768, 65535, 65536, n - 1)
    mask = prl.binarizeSauvola(t, 7, 0.2, 1).cpu().numpy()
    p = oracle.make_params(oracle.SAUVOLA, 7, 0.2, 1)
    for i in probe:
        assert np.array_equal(mask[i], oracle.binarize(gray[i], p)), i
    inv = prl.bitwise_not(t).cpu().numpy()
    assert np.array_equal(inv[list(probe)], 255 - gray[list(probe)])
    skel = prl.thinGuoHall(torch.from_numpy(np.where(gray > 128, 255, 0).astype(np.uint8)).to(cuda_device)).cpu().numpy()
    for i in probe:
        assert np.array_equal(skel[i], oracle.thin(np.where(gray[i] > 128, 255, 0).astype(np.uint8), 1)), i
    bgr = torch.from_numpy(np.repeat(gray[:66000, :, :, None], 3, axis=3).copy()).to(cuda_device)
    g2 = prl.cvtColorBGR2GRAY(bgr).cpu().numpy()
    for i in (0, 32768, 65535, 65999):
        assert np.array_equal(g2[i], oracle.bgr2gray(np.repeat(gray[i][:, :, None], 3, axis=2).copy())), i


def _oracle_chain5(oracle, img, channels, w, k, morph, strength, thin, bgnorm=True, deskew=True):
    """BASELINE config 5 composed from the oracle's stages: deskew -> denoise -> backgroundNormalization -> gray ->
    Sauvola -> bitwise_not -> thinning."""
    cur, info = (oracle.deskew(img) if deskew else (img, dict(angle=0.0)))
    if strength is not None:
        cur = oracle.denoise(np.ascontiguousarray(cur), strength, threads=8)
    if bgnorm:
        cur = oracle.bgnorm(np.ascontiguousarray(cur))
    if cur.ndim == 3:
        cur = oracle.bgr2gray(np.ascontiguousarray(cur))
    mask = oracle.binarize(np.ascontiguousarray(cur), oracle.make_params(oracle.SAUVOLA, w, k, morph))
    return (mask if thin < 0 else oracle.thin(255 - mask, thin)), info["angle"]


@pytest.mark.parametrize("channels", [3, 4])
def test_config5_chain_matches_the_composed_oracle(prl, oracle, cuda_device, channels):
    """deskew -> NL-means -> backgroundNormalization -> Sauvola -> Zhang-Suen thinning (BASELINE config 5) on a batch whose
    pages come out in different sizes: skewed pages as max(W,H) squares, a straight and a blank page unchanged."""
    import torch
    from prlib_amd import synth

    h, w = 150, 208
    pages = []
    for i, skew in enumerate((2.0, 0.0, -3.0)):
        g = synth.text_page_numpy(h, w, 40 + i, skew_deg=skew, shading=0.3)
        rng = np.random.default_rng(i)
        col = np.clip(g[..., None].astype(np.int32) + rng.normal(0, 6, (h, w, channels)), 0, 255).round().astype(np.uint8)
        pages.append(col)
    pages.append(np.full((h, w, channels), 228, np.uint8))   # blank: no segments, stays 150 x 208
    batch = np.stack(pages)
    outs, angles = prl.process_pages(torch.from_numpy(batch).to(cuda_device), channels, prl.SAUVOLA, 31, 0.34, 0,
                                     denoise_strength=10.0, thin=0, deskew=True, background_normalization=True)
    sizes = set()
    for i in range(len(pages)):
        want, ang = _oracle_chain5(oracle, batch[i], channels, 31, 0.34, 0, 10.0, 0)
        got = outs[i].cpu().numpy()
        assert angles[i] == ang
        assert got.shape == want.shape, (i, got.shape, want.shape)
        assert np.array_equal(got, want), f"page {i}: {int((got != want).sum())} mismatching pixels"
        sizes.add(got.shape)
    assert len(sizes) >= 2 and angles[-1] == 0.0


@pytest.mark.parametrize("overlap", ["2", "1", "0"])
def test_config5_chain_in_several_passes(prl, oracle, cuda_device, tmp_path, overlap):
    """With more pages than one pass of the angle search takes, the chain searches the angles of pass k+1 on a side stream
    (helper thread) while pass k is rotated, denoised and binarized.  PRL_HIP_DESKEW_WORK_MB (read once per process) is
    shrunk in a child process so that 5 small pages need 5 passes; the result must equal the one-pass result, which the
    test above compares with the oracle page by page."""
    import subprocess
    import sys
    import torch
    from prlib_amd import synth

    h, w = 150, 208
    pages = []
    for i, skew in enumerate((2.0, 0.0, -3.0, 1.0)):
        g = synth.text_page_numpy(h, w, 60 + i, skew_deg=skew, shading=0.3)
        pages.append(np.repeat(g[..., None], 3, axis=2))
    pages.append(np.full((h, w, 3), 228, np.uint8))
    batch = np.stack(pages)
    np.save(tmp_path / "in.npy", batch)
    outs, angles = prl.process_pages(torch.from_numpy(batch).to(cuda_device), 3, prl.SAUVOLA, 31, 0.34, 0,
                                     denoise_strength=10.0, thin=0, deskew=True, background_normalization=True)
    want0, ang0 = _oracle_chain5(oracle, batch[0], 3, 31, 0.34, 0, 10.0, 0)
    assert angles[0] == ang0 and np.array_equal(outs[0].cpu().numpy(), want0)
    code = r'''
import numpy as np, torch, sys
sys.path.insert(0, %r)
import prlib_amd
prlib_amd._capi.use_library(prlib_amd._capi.HOOKS_LIB_PATH)   # the build that reads the PRL_HIP_* tuning knobs
batch = np.load(%r)
outs, angles = prlib_amd.process_pages(torch.from_numpy(batch).cuda(), 3, prlib_amd.SAUVOLA, 31, 0.34, 0,
                                       denoise_strength=10.0, thin=0, deskew=True, background_normalization=True)
np.savez(%r, angles=np.asarray(angles), **{"p%%d" %% i: o.cpu().numpy() for i, o in enumerate(outs)})
print("DONE")
''' % (ROOT, str(tmp_path / "in.npy"), str(tmp_path / "out.npz"))
    # PRL_HIP_CHAIN_OVERLAP: 2 = head / body / tail around the next pass's search (default), 1 = search beside all stages, 0 = serial
    env = dict(os.environ, PRL_HIP_DESKEW_WORK_MB="1", PRL_HIP_CHAIN_OVERLAP=overlap)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "DONE" in r.stdout, r.stdout + r.stderr
    z = np.load(tmp_path / "out.npz")
    assert np.array_equal(z["angles"], np.asarray(angles))
    for i in range(len(pages)):
        assert np.array_equal(z["p%d" % i], outs[i].cpu().numpy()), i


def test_tail_aware_passes_give_the_same_pages(prl, cuda_device):
    """More pages than one default pass (192 with denoise) and one page that is almost all ink: the chain's census of the dark
    pixels finds that the other pages' search hides behind that page's, and runs the batch as one large pass instead of
    192 + 68 (glue.hip, round 5).  Page by page the results equal those of the same pages sent in two calls small enough to be
    single default passes (which take no census)."""
    import torch
    from prlib_amd import synth

    h, w, n = 120, 176, 260
    pages = np.stack([np.repeat(synth.text_page_numpy(h, w, 300 + i, skew_deg=(i % 7) - 3.0, shading=0.2)[..., None], 3, axis=2) for i in range(n)])
    rng = np.random.default_rng(5)
    pages[17] = np.clip(rng.normal(40, 12, (h, w, 3)), 0, 255).astype(np.uint8)      # a dark photograph: nearly every pixel is a point
    pages[17, 30:90, 20:150] = 235                                                  # ... with a bright sheet in it
    t = torch.from_numpy(pages).to(cuda_device)
    kw = dict(denoise_strength=5.5, thin=0, deskew=True, background_normalization=True)
    prl.deskew_stats(reset=True)
    outs, angles = prl.process_pages(t, 3, prl.SAUVOLA, 15, 0.34, 0, **kw)
    assert prl.deskew_stats().pages == n
    ref_outs, ref_angles = [], []
    for lo, hi in ((0, 130), (130, n)):
        o, a = prl.process_pages(t[lo:hi], 3, prl.SAUVOLA, 15, 0.34, 0, **kw)
        ref_outs += o
        ref_angles += list(a)
    assert list(angles) == ref_angles
    for i in range(n):
        assert outs[i].shape == ref_outs[i].shape and torch.equal(outs[i], ref_outs[i]), i


def test_chain_on_host_pages(prl, oracle, cuda_device):
    """prl_hip_chain_batch_host: the caller's host pages through upload -> chain -> download, in device chunks of two pages
    (PRL_HIP_CHAIN_HOST_PAGES=2 in a child process) so that upload, chain and download of different chunks overlap; equal
    to the device-resident entry page by page, page 0 also to the composed oracle."""
    import subprocess
    import sys
    import torch
    from prlib_amd import synth

    h, w = 150, 208
    pages = []
    for i, skew in enumerate((2.0, 0.0, -3.0, 1.0, -1.5)):
        g = synth.text_page_numpy(h, w, 80 + i, skew_deg=skew, shading=0.3)
        pages.append(np.repeat(g[..., None], 3, axis=2))
    batch = np.stack(pages)
    outs, angles = prl.process_pages(torch.from_numpy(batch).to(cuda_device), 3, prl.SAUVOLA, 31, 0.34, 0,
                                     denoise_strength=10.0, thin=0, deskew=True, background_normalization=True)
    got, ang = prl.process_pages_host(list(batch), prl.SAUVOLA, 31, 0.34, 0, denoise_strength=10.0, thin=0, deskew=True,
                                      background_normalization=True, n_devices=1)
    assert np.array_equal(ang, np.asarray(angles))
    for i in range(len(pages)):
        assert np.array_equal(got[i], outs[i].cpu().numpy()), i
    want0, ang0 = _oracle_chain5(oracle, batch[0], 3, 31, 0.34, 0, 10.0, 0)
    assert ang[0] == ang0 and np.array_equal(got[0], want0)
    # gray pages without deskew: uniform result size through the same entry
    gray = [np.ascontiguousarray(p[..., 0]) for p in pages]
    got2, ang2 = prl.process_pages_host(gray, prl.NICK, 21, -0.1, 1, background_normalization=True)
    ref2 = prl.process_pages(torch.from_numpy(np.stack(gray)).to(cuda_device), 1, prl.NICK, 21, -0.1, 1,
                             background_normalization=True).cpu().numpy()
    assert all(np.array_equal(got2[i], ref2[i]) for i in range(len(gray))) and not ang2.any()
    code = r'''
import numpy as np, sys
sys.path.insert(0, %r)
import prlib_amd
prlib_amd._capi.use_library(prlib_amd._capi.HOOKS_LIB_PATH)   # the build that reads the PRL_HIP_* tuning knobs
batch = np.load(%r)
outs, angles = prlib_amd.process_pages_host(list(batch), prlib_amd.SAUVOLA, 31, 0.34, 0, denoise_strength=10.0, thin=0, deskew=True,
                                            background_normalization=True, n_devices=1)
np.savez(%r, angles=angles, **{"p%%d" %% i: o for i, o in enumerate(outs)})
print("DONE")
'''
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        np.save(os.path.join(td, "in.npy"), batch)
        env = dict(os.environ, PRL_HIP_CHAIN_HOST_PAGES="2")
        r = subprocess.run([sys.executable, "-c", code % (ROOT, os.path.join(td, "in.npy"), os.path.join(td, "out.npz"))],
                           capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0 and "DONE" in r.stdout, r.stdout + r.stderr
        z = np.load(os.path.join(td, "out.npz"))
        assert np.array_equal(z["angles"], ang)
        for i in range(len(pages)):
            assert np.array_equal(z["p%d" % i], got[i]), i


def test_host_entries_with_several_device_workers(prl, cuda_device, tmp_path):
    """The *_batch_host entries start one worker thread per device, each on its block of the page list.  A one-GPU box has
    one worker; PRL_HIP_FAKE_DEVICES=3 (child process) starts three, all on the real device: blocks of 3 + 2 + 2 (binarize)
    and 2 + 2 + 1 (chain) pages, results in the caller's order and equal to the one-worker results."""
    import subprocess
    import sys
    from prlib_amd import synth

    gray = np.stack([synth.page_numpy(300, 520, index=70 + i) for i in range(7)])
    col = np.stack([np.repeat(synth.text_page_numpy(150, 208, 95 + i, skew_deg=s, shading=0.3)[..., None], 3, axis=2)
                    for i, s in enumerate((2.0, -1.0, 0.0, 3.0, -2.5))])
    np.save(tmp_path / "gray.npy", gray)
    np.save(tmp_path / "col.npy", col)
    want_b = prl.binarize_pages_host(list(gray), prl.make_params(prl.NICK, 31, -0.1, 1), n_devices=1)
    want_c, want_a = prl.process_pages_host(list(col), prl.SAUVOLA, 31, 0.34, 0, denoise_strength=10.0, thin=0, deskew=True,
                                            background_normalization=True, n_devices=1)
    code = r'''
import numpy as np, sys
sys.path.insert(0, %r)
import prlib_amd
prlib_amd._capi.use_library(prlib_amd._capi.HOOKS_LIB_PATH)   # the build that reads the PRL_HIP_* tuning knobs
gray, col = np.load(%r), np.load(%r)
b = prlib_amd.binarize_pages_host(list(gray), prlib_amd.make_params(prlib_amd.NICK, 31, -0.1, 1))
c, a = prlib_amd.process_pages_host(list(col), prlib_amd.SAUVOLA, 31, 0.34, 0, denoise_strength=10.0, thin=0, deskew=True,
                                    background_normalization=True)
np.savez(%r, b=b, a=a, **{"c%%d" %% i: x for i, x in enumerate(c)})
print("DONE")
''' % (ROOT, str(tmp_path / "gray.npy"), str(tmp_path / "col.npy"), str(tmp_path / "out.npz"))
    env = dict(os.environ, PRL_HIP_FAKE_DEVICES="3")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "DONE" in r.stdout, r.stdout + r.stderr
    z = np.load(tmp_path / "out.npz")
    assert np.array_equal(z["b"], want_b) and np.array_equal(z["a"], want_a)
    for i in range(len(col)):
        assert np.array_equal(z["c%d" % i], want_c[i]), i


def test_release_workspace_between_calls(prl, cuda_device):
    """prl_hip_release_workspace frees every cached buffer (stage area, angle-search workspace, host page buffers and pinned
    slots, per-stream workspaces); the next calls allocate again and give the same results."""
    import torch
    from prlib_amd import synth, _capi

    pages = [np.repeat(synth.text_page_numpy(120, 176, 90 + i, skew_deg=s, shading=0.2)[..., None], 3, axis=2) for i, s in enumerate((1.5, -2.0))]
    kw = dict(denoise_strength=10.0, thin=0, deskew=True, background_normalization=True)
    a, ang_a = prl.process_pages_host(pages, prl.SAUVOLA, 31, 0.34, 0, n_devices=1, **kw)
    d, ang_d = prl.process_pages(torch.from_numpy(np.stack(pages)).to(cuda_device), 3, prl.SAUVOLA, 31, 0.34, 0, **kw)
    d = [x.cpu().numpy() for x in d]
    torch.cuda.synchronize()
    _capi.check(_capi.lib().prl_hip_release_workspace())
    b, ang_b = prl.process_pages_host(pages, prl.SAUVOLA, 31, 0.34, 0, n_devices=1, **kw)
    e, ang_e = prl.process_pages(torch.from_numpy(np.stack(pages)).to(cuda_device), 3, prl.SAUVOLA, 31, 0.34, 0, **kw)
    assert np.array_equal(ang_a, ang_b) and np.array_equal(np.asarray(ang_d), np.asarray(ang_e)) and np.array_equal(ang_a, np.asarray(ang_d))
    for i in range(len(pages)):
        assert np.array_equal(a[i], b[i]) and np.array_equal(d[i], e[i].cpu().numpy()) and np.array_equal(a[i], d[i])


def test_chain_stage_subsets(prl, oracle, cuda_device):
    import torch
    from prlib_amd import synth

    g = synth.text_page_numpy(130, 170, 50, skew_deg=1.5, shading=0.4)
    t = torch.from_numpy(np.stack([g, g[::-1].copy()])).to(cuda_device)
    # gray pages: deskew + backgroundNormalization + Sauvola, no thinning
    outs, angles = prl.process_pages(t, 1, prl.SAUVOLA, 15, 0.2, 1, deskew=True, background_normalization=True)
    for i, src in enumerate((g, g[::-1].copy())):
        want, ang = _oracle_chain5(oracle, src, 1, 15, 0.2, 1, None, -1)
        assert angles[i] == ang and np.array_equal(outs[i].cpu().numpy(), want)
    # backgroundNormalization without deskew goes through the uniform-size entry
    got = prl.process_pages(t, 1, prl.SAUVOLA, 15, 0.2, 0, background_normalization=True, thin=1).cpu().numpy()
    for i, src in enumerate((g, g[::-1].copy())):
        want, _ = _oracle_chain5(oracle, src, 1, 15, 0.2, 0, None, 1, deskew=False)
        assert np.array_equal(got[i], want)
    # the uniform entry refuses deskew (per-page sizes)
    from prlib_amd import _capi
    import ctypes as C
    cp = _capi.ChainParams()
    _capi.lib().prl_hip_default_chain_params(C.byref(cp))
    cp.deskew = 1
    o = torch.empty((2, 130, 170), dtype=torch.uint8, device=cuda_device)
    assert _capi.lib().prl_hip_chain_batch_device(C.byref(cp), 2, 1, t.data_ptr(), t.stride(0), t.stride(1), 170, 130,
                                                  o.data_ptr(), o.stride(0), o.stride(1), None) == _capi.PRL_ERR_BAD_ARG


def test_chain_without_thinning_in_deferred_mode_redoes_overflow_pages(prl, oracle, cuda_device):
    """with prl_hip_set_deferred_completion(1) the chain's binarize stage reads library-owned scratch, so its
    flags must be resolved inside the chain.  Page 0 (flat, threshold exactly on the decision boundary) overflows the fix-up
    list and is redone literally; the second call would overwrite the scratch a late redo would read."""
    import torch
    from test_binarize_gpu import _flat_boundary_k, _pages

    w, c = 15, 200
    k = _flat_boundary_k(c, w, c)
    gray = [np.full((640, 700), c, np.uint8), _pages((640, 700), ["doc"], seed=29)[0]]
    other = _pages((640, 700), ["doc", "doc"], seed=31)
    bgr = lambda g: np.repeat(np.stack(g)[..., None], 3, axis=-1)   # BGR2GRAY of a gray-replicated page is the page
    t1, t2 = torch.from_numpy(bgr(gray)).to(cuda_device), torch.from_numpy(bgr(other)).to(cuda_device)
    prl.set_deferred_completion(True)
    try:
        got1 = prl.process_pages(t1, 3, prl.SAUVOLA, w, k, 0)
        got2 = prl.process_pages(t2, 3, prl.SAUVOLA, w, k, 0)      # reuses (overwrites) the staging area of the first call
        prl.finish(cuda_device)
        g1, g2 = got1.cpu().numpy(), got2.cpu().numpy()
    finally:
        prl.set_deferred_completion(False)
    po = oracle.make_params(oracle.SAUVOLA, w, k, 0)
    for got, pages in ((g1, gray), (g2, other)):
        for i in range(2):
            want = oracle.binarize(oracle.bgr2gray(bgr([pages[i]])[0]), po)
            assert np.array_equal(got[i], want), int((got[i] != want).sum())
