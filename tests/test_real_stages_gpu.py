"""The reference's own inputs through the stages around the binarizers (tests/golden/stages/*.npz, data only; made by
tests/golden/make_golden_stages.py in the build container).

  denoise_*   the six images of test_data/denoise/salt_pepper (whole, 8UC3 as cv::imread hands them; the RGBA one also as
              8UC4) through prl::denoise at the header default 5.5 (denoiseNLM.h:32) and BASELINE config 4's 10
  chain_*     ten colour originals of test_data/binarize (>= 700 x 1200) through prl::deskew, prl::findAngle,
              prl::backgroundNormalization and BASELINE config 5 - stage by stage through the public entry points AND as the
              one-call chain (prl_hip_chain_pages_device), single pages and a same-size batch

Expected values are the CPU oracle's, recorded at generation time as CRC-32s (full outputs for small images, the final
skeleton bit-packed); the CPU tests recompute a sample of them with today's oracle.  The GPU test prints, per page, the
Hough search's point / segment counts against the room of its lists (prl_hip_last_deskew_stats) and writes the records to
gpurun_out/real_stages.jsonl (copied to profiles/r05/).
"""
import glob
import json
import os
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGES = os.path.join(ROOT, "tests", "golden", "stages")
DENOISE = sorted(glob.glob(os.path.join(STAGES, "denoise_*.npz")))
CHAIN = sorted(glob.glob(os.path.join(STAGES, "chain_*.npz")))


def _name(p):
    return os.path.basename(p)[:-4]


def crc(a) -> int:
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def _key(s: float) -> str:
    return ("%g" % s).replace(".", "p")


def test_fixture_set_is_the_one_the_verdict_asked_for():
    assert len(DENOISE) == 7 and len(CHAIN) >= 8
    assert sum(np.load(p)["bgr"].shape[2] == 4 for p in DENOISE) == 1
    angles = []
    for p in CHAIN:
        z = np.load(p)
        h, w, c = z["bgr"].shape
        assert c == 3 and min(h, w) >= 700 and max(h, w) >= 1200, (p, h, w)
        angles.append(float(z["deskew_angle"][0]))
    # straight pages, small skews, the exact -45 / -90 votes real rulings produce
    assert any(a == 0.0 for a in angles) and any(0 < abs(a) < 3 for a in angles) and any(abs(a) >= 45 for a in angles)
    assert sum(os.path.getsize(p) for p in DENOISE + CHAIN) < 25 * 2**20


@pytest.mark.parametrize("path", DENOISE, ids=[_name(p) for p in DENOISE])
def test_oracle_reproduces_the_denoise_fixtures(oracle, path):
    z = np.load(path)
    img = z["bgr"]
    if img.shape[0] * img.shape[1] > 400_000:
        pytest.skip("the 1440 x 972 photograph is recomputed on the GPU box only (CPU suite budget)")
    for s in z["strengths"]:
        out = oracle.denoise(img, float(s), threads=8)
        assert crc(out) == int(z["crc_" + _key(s)][0]), (_name(path), s)
        if "out_" + _key(s) in z.files:
            assert np.array_equal(out, z["out_" + _key(s)])


def _oracle_chain(oracle, bgr, params):
    w, k, morph, strength, thin = params
    rot, info = oracle.deskew(bgr)
    den = oracle.denoise(np.ascontiguousarray(rot), float(strength), threads=8)
    bg = oracle.bgnorm(np.ascontiguousarray(den))
    gray = oracle.bgr2gray(np.ascontiguousarray(bg))
    mask = oracle.binarize(np.ascontiguousarray(gray), oracle.make_params(oracle.SAUVOLA, int(w), float(k), int(morph)))
    return info, rot, den, bg, gray, mask, oracle.thin(255 - mask, int(thin))


@pytest.mark.parametrize("path", CHAIN[-1:] + CHAIN[:1], ids=[_name(p) for p in CHAIN[-1:] + CHAIN[:1]])
def test_oracle_reproduces_the_chain_fixtures(oracle, path):
    """Two of the ten on the CPU (a straight page and a -45 degree one); the GPU test holds all ten to the recorded CRCs."""
    z = np.load(path)
    info, rot, den, bg, gray, mask, skel = _oracle_chain(oracle, z["bgr"], z["chain_params"])
    assert info["angle"] == float(z["deskew_angle"][0]) and info["n_lines"] == int(z["deskew_segments"][0])
    assert info["otsu"] == int(z["deskew_otsu"][0]) and tuple(rot.shape) == tuple(z["deskew_shape"])
    for name, a in (("deskew", rot), ("chain_denoise", den), ("chain_bgnorm", bg), ("chain_gray", gray), ("chain_mask", mask),
                    ("chain_skeleton", skel)):
        assert crc(a) == int(z["crc_" + name][0]), (_name(path), name)
    assert crc(oracle.bgnorm(z["bgr"])) == int(z["crc_bgnorm_of_input"][0])
    want = np.unpackbits(z["chain_skeleton"], axis=1)[:, :skel.shape[1]] * 255
    assert np.array_equal(skel, want)


def _emit(records, name):
    for r in records:
        print(json.dumps(r))
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, name), "w") as f:
            for r in records:
                f.write(json.dumps(r) + "\n")


@pytest.mark.gpu
def test_real_photographs_through_denoise(prl, oracle, cuda_device):
    import torch

    records = []
    for path in DENOISE:
        z = np.load(path)
        img = z["bgr"]
        t = torch.from_numpy(img).to(cuda_device)
        for s in z["strengths"]:
            got = prl.denoise(t, float(s)).cpu().numpy()
            ok = crc(got) == int(z["crc_" + _key(s)][0])
            if not ok:   # say where
                want = oracle.denoise(img, float(s), threads=16)
                d = np.abs(got.astype(np.int32) - want)
                pytest.fail(f"{_name(path)} strength {s}: {int((d > 0).sum())} bytes differ from the oracle (max {int(d.max())})")
            if "out_" + _key(s) in z.files:
                assert np.array_equal(got, z["out_" + _key(s)])
            records.append({"image": _name(path), "shape": list(img.shape), "strength": float(s), "crc32": crc(got), "equal_to_oracle": ok})
        host = prl.denoise(img, 5.5)   # the cv::Mat wrapper's entry (host image, staged by the library)
        assert crc(host) == int(z["crc_5p5"][0])
    # the whole set once more against today's oracle, byte for byte, at the header default (the 1440 x 972 photograph included)
    for path in DENOISE:
        img = np.load(path)["bgr"]
        want = oracle.denoise(img, 5.5, threads=16)
        assert np.array_equal(prl.denoise(torch.from_numpy(img).to(cuda_device), 5.5).cpu().numpy(), want), _name(path)
    _emit(records, "real_denoise.jsonl")


@pytest.mark.gpu
def test_real_colour_scans_through_deskew_bgnorm_and_the_chain(prl, oracle, cuda_device):
    import torch

    records = []
    by_shape = {}
    for path in CHAIN:
        z = np.load(path)
        bgr = z["bgr"]
        w, k, morph, strength, thin = z["chain_params"]
        t = torch.from_numpy(bgr).to(cuda_device)
        rec = {"image": _name(path), "height": int(bgr.shape[0]), "width": int(bgr.shape[1])}

        # prl::findAngle on the page prl::deskew thresholds (deskew.cpp:224-226): segments and angle
        gray = prl.cvtColorBGR2GRAY(t)
        binary = torch.where(gray > int(z["deskew_otsu"][0]), 255, 0).to(torch.uint8)
        prl.deskew_stats(reset=True)
        ang, nseg = prl.findAngle(binary, return_segments=True)
        st = prl.deskew_stats().as_dict()
        assert ang == float(z["deskew_angle"][0]) and nseg == int(z["deskew_segments"][0]), (rec, ang, nseg)
        assert st["pages"] == 1 and st["segments"] == nseg and st["min_page_headroom"] >= 0
        rec.update(hough_points=st["points"], hough_segments=nseg, segment_capacity=st["segment_capacity"],
                   segment_headroom=st["min_page_headroom"], clipped=False, angle=ang)

        # prl::deskew
        outs, angles = prl.deskew(t[None])
        rot = outs[0]
        assert angles[0] == ang and tuple(rot.shape) == tuple(z["deskew_shape"]), (rec, angles[0], rot.shape)
        assert crc(rot.cpu().numpy()) == int(z["crc_deskew"][0]), f"{rec['image']}: prl::deskew differs from the oracle"
        # prl::backgroundNormalization of the page as it is
        assert crc(prl.backgroundNormalization(t).cpu().numpy()) == int(z["crc_bgnorm_of_input"][0]), rec

        # BASELINE config 5 stage by stage (device tensors between the public entry points) ...
        den = prl.denoise(rot.contiguous(), float(strength))
        assert crc(den.cpu().numpy()) == int(z["crc_chain_denoise"][0]), f"{rec['image']}: denoise stage"
        bg = prl.backgroundNormalization(den)
        assert crc(bg.cpu().numpy()) == int(z["crc_chain_bgnorm"][0]), f"{rec['image']}: backgroundNormalization stage"
        g8 = prl.cvtColorBGR2GRAY(bg)
        assert crc(g8.cpu().numpy()) == int(z["crc_chain_gray"][0]), f"{rec['image']}: gray stage"
        mask = prl.binarizeSauvola(g8, int(w), float(k), int(morph))
        assert crc(mask.cpu().numpy()) == int(z["crc_chain_mask"][0]), f"{rec['image']}: Sauvola stage"
        sk = prl.thinZhangSuen(prl.bitwise_not(mask))
        want = np.unpackbits(z["chain_skeleton"], axis=1)[:, :int(z["chain_shape"][1])] * 255
        assert np.array_equal(sk.cpu().numpy(), want), f"{rec['image']}: thinning stage"

        # ... and as the one call
        one, a1 = prl.process_pages(t, 3, prl.SAUVOLA, int(w), float(k), int(morph), denoise_strength=float(strength),
                                    thin=int(thin), deskew=True, background_normalization=True)
        assert a1 == ang and np.array_equal(one.cpu().numpy(), want), f"{rec['image']}: prl_hip_chain_pages_device"
        rec.update(result=list(want.shape), skeleton_pixels=int((want > 0).sum()), equal_to_oracle=True)
        records.append(rec)
        by_shape.setdefault(bgr.shape, []).append((bgr, want, ang))

    # the same-size pages as ONE batch through the chain (different angles -> different result sizes inside one call)
    shape, group = max(by_shape.items(), key=lambda kv: len(kv[1]))
    assert len(group) >= 3
    batch = torch.from_numpy(np.stack([g[0] for g in group])).to(cuda_device)
    prl.deskew_stats(reset=True)
    outs, angles = prl.process_pages(batch, 3, prl.SAUVOLA, 31, 0.34, 0, denoise_strength=10.0, thin=0, deskew=True,
                                     background_normalization=True)
    st = prl.deskew_stats().as_dict()
    for (bgr, want, ang), o, a in zip(group, outs, angles):
        assert a == ang and np.array_equal(o.cpu().numpy(), want)
    assert st["pages"] == len(group) and st["min_page_headroom"] >= 0
    records.append({"batch_of": len(group), "shape": list(shape), "hough": st, "result_sizes": sorted({tuple(o.shape) for o in outs})})
    _emit(records, "real_stages.jsonl")


@pytest.mark.gpu
def test_real_colour_scans_through_the_remaining_entry_points(prl, oracle, cuda_device):
    """The rest of the built surface on the same ten colour scans, against the oracle run beside it: the two local-variance
    binarizers (the one without filters is integer-exact; the filtered one carries its stated 1e-3 tolerance), Guo-Hall
    thinning of the Sauvola mask, the (2n+1)^2 closing / opening on its own, the BGRA form of the page through
    backgroundNormalization (the fourth channel is dropped, src/formatConvert.cpp:193-206), and prl::rotate by the page's own
    deskew angle."""
    import torch

    for path in CHAIN:
        z = np.load(path)
        bgr = z["bgr"]
        name = _name(path)
        t = torch.from_numpy(bgr).to(cuda_device)
        got = prl.binarizeByLocalVariancesWithoutFilters(t[None])[0].cpu().numpy()
        assert np.array_equal(got, oracle.binarize_lv_nofilters(bgr)), f"{name}: binarizeByLocalVariancesWithoutFilters"
        got = prl.binarizeByLocalVariances(t[None])[0].cpu().numpy()
        want = oracle.binarize_lv(bgr)
        assert (got != want).mean() <= 1e-3, f"{name}: binarizeByLocalVariances differs on {(got != want).mean():.2e} of the page"
        gray = oracle.bgr2gray(bgr)
        mask = prl.binarizeSauvola(torch.from_numpy(gray).to(cuda_device), 31, 0.34, 0)
        m_host = mask.cpu().numpy()
        assert np.array_equal(m_host, oracle.binarize(gray, oracle.make_params(oracle.SAUVOLA, 31, 0.34, 0))), name
        gh = prl.thinGuoHall(prl.bitwise_not(mask)).cpu().numpy()
        assert np.array_equal(gh, oracle.thin(255 - m_host, 1)), f"{name}: thinGuoHall"
        for it in (3, -2):
            assert np.array_equal(prl.morph(mask, it).cpu().numpy(), oracle.morph(m_host, it)), f"{name}: morphology {it}"
        bgra = np.concatenate([bgr, np.full(bgr.shape[:2] + (1,), 200, np.uint8)], axis=2)
        assert np.array_equal(prl.backgroundNormalization(torch.from_numpy(bgra).to(cuda_device)).cpu().numpy(), oracle.bgnorm(bgra)), name
        ang = float(z["deskew_angle"][0]) or 7.5
        r = prl.rotate(t[None], [ang])[0].cpu().numpy()
        assert np.array_equal(r, oracle.rotate(bgr, ang)), f"{name}: rotate({ang})"
