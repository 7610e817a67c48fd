"""The CPU oracle (oracle/prl_oracle*.c) against (1) an independently written whole-plane numpy model,
(2) hand-derived known answers, (3) the committed golden fixtures.  PARITY UNPINNED against the real
reference: it has no tests or golden outputs and cannot be built here (see oracle/prl_oracle.h)."""
import glob
import os

import numpy as np
import pytest

from oracle import numpy_model as nm

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cases(rng, n):
    for trial in range(n):
        h, w = int(rng.integers(20, 90)), int(rng.integers(20, 90))
        kind = trial % 4
        if kind == 0:
            img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        elif kind == 1:
            img = np.clip(rng.normal(215, 12, (h, w)), 0, 255).round().astype(np.uint8)
        elif kind == 2:
            img = (rng.integers(0, 2, (h, w)) * 255).astype(np.uint8)
        else:
            img = np.full((h, w), int(rng.integers(0, 256)), np.uint8)
            img[h // 3:h // 2, w // 4:w // 2] = int(rng.integers(0, 256))
        yield img


@pytest.mark.parametrize("method", range(5))
def test_oracle_equals_numpy_model(oracle, method):
    rng = np.random.default_rng(100 + method)
    checked = 0
    for img in _cases(rng, 24):
        win = int(rng.choice([3, 5, 7, 15, 21, 31, 101]))
        k = float(rng.choice([0.34, 0.01, -0.01, -0.2, 0.5]))
        morph = int(rng.choice([0, 0, 1, 2, -1, -2]))
        p = oracle.make_params(method, win, k, morph)
        try:
            a = oracle.binarize(img, p)
        except oracle.OracleError as e:
            assert e.status == oracle.PRL_ERR_EMPTY_RECT
            with pytest.raises(nm.EmptyRect):
                nm.binarize(img, method, win, k, morph)
            continue
        assert np.array_equal(a, nm.binarize(img, method, win, k, morph))
        t_c = oracle.threshold_plane(img, p)
        _, t_np, _ = nm.threshold_plane(img, method, win, k)
        assert np.array_equal(t_c, t_np, equal_nan=True)  # float64 thresholds agree to the last bit
        checked += 1
    assert checked >= 10


def test_known_answers(oracle):
    # flat page p: m = p (w-1)^2 / w^2, var = p^2 (w-1)^2/w^2 (1 - (w-1)^2/w^2)   
    p, w, k = 255.0, 31, 0.34
    r = (w - 1) ** 2 / w ** 2
    m, s = p * r, (p * p * r * (1 - r)) ** 0.5
    want = m * (1 + k * (s / 128 - 1))
    t = oracle.threshold_plane(np.full((64, 64), 255, np.uint8), oracle.make_params(oracle.SAUVOLA, w, k, 0))
    assert abs(want - 197.0565) < 1e-3 and np.allclose(t, want, rtol=0, atol=1e-9)
    assert oracle.binarize(np.full((64, 64), 255, np.uint8), oracle.make_params(oracle.SAUVOLA, w, k, 0)).min() == 255
    for method in range(5):  # all-zero page: p = 0 is never > T8
        assert oracle.binarize(np.zeros((50, 60), np.uint8), oracle.make_params(method, 15, 0.2, 0)).max() == 0


def test_output_sizes_and_errors(oracle):
    page = np.zeros((100, 120), np.uint8)
    assert oracle.binarize(page, oracle.make_params(oracle.SAUVOLA, 31, 0.3, 0)).shape == (99, 119)
    assert oracle.binarize(page, oracle.make_params(oracle.NIBLACK, 101, 0.3, 0)).shape == (100, 120)  # clamped, even w
    assert oracle.binarize(page, oracle.make_params(oracle.NICK, 21, 0.3, 0)).shape == (79, 99)
    for bad in (30, 1, 0, -5):
        st, _ = oracle.geometry(oracle.make_params(oracle.SAUVOLA, bad, 0.3, 0), 120, 100)
        assert st == oracle.PRL_ERR_BAD_WINDOW
    st, _ = oracle.geometry(oracle.make_params(oracle.SAUVOLA, 31, 0.3, 0), 0, 0)
    assert st == oracle.PRL_ERR_EMPTY
    st, _ = oracle.geometry(oracle.make_params(oracle.WOLFJOLION, 101, 0.3, 0), 120, 100)
    assert st == oracle.PRL_ERR_EMPTY_RECT


def test_sat_u8_is_cvround_then_clamp(oracle):
    cases = {0.5: 0, 1.5: 2, 2.5: 2, 3.5: 4, 254.5: 254, 255.5: 255, 300.0: 255, -0.5: 0, -3.0: 0,
             float("nan"): 0, float("inf"): 0, -float("inf"): 0, 2147483647.4: 255, 2147483647.5: 0, 1e10: 0, -1e10: 0}
    for v, want in cases.items():
        assert oracle.sat_u8(v) == want, v
    arr = np.array(list(cases.keys()))
    assert list(nm.sat_u8(arr)) == list(cases.values())


@pytest.mark.parametrize("n", [1, 2, 3, -1, -2])
def test_morphology_equals_scipy_rect(oracle, n):
    rng = np.random.default_rng(n + 10)
    mask = ((rng.random((70, 95)) < 0.3) * 255).astype(np.uint8)
    assert np.array_equal(oracle.morph(mask, n), nm.morph(mask, n))


def test_bgr2gray_and_otsu(oracle):
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [10, 200, 90]]], np.uint8)
    assert list(oracle.bgr2gray(px)[0]) == [29, 150, 76, 255, (10 * 1868 + 200 * 9617 + 90 * 4899 + 8192) >> 14]
    img = np.concatenate([np.full((20, 30), 50, np.uint8), np.full((20, 30), 200, np.uint8)], axis=1)
    thr, out = oracle.otsu(img)
    assert 50 <= thr < 200 and out[:, :30].max() == 0 and out[:, 30:].min() == 255


def _nlm_naive(img, h, lut):
    a = img if img.ndim == 3 else img[:, :, None]
    hh, ww, ch = a.shape
    e = np.pad(a, ((13, 13), (13, 13), (0, 0)), mode="reflect").astype(np.int64)
    out = np.zeros_like(a)
    for i in range(hh):
        for j in range(ww):
            est, ws = np.zeros(ch, np.int64), 0
            t0 = e[i + 10:i + 17, j + 10:j + 17]
            for dy in range(-10, 11):
                for dx in range(-10, 11):
                    t1 = e[i + 10 + dy:i + 17 + dy, j + 10 + dx:j + 17 + dx]
                    d = int(((t0 - t1) ** 2).sum()) >> 6
                    wgt = int(lut[d]) if d < len(lut) else 0
                    ws += wgt
                    est += wgt * e[i + 13 + dy, j + 13 + dx]
            out[i, j] = np.minimum((est + ws // 2) // ws, 255)
    return out if img.ndim == 3 else out[:, :, 0]


def test_nlm_equals_direct_definition(oracle):
    rng = np.random.default_rng(1)
    img = np.clip(rng.normal(128, 20, (10, 12)), 0, 255).astype(np.uint8)
    assert np.array_equal(oracle.nlm_planes(img, 10.0), _nlm_naive(img, 10.0, oracle.nlm_weights(1, 10.0)))
    img2 = np.clip(rng.normal(128, 6, (8, 9, 2)), 0, 255).astype(np.uint8)
    assert np.array_equal(oracle.nlm_planes(img2, 3.0), _nlm_naive(img2, 3.0, oracle.nlm_weights(2, 3.0)))


def test_nlm_weight_table(oracle):
    lut = oracle.nlm_weights(1, 10.0)
    assert len(lut) == 49785 and lut[0] == 19096          # fixed_point_mult = INT_MAX / (21*21*255)
    assert 500 < np.nonzero(lut)[0].max() < 560           # zero beyond ~h^2 ln(1000) 49/64
    assert np.all(np.diff(lut.astype(np.int64)) <= 0)
    lut2 = oracle.nlm_weights(2, 3.0)
    assert len(lut2) == 99570 and 85 < np.nonzero(lut2)[0].max() < 105


def test_lab_conversion_sanity(oracle):
    gray = np.full((3, 3, 3), 200, np.uint8)
    lab = oracle.lbgr2lab(gray)
    assert tuple(lab[0, 0]) == (232, 128, 128)            # L = 116 cbrt(200/255) - 16 = 90.96 -> *2.55
    assert np.array_equal(oracle.lab2lbgr(lab), gray)
    rng = np.random.default_rng(3)
    bgr = rng.integers(0, 256, (32, 32, 3), dtype=np.uint8)
    assert np.abs(oracle.lab2lbgr(oracle.lbgr2lab(bgr)).astype(int) - bgr).max() <= 4   # 8-bit Lab quantisation
    assert oracle.lab2lbgr(oracle.lbgr2lab(bgr), 4)[:, :, 3].min() == 255


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "0*.npz"))))
def test_golden_binarize_fixtures(oracle, path):
    z = np.load(path)
    src = z["source"]
    gray = src if src.ndim == 2 else oracle.bgr2gray(np.ascontiguousarray(src))
    assert np.array_equal(gray, z["gray"])
    n = 0
    for key in z.files:
        if not key.startswith(("mask_", "status_")):
            continue
        kind, m, w, k, mo = key.split("_")
        p = oracle.make_params(int(m), int(w), float(k), int(mo))
        if kind == "status":
            with pytest.raises(oracle.OracleError):
                oracle.binarize(gray, p)
            continue
        shape = tuple(z["shape_" + key[5:]])
        want = np.unpackbits(z[key], axis=1)[:, :shape[1]].astype(np.uint8) * 255
        assert np.array_equal(oracle.binarize(gray, p), want), key
        assert np.array_equal(nm.binarize(gray, int(m), int(w), float(k), int(mo)), want), key
        n += 1
    assert n >= 8


def test_golden_nlm_fixture(oracle):
    z = np.load(os.path.join(GOLDEN, "nlm_0050_crop.npz"))
    lab = oracle.lbgr2lab(z["noisy"])
    assert np.array_equal(lab, z["lab"])
    assert np.array_equal(oracle.nlm_planes(np.ascontiguousarray(lab[:, :, 0]), 10.0), z["l_h10"])
    assert np.array_equal(oracle.nlm_planes(np.ascontiguousarray(lab[:, :, 1:]), 3.0), z["ab_h3"])
    assert np.array_equal(oracle.denoise(z["noisy"], 10.0), z["denoised_s10"])


def test_thinning_known_shapes(oracle):
    # a 1-pixel-wide line is already thin; a filled rectangle collapses to a thin set; borders never change
    line = np.zeros((20, 30), np.uint8)
    line[10, 3:27] = 255
    for m in (0, 1):
        assert np.array_equal(oracle.thin(line, m), line)
    rect = np.zeros((40, 60), np.uint8)
    rect[10:30, 15:45] = 255
    for m in (0, 1):
        out, passes = oracle.thin(rect, m, return_passes=True)
        assert 0 < (out > 0).sum() < 60 and passes > 5
        assert set(np.unique(out).tolist()) <= {0, 255}
    border = np.full((12, 12), 255, np.uint8)
    out = oracle.thin(border, 0)
    assert out[0].min() == 255 and out[-1].min() == 255 and out[:, 0].min() == 255 and out[:, -1].min() == 255
    odd = np.array([[0, 1, 2, 3], [254, 255, 7, 8]], np.uint8)       # `&= 1`: only bit 0 is foreground
    assert np.array_equal(oracle.thin(odd, 0), (odd & 1) * 255)
