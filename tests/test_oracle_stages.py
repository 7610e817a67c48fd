"""CPU checks of the round-2 oracle additions (test infrastructure): backgroundNormalization against an independently
written numpy model and hand-derived answers; HoughLinesP / findAngle / rotate against properties OpenCV's definitions
imply (the reference ships no vectors for these stages: parity unpinned, see oracle/prl_oracle_*.c headers)."""
import numpy as np
import pytest

from prlib_amd import synth


def _pages():
    yield synth.page_numpy(203, 317, index=11)
    yield synth.text_page_numpy(330, 250, 3, skew_deg=1.5, shading=0.6)
    p = synth.page_numpy(150, 200, index=5)
    p[30:110, 40:160] = 20          # a block of foreground: tiles without background, holes to fill
    yield p
    p = synth.page_numpy(160, 120, index=6)
    p[:, :35] = 10                  # whole tile columns without data on the left
    p[:, 95:] = 0                   # ... and on the right
    yield p
    yield np.full((90, 70), 100, np.uint8)


def test_bgnorm_oracle_matches_numpy_model(oracle):
    from oracle import numpy_model

    for p in _pages():
        assert np.array_equal(oracle.bgnorm(p), numpy_model.bgnorm_model(p))
    rng = np.random.default_rng(0)
    col = rng.integers(0, 256, (95, 123, 3), dtype=np.uint8)
    assert np.array_equal(oracle.bgnorm(col), numpy_model.bgnorm_model(col))
    col4 = rng.integers(40, 256, (77, 91, 4), dtype=np.uint8)
    got = oracle.bgnorm(col4)
    assert got.shape == (77, 91, 3) and np.array_equal(got, numpy_model.bgnorm_model(col4))


def test_bgnorm_known_answers(oracle):
    # flat page p: every tile averages p, the smoothed map is p away from the map border, inverse 51200 // p,
    # result (p * (51200 // p)) >> 8: 100 -> 200, 200 -> 200, 255 -> 200 (integer truncation of 51200/255 = 200)
    for p, want in ((100, 200), (200, 200), (255, 199), (60, 199)):
        out = oracle.bgnorm(np.full((150, 120), p, np.uint8))
        assert out[45:105, 40:80].min() == out[45:105, 40:80].max() == (p * (51200 // p)) >> 8 == want
    # a page without any background (everything below the foreground threshold): "map not made" -> copy of the source
    dark = np.full((90, 80), 30, np.uint8)
    assert np.array_equal(oracle.bgnorm(dark), dark)
    # smaller than 5 x 5 tiles, or no complete tile at all: copy
    small = synth.page_numpy(60, 49, index=1)
    assert np.array_equal(oracle.bgnorm(small), small)
    tiny = synth.page_numpy(14, 9, index=1)
    assert np.array_equal(oracle.bgnorm(tiny), tiny)
    # saturation: a bright pixel over a dark background estimate clamps at 255
    page = np.full((150, 120), 80, np.uint8)
    page[70, 60] = 250
    assert oracle.bgnorm(page)[70, 60] == 255


def test_bgnorm_channel_conventions(oracle):
    # formatConvert.cpp: 3-channel pages are processed per channel with the foreground mask from channel 1 (green slot);
    # the 4th channel of a 4-channel page is dropped
    rng = np.random.default_rng(4)
    g = synth.text_page_numpy(120, 100, 2, shading=0.4)
    col = np.stack([g, g, g], axis=-1)
    out = oracle.bgnorm(col)
    one = oracle.bgnorm(g)
    for c in range(3):
        assert np.array_equal(out[:, :, c], one)
    col4 = np.concatenate([col, rng.integers(0, 256, g.shape + (1,), dtype=np.uint8)], axis=-1)
    assert np.array_equal(oracle.bgnorm(col4), out)
    # the mask comes from channel 1 only: darkening channel 0 alone must not change channel 2's result
    c2 = col.copy()
    c2[40:60, 30:50, 0] = 5
    assert np.array_equal(oracle.bgnorm(c2)[:, :, 2], out[:, :, 2])


def test_otsu_and_rotate_special_angles(oracle):
    p = synth.text_page_numpy(60, 90, 1)
    assert np.array_equal(oracle.rotate(p, 90.0), np.rot90(p, -1))     # transpose + flip around y = clockwise quarter turn
    assert np.array_equal(oracle.rotate(p, 180.0), p[::-1, ::-1])
    assert np.array_equal(oracle.rotate(p, 270.0), np.rot90(p, 1))
    assert np.array_equal(oracle.rotate(p, 450.0), np.rot90(p, -1))    # fmod(angle, 360)
    col = np.stack([p, 255 - p, p // 2], axis=-1)
    assert np.array_equal(oracle.rotate(col, 180.0), col[::-1, ::-1])


def test_rotate_general_properties(oracle):
    p = synth.text_page_numpy(80, 120, 2)
    # angle 0 goes through warpAffine with the identity: a 120 x 120 canvas, the page in the top-left corner, white border
    # (the border value 0 of the inverted image)
    out = oracle.rotate(p, 0.0)
    assert out.shape == (120, 120)
    assert np.array_equal(out[:80, :120], p) and (out[80:] == 255).all()
    # a full turn is the identity as well (cos/sin of 2 pi differ from 1/0 by ~1e-16, far below the 2^-10 grid)
    assert np.array_equal(oracle.rotate(p, 360.0 - 1e-9)[:80], p)
    # small rotations keep the ink mass within a few percent and produce a square canvas
    r = oracle.rotate(p, 3.0)
    assert r.shape == (120, 120)
    ink = lambda a: float((255 - a.astype(np.int64)).sum())
    assert abs(ink(r) - ink(p)) / ink(p) < 0.05
    # three channels rotate independently and identically
    col = np.stack([p, p, p], axis=-1)
    rc = oracle.rotate(col, 3.0)
    assert all(np.array_equal(rc[:, :, c], r) for c in range(3))
    # matrix: getRotationMatrix2D about (60, 60), inverted: the centre maps to itself
    m = oracle.rotate_matrix(120, 80, 17.0)
    assert abs(m[0] * 60 + m[1] * 60 + m[2] - 60) < 1e-9 and abs(m[3] * 60 + m[4] * 60 + m[5] - 60) < 1e-9


def test_houghp_finds_drawn_segments(oracle):
    img = np.zeros((200, 300), np.uint8)
    img[50, 20:280] = 255                      # one horizontal segment of 260 px
    img[20:190, 150] = 255                     # one vertical segment of 170 px
    lines = oracle.houghp(img, 100, 100, 5)
    assert len(lines) >= 2
    horiz = [l for l in lines if l[1] == l[3] == 50]
    vert = [l for l in lines if l[0] == l[2] == 150]
    assert horiz and vert
    assert max(abs(int(l[2]) - int(l[0])) for l in horiz) >= 250
    assert max(abs(int(l[3]) - int(l[1])) for l in vert) >= 160
    # nothing above the vote threshold -> no segments; an empty image -> no segments
    assert len(oracle.houghp(img, 1000, 100, 5)) == 0
    assert len(oracle.houghp(np.zeros((50, 60), np.uint8), 10, 10, 2)) == 0
    # deterministic (cv::RNG is seeded with a constant)
    assert np.array_equal(lines, oracle.houghp(img, 100, 100, 5))


@pytest.mark.parametrize("skew", [-4.0, -1.0, 2.5])
def test_deskew_oracle_recovers_the_skew(oracle, skew):
    p = synth.text_page_numpy(520, 400, 7, skew_deg=skew)
    out, info = oracle.deskew(p)
    assert info["n_lines"] > 20 and abs(info["angle"] - skew) < 0.7     # clusters are 0.01 rad = 0.57 degrees wide
    assert out.shape == (520, 520)
    # the vote is what deskew.cpp:158-201 says: first-fit clusters of atan2, the most populated one's first angle
    thr, binary = oracle.otsu(p)
    lines = oracle.houghp(255 - binary, 100, int(np.rint(np.float32(400) / np.float32(8))), 20)
    assert len(lines) == info["n_lines"] and oracle.vote_angle(lines) == info["angle"]
    # a blank page has no segments: angle 0, output = input
    blank = np.full((100, 80), 230, np.uint8)
    out, info = oracle.deskew(blank)
    assert info["angle"] == 0.0 and np.array_equal(out, blank)
