"""GPU parity of prl::deskew / prl::rotate / HoughLinesP against the CPU oracle: the segments,
the angle and every output byte are identical."""
import numpy as np
import pytest

from prlib_amd import synth

pytestmark = pytest.mark.gpu


def test_houghp_segments_equal_oracle(prl, oracle, cuda_device):
    import torch

    img = np.zeros((200, 300), np.uint8)
    img[50, 20:280] = 255
    img[20:190, 150] = 255
    rng = np.random.default_rng(1)
    img[rng.integers(0, 200, 900), rng.integers(0, 300, 900)] = 128       # clutter: any non-zero value is a point
    for thr, ll, gap in ((100, 100, 5), (60, 40, 3), (1000, 10, 2)):
        got = prl.houghp(torch.from_numpy(img).to(cuda_device), thr, ll, gap)
        want = oracle.houghp(img, thr, ll, gap)
        assert np.array_equal(got, want), (thr, ll, gap, len(got), len(want))
    # a text page the way findAngle sees it
    p = synth.text_page_numpy(300, 420, 5, skew_deg=2.0)
    thr, binary = oracle.otsu(p)
    inv = 255 - binary
    got = prl.houghp(torch.from_numpy(inv).to(cuda_device), 100, 52, 20)
    want = oracle.houghp(inv, 100, 52, 20)
    assert len(want) > 20 and np.array_equal(got, want)
    assert len(prl.houghp(torch.zeros((40, 50), dtype=torch.uint8, device=cuda_device), 10, 10, 2)) == 0


def test_find_angle_equals_oracle(prl, oracle, cuda_device):
    """prl::findAngle (deskew.h:62, deskew.cpp:139-205) as an entry point of its own: the thresholded page prl::deskew hands
    it (:224-226), a batch with per-page angles, a strided view, a page without points, and the cv::Mat wrapper's host form."""
    import ctypes as C

    import torch
    from prlib_amd import _capi

    pages, want, nseg = [], [], []
    for i, skew in enumerate((2.0, -3.5, 0.0, 7.0)):
        p = synth.text_page_numpy(300, 420, 20 + i, skew_deg=skew)
        _, binary = oracle.otsu(p)
        a, n = oracle.find_angle(binary)
        pages.append(binary); want.append(a); nseg.append(n)
    assert sum(a != 0.0 for a in want) >= 2 and min(nseg) > 5
    t = torch.from_numpy(np.stack(pages)).to(cuda_device)
    got, got_n = prl.findAngle(t, return_segments=True)
    assert got.tolist() == want and got_n.tolist() == nseg          # exact: same segments, same host atan2 / vote
    assert prl.findAngle(t[1]) == want[1]
    # rows 500 bytes apart (a cv::Mat ROI)
    wide = torch.full((4, 300, 500), 255, dtype=torch.uint8, device=cuda_device)
    wide[:, :, 37:457] = t
    assert prl.findAngle(wide[:, :, 37:457]).tolist() == want
    # no point at all (every pixel 255) -> no segment -> 0.0 (deskew.cpp:154-157); a gray page: points are the pixels != 255
    assert prl.findAngle(torch.full((120, 200), 255, dtype=torch.uint8, device=cuda_device)) == 0.0
    g = synth.text_page_numpy(200, 260, 3, skew_deg=1.0)
    assert prl.findAngle(torch.from_numpy(g).to(cuda_device)) == oracle.find_angle(g)[0]
    assert np.array_equal(prl.findOrientation(t), np.zeros(4))
    # host form
    ang, n = C.c_double(-1.0), C.c_int32(-1)
    b = np.ascontiguousarray(pages[0])
    _capi.check(_capi.lib().prl_hip_find_angle_host(b.ctypes.data, b.strides[0], 420, 300, C.byref(ang), C.byref(n)))
    assert ang.value == want[0] and n.value == nseg[0]
    assert _capi.lib().prl_hip_find_angle_host(b.ctypes.data, 100, 420, 300, C.byref(ang), None) == _capi.PRL_ERR_BAD_ARG
    assert _capi.lib().prl_hip_find_angle_host(None, 420, 420, 300, C.byref(ang), None) == _capi.PRL_ERR_EMPTY


@pytest.mark.parametrize("c", [1, 3, 4])
@pytest.mark.parametrize("angle", [0.0, 3.0, -7.25, 45.0, 90.0, 180.0, 270.0, 450.0, 359.999, 1e-9, 123.456])
def test_rotate_matches_oracle(prl, oracle, cuda_device, c, angle):
    import torch

    rng = np.random.default_rng(int(abs(angle) * 10) + c)
    g = synth.text_page_numpy(97, 141, 2, skew_deg=1.0)
    img = g if c == 1 else np.stack([g] + [rng.integers(0, 256, g.shape, dtype=np.uint8) for _ in range(c - 1)], axis=-1)
    t = torch.from_numpy(np.stack([img, img[::-1].copy()])).to(cuda_device)
    outs = prl.rotate(t, angle)
    for i, src in enumerate((img, img[::-1].copy())):
        want = oracle.rotate(src, angle)
        got = outs[i].cpu().numpy()
        assert got.shape == want.shape and np.array_equal(got, want), f"{(got != want).sum()} bytes differ"


def test_rotate_tall_page_and_per_page_angles(prl, oracle, cuda_device):
    import torch

    g = synth.text_page_numpy(180, 75, 4)
    t = torch.from_numpy(np.stack([g, g, g])).to(cuda_device)
    angles = [2.0, 90.0, -33.0]
    outs = prl.rotate(t, angles)
    for i, a in enumerate(angles):
        assert np.array_equal(outs[i].cpu().numpy(), oracle.rotate(g, a))


@pytest.mark.parametrize("c", [1, 3])
def test_deskew_matches_oracle(prl, oracle, cuda_device, c):
    import torch

    pages = []
    for i, skew in enumerate((2.5, -4.0, 0.0, 1.0)):
        g = synth.text_page_numpy(390, 300, 20 + i, skew_deg=skew, shading=0.2)
        pages.append(g if c == 1 else np.stack([g, np.clip(g.astype(int) + 8, 0, 255).astype(np.uint8), g], axis=-1))
    blank = np.full_like(pages[0], 230)          # no segments -> angle 0 -> clone
    pages.append(blank)
    batch = np.stack(pages)
    outs, angles = prl.deskew(torch.from_numpy(batch).to(cuda_device))
    for i in range(len(pages)):
        want, info = oracle.deskew(batch[i])
        assert angles[i] == info["angle"], (i, angles[i], info)
        got = outs[i].cpu().numpy()
        assert got.shape == want.shape and np.array_equal(got, want), f"page {i}: {(got != want).sum()} bytes differ"
    assert angles[-1] == 0.0 and outs[-1].shape[:2] == (390, 300)


_CHILD = r'''
import numpy as np, torch, sys
sys.path.insert(0, %r)
import prlib_amd
prlib_amd._capi.use_library(prlib_amd._capi.HOOKS_LIB_PATH)   # the build that reads the PRL_HIP_* tuning knobs
from prlib_amd import synth
from oracle import capi as oc
bad = 0
img = np.zeros((200, 300), np.uint8); img[50, 20:280] = 255; img[20:190, 150] = 255
rng = np.random.default_rng(1); img[rng.integers(0, 200, 900), rng.integers(0, 300, 900)] = 128
for thr, ll, gap in ((100, 100, 5), (60, 40, 3)):
    bad += not np.array_equal(prlib_amd.houghp(torch.from_numpy(img).cuda(), thr, ll, gap), oc.houghp(img, thr, ll, gap))
pages = np.stack([synth.text_page_numpy(390, 300, 20 + i, skew_deg=s, shading=0.2) for i, s in enumerate((2.5, -4.0, 0.0, 1.0))])
outs, angles = prlib_amd.deskew(torch.from_numpy(pages).cuda())
for i in range(len(pages)):
    want, info = oc.deskew(pages[i])
    bad += angles[i] != info["angle"] or not np.array_equal(outs[i].cpu().numpy(), want)
big = synth.text_page_numpy(1200, 900, 4, skew_deg=-2.5)       # large enough for a group of several workgroups
_, binary = oc.otsu(big)
bad += not np.array_equal(prlib_amd.houghp(torch.from_numpy(255 - binary).cuda(), 100, 112, 20), oc.houghp(255 - binary, 100, 112, 20))
print("BAD", bad)
'''


def _run_child(extra_env):
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PRL_HIP_DEBUG="1", **extra_env)
    r = subprocess.run([sys.executable, "-c", _CHILD % root], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "BAD 0" in r.stdout, r.stdout + r.stderr[-3000:]
    return r.stderr


def test_single_wavefront_kernel_gives_the_same_segments():
    """k_ppht (one wavefront per page: PRL_HIP_PPHT_GROUP=0 PRL_HIP_PPHT_MW=0, read once per process, hence the child) against the
    oracle on the cases above."""
    err = _run_child({"PRL_HIP_PPHT_GROUP": "0", "PRL_HIP_PPHT_MW": "0"})
    assert "group kernel" not in err


def test_three_wavefront_kernel_gives_the_same_segments():
    """k_ppht_mw (the accumulator in device memory, three wavefronts per page: what takes the pages the group kernel leaves)."""
    err = _run_child({"PRL_HIP_PPHT_GROUP": "0"})
    assert "group kernel" not in err


def test_group_kernel_is_the_default_and_takes_every_page():
    err = _run_child({})
    assert "group kernel" in err and all(" 0 pages left to k_ppht_mw" in ln for ln in err.splitlines() if "pages left" in ln), err[-2000:]


def test_group_that_gives_up_is_redone_in_the_same_call():
    """A member that falls silent (hooks build: member 1 of group 0 after 40 exchanges): the others run out of patience (20 ms here),
    raise the group's abort word and leave; the pages not marked done go through k_ppht_mw in the same call - same segments."""
    err = _run_child({"PRL_HIP_PPHT_GROUP_KILL": "40", "PRL_HIP_PPHT_GROUP_SPIN_MS": "20"})
    left = [int(ln.split(";")[1].split()[0]) for ln in err.splitlines() if "pages left to k_ppht_mw" in ln]
    assert left and max(left) >= 1, err[-2000:]


def test_deskew_argument_errors(prl, cuda_device):
    import torch

    from prlib_amd import _capi
    L = _capi.lib()
    t = torch.zeros((64, 64), dtype=torch.uint8, device=cuda_device)
    o = torch.zeros((64, 64), dtype=torch.uint8, device=cuda_device)
    wh = np.zeros(2, np.int32)
    assert L.prl_hip_deskew_batch_device(1, 1, t.data_ptr(), 4096, 64, 0, 64, o.data_ptr(), 4096, 64, wh.ctypes.data, None, None) == _capi.PRL_ERR_EMPTY
    assert L.prl_hip_deskew_batch_device(1, 2, t.data_ptr(), 4096, 128, 64, 64, o.data_ptr(), 4096, 128, wh.ctypes.data, None, None) == _capi.PRL_ERR_BAD_CHANNELS
    assert L.prl_hip_deskew_batch_device(1, 1, t.data_ptr(), 4096, 64, 64, 64, o.data_ptr(), 4096, 32, wh.ctypes.data, None, None) == _capi.PRL_ERR_BAD_ARG
