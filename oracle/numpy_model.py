"""Independent whole-plane numpy model of PRLib's five local-adaptive binarizers.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED against the real reference (no OpenCV in this image).

Purpose: cross-check oracle/prl_oracle.c.  The C oracle evaluates one pixel at a time; this model
is written the way the reference itself is written — a sequence of whole-plane float64 array
operations, one numpy ufunc per OpenCV call — so a slip in either restatement shows up as a
bit-level disagreement.  numpy ufuncs round once per operation and never fuse multiply-add.

Reference lines (PRLib tree): src/binarizations/binarizeSauvola.cpp:32-134, binarizeNiblack.cpp:32-127,
binarizeWolfJolion.cpp:33-148, binarizeNICK.cpp:33-144, binarizeFeng.cpp:31-164.
"""
from __future__ import annotations

import numpy as np

SAUVOLA, NIBLACK, WOLFJOLION, NICK, FENG = range(5)


class EmptyInput(ValueError):
    pass


class BadWindow(ValueError):
    pass


class EmptyRect(ValueError):
    pass


def geometry(method: int, window: int, width: int, height: int):
    """(w, half, out_w, out_h) — binarizeSauvola.cpp:38-66 / binarizeWolfJolion.cpp:69."""
    if width <= 0 or height <= 0:
        raise EmptyInput("Input image for binarization is empty")
    if not (window > 1 and window % 2 == 1):
        raise BadWindow("Window size must satisfy (windowSize > 1) && ((windowSize % 2) == 1)")
    w = min(window, min(width, height))
    half = w // 2
    if method in (SAUVOLA, NIBLACK):
        ow, oh = width + 2 * half - w, height + 2 * half - w
    else:
        ow, oh = width - w, height - w
    if ow <= 0 or oh <= 0:
        raise EmptyRect("processing rectangle is empty")
    return w, half, ow, oh


def _filter4(t: np.ndarray, w: int, ow: int, oh: int, f: float) -> np.ndarray:
    """cv::filter2D with taps (0,0)+f (0,w-1)-f (w-1,0)-f (w-1,w-1)+f, row-major accumulation."""
    a = t[0:oh, 0:ow]
    b = t[0:oh, w - 1:w - 1 + ow]
    c = t[w - 1:w - 1 + oh, 0:ow]
    d = t[w - 1:w - 1 + oh, w - 1:w - 1 + ow]
    s0 = np.zeros((oh, ow), dtype=np.float64)
    s0 = s0 + f * a
    s0 = s0 + (-f) * b
    s0 = s0 + (-f) * c
    s0 = s0 + f * d
    return s0


def sat_u8(t: np.ndarray) -> np.ndarray:
    """saturate_cast<uchar>(double): NaN/inf/out-of-int32 -> 0, else rint (half-even) and clamp."""
    t = np.asarray(t, dtype=np.float64)
    with np.errstate(invalid="ignore"):
        r = np.rint(t)
        bad = ~np.isfinite(r) | (r < -2147483648.0) | (r > 2147483647.0)
        r = np.where(bad, 0.0, r)
        return np.clip(r, 0.0, 255.0).astype(np.uint8)


def mean_dev(img: np.ndarray, method: int, window: int):
    h_, w_ = img.shape
    w, half, ow, oh = geometry(method, window, w_, h_)
    padded = np.pad(img, half, mode="edge")
    pd = padded.astype(np.float64)
    ii = np.cumsum(np.cumsum(pd, axis=0), axis=1)
    iq = np.cumsum(np.cumsum(pd * pd, axis=0), axis=1)
    f = 1.0 / float(w * w)
    m = _filter4(ii, w, ow, oh, f)
    m2 = m * m
    q = _filter4(iq, w, ow, oh, f)
    q = q - m2
    with np.errstate(invalid="ignore"):
        s = np.sqrt(q)
    return padded, m, s, (w, half, ow, oh)


def threshold_plane(img: np.ndarray, method: int, window: int, k: float = 0.0,
                    alpha1: float = 0.75, k1: float = 0.2, k2: float = 0.03, gamma: float = 2.0):
    padded, m, s, geo = mean_dev(img, method, window)
    with np.errstate(invalid="ignore", divide="ignore"):
        if method == SAUVOLA:
            rback = 1.0 / 128.0
            d = s * (k * rback) + (1.0 - k)
            t = m * d
        elif method == NIBLACK:
            t = s * k + m
        elif method == WOLFJOLION:
            imin = float(padded.min())
            finite = s[~np.isnan(s)]
            smax = float(finite.max()) if finite.size else -np.finfo(np.float64).max
            coeff = np.float64(k) / np.float64(smax)
            d = s * coeff + (-k)
            e = m * 1.0 + (-imin)
            t = m + d * e
        elif method == NICK:
            c = m * m
            c = c + s * s
            c = np.sqrt(c)
            t = (m * 1.0 + c * k) + 0.0
        elif method == FENG:
            imin = float(padded.min())
            r = np.where(s != 0.0, s / np.where(s != 0.0, s, 1.0), 0.0)
            r = np.where(np.isnan(s), np.nan, r)
            r2 = r * r if gamma == 2.0 else np.power(r, gamma)
            a3 = r2 * k2 + 0.0
            c1 = 1.0 - alpha1
            c2 = r2 * r
            c3 = (a3 * imin + c2 * (-imin)) + 0.0
            t = c2 * 1.0 + c1
            t = t * m
            t = t + c3
        else:
            raise ValueError("unknown method")
    return padded, t, geo


def morph(mask: np.ndarray, iterations: int) -> np.ndarray:
    """cv::dilate/erode with 3x3 x n = (2n+1)^2 rectangle, out-of-image ignored."""
    if iterations == 0:
        return mask.copy()
    from scipy import ndimage

    n = abs(iterations)
    size = 2 * n + 1

    def dil(a):
        return ndimage.maximum_filter(a, size=size, mode="constant", cval=0)

    def ero(a):
        return ndimage.minimum_filter(a, size=size, mode="constant", cval=255)

    return ero(dil(mask)) if iterations > 0 else dil(ero(mask))


def binarize(img: np.ndarray, method: int, window: int, k: float = 0.0, morph_iterations: int = 0,
             alpha1: float = 0.75, k1: float = 0.2, k2: float = 0.03, gamma: float = 2.0) -> np.ndarray:
    img = np.ascontiguousarray(img, dtype=np.uint8)
    padded, t, (w, half, ow, oh) = threshold_plane(img, method, window, k, alpha1, k1, k2, gamma)
    t8 = sat_u8(t)
    pix = padded[half:half + oh, half:half + ow]
    out = np.where(pix > t8, 255, 0).astype(np.uint8)
    return morph(out, morph_iterations)


# ---- backgroundNormalization: an independently written whole-plane model of pixBackgroundNormSimple -------------------
# (same published algorithm as prl_oracle_bgnorm.c, different code: scipy morphology, reshape/sum tiles, pandas-free
# forward fill, integral-image box sums in float32 numpy).  Used by tests/test_oracle_stages.py to catch slips in the C.

def bgnorm_model(img: np.ndarray) -> np.ndarray:
    from scipy import ndimage

    a = img if img.ndim == 3 else img[:, :, None]
    h, w, c = a.shape
    och = 1 if c == 1 else 3
    sx, sy = 10, 15
    gray = a[:, :, 0 if c == 1 else 1]
    fg = ndimage.binary_dilation(gray < 60, structure=np.ones((7, 7), bool))  # zero beyond the border
    mw, mh, nx, ny = -(-w // sx), -(-h // sy), w // sx, h // sy
    inv = np.zeros((och, mh, mw), np.int64)
    ok = nx > 0 and ny > 0 and mw >= 5 and mh >= 5
    for ch in range(och):
        if not ok:
            break
        m = np.zeros((mh, mw), np.int64)
        keep = ~fg[: ny * sy, : nx * sx]
        vals = a[: ny * sy, : nx * sx, ch].astype(np.int64) * keep
        cnt = keep.reshape(ny, sy, nx, sx).sum(axis=(1, 3))
        sm = vals.reshape(ny, sy, nx, sx).sum(axis=(1, 3))
        m[:ny, :nx] = np.where(cnt >= 40, sm // np.maximum(cnt, 1), 0)
        has = (m[:ny, :nx] != 0).any(axis=0)
        if not has.any():
            ok = False
            break
        for j in np.nonzero(has)[0]:          # column fill: first value upwards, then carry downwards to the last row
            col = m[:, j]
            first = np.nonzero(col[:ny])[0][0]
            col[:first] = col[first]
            for i in range(1, mh):
                if col[i] == 0:
                    col[i] = col[i - 1]
        good = np.nonzero(has)[0]
        for j in range(mw):                   # columns without data (incl. the column of incomplete tiles)
            if j < nx and has[j]:
                continue
            left = good[good < j]
            m[:, j] = m[:, left[-1]] if len(left) else m[:, good[0]]
        if mw > nx:
            m[:, mw - 1] = m[:, mw - 2]
        # blockconv(2, 1): Leptonica's accumulator differences drop row / column 0 from windows clamped at the top / left
        acc = m.cumsum(0).cumsum(1)
        ii, jj = np.mgrid[0:mh, 0:mw]
        imin, imax = np.maximum(ii - 2, 0), np.minimum(ii + 1, mh - 1)
        jmin, jmax = np.maximum(jj - 3, 0), np.minimum(jj + 2, mw - 1)
        box = acc[imax, jmax] - acc[imax, jmin] - acc[imin, jmax] + acc[imin, jmin]
        norm = np.float32(1.0 / (np.float32(5) * np.float32(3)))
        val = ((norm * box.astype(np.float32)).astype(np.float64) + 0.5).astype(np.int64) & 255
        hn = np.where(ii <= 1, np.maximum(1, 1 + ii), np.where(ii >= mh - 1, 1 + mh - ii, 0))
        wn = np.where(jj <= 2, np.maximum(1, 2 + jj), np.where(jj >= mw - 2, 2 + mw - jj, 0))
        t = val.astype(np.float32)
        t = np.where(hn > 0, t * (np.float32(3) / np.maximum(hn, 1).astype(np.float32)), t).astype(np.float32)
        t = np.where(wn > 0, t * (np.float32(5) / np.maximum(wn, 1).astype(np.float32)), t).astype(np.float32)
        val = np.where((hn > 0) | (wn > 0), np.minimum(t, np.float32(255)).astype(np.int64), val)
        inv[ch] = np.where(val > 0, 51200 // np.maximum(val, 1), 100)
    out = a[:, :, :och].copy()
    if ok:
        ty, tx = np.arange(h) // sy, np.arange(w) // sx
        for ch in range(och):
            out[:, :, ch] = np.minimum(255, (a[:, :, ch].astype(np.int64) * inv[ch][np.ix_(ty, tx)]) >> 8).astype(np.uint8)
    return out if img.ndim == 3 else out[:, :, 0]
