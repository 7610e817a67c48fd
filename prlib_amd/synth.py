"""Seeded synthetic document pages: u8 grayscale "paper" N(215,12) with dark stroke
rectangles.  Two generators with the same statistics:
  page_numpy  — the specified generator (numpy PCG64, seed = 1000 + page index); small parity cases
  pages_torch — the same recipe drawn on the device with torch's generator, for full-size batches
                (256 x 4096^2 would take minutes on host cores).  Parity runs download these pages.
"""
from __future__ import annotations

import numpy as np


def page_numpy(height: int, width: int, index: int = 0) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(1000 + index))
    page = rng.normal(215.0, 12.0, size=(height, width))
    n_strokes = max(1, (height * width) // 900)
    ys = rng.integers(0, height, n_strokes)
    xs = rng.integers(0, width, n_strokes)
    hs = rng.integers(2, 6, n_strokes)
    ws = rng.integers(5, 40, n_strokes)
    ds = rng.integers(90, 170, n_strokes)
    for y, x, h, w, d in zip(ys, xs, hs, ws, ds):
        page[y:y + h, x:x + w] -= d
    return np.clip(np.rint(page), 0, 255).astype(np.uint8)


def pages_torch(n_pages: int, height: int, width: int, device, seed: int = 1000, pitch: int | None = None):
    """N x H x W uint8 pages on `device` (a view of an N x H x pitch buffer when pitch is given)."""
    import torch

    pitch = pitch or width
    buf = torch.empty((n_pages, height, pitch), dtype=torch.uint8, device=device)
    gen = torch.Generator(device=device)
    n_strokes = max(1, (height * width) // 900)
    for i in range(n_pages):
        gen.manual_seed(seed + i)
        page = torch.empty((height, width), dtype=torch.float32, device=device).normal_(215.0, 12.0, generator=gen)
        # strokes: subtract darkness over random small rectangles via a coarse scatter + box spread
        ys = torch.randint(0, height, (n_strokes,), device=device, generator=gen)
        xs = torch.randint(0, width, (n_strokes,), device=device, generator=gen)
        hs = torch.randint(2, 6, (n_strokes,), device=device, generator=gen)
        ws = torch.randint(5, 40, (n_strokes,), device=device, generator=gen)
        ds = torch.randint(90, 170, (n_strokes,), device=device, generator=gen).to(torch.float32)
        # 2-D difference array: +d at (y,x), -d at (y,x+w), -d at (y+h,x), +d at (y+h,x+w); cumsum twice
        diff = torch.zeros((height + 8, width + 48), dtype=torch.float32, device=device)
        flat = diff.view(-1)
        wd = width + 48
        flat.index_add_(0, ys * wd + xs, ds)
        flat.index_add_(0, ys * wd + xs + ws, -ds)
        flat.index_add_(0, (ys + hs) * wd + xs, -ds)
        flat.index_add_(0, (ys + hs) * wd + xs + ws, ds)
        dark = diff.cumsum(0).cumsum(1)[:height, :width]
        page = (page - dark).round_().clamp_(0, 255)
        buf[i, :, :width] = page.to(torch.uint8)
    return buf[:, :, :width]


def text_page_numpy(height: int, width: int, index: int = 0, skew_deg: float = 0.0, shading: float = 0.0) -> np.ndarray:
    """A page of "text lines" (rows of word-like dark runs) drawn at `skew_deg`, on paper whose brightness falls off by
    `shading` (0..1) towards one corner - the input deskew and backgroundNormalization are written for.  Seeded."""
    rng = np.random.Generator(np.random.PCG64(5000 + index))
    a = np.deg2rad(skew_deg)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float64)
    u = xx * np.cos(a) + yy * np.sin(a)
    v = -xx * np.sin(a) + yy * np.cos(a)
    pitch, line_h = 34.0, 9.0
    margin = 0.06 * min(height, width)
    span = int(np.hypot(height, width)) + 64
    # word pattern along a line: 1 = ink; words of 20-90 px separated by 10-22 px, different on every line
    n_lines = int(np.hypot(height, width) // pitch) + 4
    pat = np.zeros((2 * n_lines, 2 * span), dtype=bool)
    for ln in range(2 * n_lines):
        x = int(rng.integers(0, 30))
        while x < 2 * span:
            wl = int(rng.integers(20, 90))
            pat[ln, x:x + wl] = True
            x += wl + int(rng.integers(10, 22))
    li = np.floor(v / pitch).astype(np.int64)
    inside = (np.mod(v, pitch) < line_h) & (xx > margin) & (xx < width - margin) & (yy > margin) & (yy < height - margin)
    ink = inside & pat[np.clip(li + n_lines, 0, 2 * n_lines - 1), np.clip(u.astype(np.int64) + span, 0, 2 * span - 1)]
    paper = rng.normal(225.0, 6.0, size=(height, width))
    light = 1.0 - shading * (0.6 * xx / max(1, width - 1) + 0.4 * yy / max(1, height - 1))
    page = np.where(ink, rng.normal(45.0, 10.0, size=(height, width)), paper) * light
    return np.clip(np.rint(page), 0, 255).astype(np.uint8)


def text_pages_torch(n_pages: int, height: int, width: int, device, seed: int = 7000, max_skew_deg: float = 4.0,
                     shading: float = 0.3, channels: int = 1):
    """text_page_numpy's recipe drawn on the device (full-size chain batches): N x H x W [x C] uint8, one skew per page
    uniform in +-max_skew_deg (returned as well)."""
    import torch

    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    shape = (n_pages, height, width) if channels == 1 else (n_pages, height, width, channels)
    out = torch.empty(shape, dtype=torch.uint8, device=device)
    skews = (torch.rand(n_pages, generator=gen, device=device) * 2 - 1) * max_skew_deg
    yy = torch.arange(height, device=device, dtype=torch.float32)[:, None]
    xx = torch.arange(width, device=device, dtype=torch.float32)[None, :]
    pitch, line_h = 40.0, 6.0
    margin = 0.06 * min(height, width)
    span = int((height ** 2 + width ** 2) ** 0.5) + 64
    n_lines = span // int(pitch) + 4
    for i in range(n_pages):
        a = torch.deg2rad(skews[i])
        u = xx * torch.cos(a) + yy * torch.sin(a)
        v = -xx * torch.sin(a) + yy * torch.cos(a)
        # word pattern: per line a random phase and period; ink where (u + phase) mod period < 0.8 period
        phase = torch.rand(2 * n_lines, generator=gen, device=device) * 90
        period = 45 + torch.rand(2 * n_lines, generator=gen, device=device) * 60
        li = torch.clamp(torch.floor(v / pitch).long() + n_lines, 0, 2 * n_lines - 1)
        inside = (torch.remainder(v, pitch) < line_h) & (xx > margin) & (xx < width - margin) & (yy > margin) & (yy < height - margin)
        ink = inside & (torch.remainder(u + span + phase[li], period[li]) < 0.8 * period[li])
        light = 1.0 - shading * (0.6 * xx / max(1, width - 1) + 0.4 * yy / max(1, height - 1))
        for c in range(channels):
            paper = torch.empty((height, width), dtype=torch.float32, device=device).normal_(225.0, 6.0, generator=gen)
            dark = torch.empty((height, width), dtype=torch.float32, device=device).normal_(45.0, 10.0, generator=gen)
            page = (torch.where(ink, dark, paper) * light).round_().clamp_(0, 255).to(torch.uint8)
            if channels == 1:
                out[i] = page
            else:
                out[i, :, :, c] = page
    return out, skews.cpu().numpy()
