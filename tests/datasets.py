"""Deterministic synthetic inputs (SURVEY.md section 8(d) datasets).

noise        iid uniform u8 per channel, alpha = 255
photo_like   low-frequency sinusoids + multi-octave value noise quantised to u8
             (realistic energy, many exact ties), alpha = 255
flat_blocks  piecewise-constant rectangles (tie stress; parity only)
alpha_ramp   photo_like with alpha = horizontal ramp (exercises x alpha)
"""
import numpy as np


def _rng(seed):
    return np.random.Generator(np.random.PCG64(int(seed)))


def _with_alpha(rgb, channels, alpha=None):
    h, w, _ = rgb.shape
    if channels == 1:
        return rgb[:, :, :1].copy()
    if channels == 3:
        return rgb.copy()
    a = np.full((h, w, 1), 255, np.uint8) if alpha is None else alpha.reshape(h, w, 1).astype(np.uint8)
    if channels == 2:
        return np.concatenate([rgb[:, :, :1], a], axis=2)
    return np.concatenate([rgb, a], axis=2)


def noise(w, h, seed, channels=4):
    rgb = _rng(seed).integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    return np.ascontiguousarray(_with_alpha(rgb, channels))


def _upsample_bilinear(g, h, w):
    gh, gw = g.shape
    ys = np.linspace(0, gh - 1, h, dtype=np.float32)
    xs = np.linspace(0, gw - 1, w, dtype=np.float32)
    y0 = np.minimum(ys.astype(np.int32), gh - 2)
    x0 = np.minimum(xs.astype(np.int32), gw - 2)
    fy = (ys - y0)[:, None]
    fx = (xs - x0)[None, :]
    rows0 = g[y0][:, x0] * (1 - fx) + g[y0][:, x0 + 1] * fx
    rows1 = g[y0 + 1][:, x0] * (1 - fx) + g[y0 + 1][:, x0 + 1] * fx
    return rows0 * (1 - fy) + rows1 * fy


def photo_like(w, h, seed, channels=4, alpha=None):
    rng = _rng(seed)
    yy = np.arange(h, dtype=np.float32)[:, None]
    xx = np.arange(w, dtype=np.float32)[None, :]
    out = np.zeros((h, w, 3), np.float32)
    base = np.zeros((h, w), np.float32)
    for _ in range(4):      # shared low-frequency structure
        fx, fy = rng.uniform(0.5, 4.0, 2) * 2 * np.pi
        ph = rng.uniform(0, 2 * np.pi)
        base += rng.uniform(0.3, 1.0) * np.sin(fx * xx / w + fy * yy / h + ph)
    octaves = []
    n = 4
    while n < max(w, h):
        octaves.append(n)
        n *= 2
    for c in range(3):
        acc = base.copy()
        for i, n in enumerate(octaves):
            gh, gw = max(2, n * h // max(w, h) + 1), max(2, n * w // max(w, h) + 1)
            g = rng.standard_normal((gh, gw), dtype=np.float32)
            acc += _upsample_bilinear(g, h, w) * (0.9 / (i + 1))     # ~1/f amplitude
        out[:, :, c] = acc
    out -= out.min()
    out *= 255.0 / max(out.max(), 1e-6)
    rgb = np.clip(np.rint(out), 0, 255).astype(np.uint8)
    return np.ascontiguousarray(_with_alpha(rgb, channels, alpha))


def flat_blocks(w, h, seed, channels=4, nblocks=12):
    rng = _rng(seed)
    rgb = np.zeros((h, w, 3), np.uint8)
    rgb[:] = rng.integers(0, 256, 3, dtype=np.uint8)
    for _ in range(nblocks):
        x0, x1 = sorted(rng.integers(0, w + 1, 2))
        y0, y1 = sorted(rng.integers(0, h + 1, 2))
        rgb[y0:y1, x0:x1] = rng.integers(0, 256, 3, dtype=np.uint8)
    return np.ascontiguousarray(_with_alpha(rgb, channels))


def alpha_ramp(w, h, seed, channels=4):
    a = np.broadcast_to(np.linspace(0, 255, w).astype(np.uint8)[None, :], (h, w))
    return photo_like(w, h, seed, channels, alpha=np.ascontiguousarray(a))


def ellipse_mask(w, h, area_frac=0.25, channels=4):
    """filled ellipse at the centre covering ~area_frac of the image, white, alpha=255 inside"""
    yy = (np.arange(h, dtype=np.float32)[:, None] - (h - 1) / 2) / (h / 2)
    xx = (np.arange(w, dtype=np.float32)[None, :] - (w - 1) / 2) / (w / 2)
    r2 = area_frac * 4 / np.pi
    inside = (xx * xx + yy * yy) <= r2
    m = np.zeros((h, w, channels), np.uint8)
    m[inside] = 255
    return m


def band_mask(w, h, x0, x1, channels=4):
    m = np.zeros((h, w, channels), np.uint8)
    m[:, x0:x1] = 255
    return m


def top_half_mask(w, h, channels=4):
    m = np.zeros((h, w, channels), np.uint8)
    m[: h // 2] = 255
    return m


DATASETS = dict(noise=noise, photo_like=photo_like, flat_blocks=flat_blocks, alpha_ramp=alpha_ramp)
