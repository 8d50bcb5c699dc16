"""Deterministic synthetic inputs (no datasets are reachable from the build or
GPU boxes).  ``cine_phantom`` is a crude short-axis-like image: bright blood
pools inside a darker ring on a textured background, in [0, 1] like the output
of ``rescale_intensity`` (reference ``common/image_utils.py:70-77``)."""
import numpy as np


def cine_phantom(n, h, w, seed=0):
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing='ij')
    out = np.empty((n, h, w, 1), dtype=np.float32)
    for i in range(n):
        img = 0.15 + 0.10 * np.sin(xx / (7.0 + i % 5)) * np.cos(yy / (5.0 + i % 3))
        img += 0.25 * (yy / h) * (xx / w)
        for _ in range(6):
            cy, cx = rng.uniform(0.15, 0.85) * h, rng.uniform(0.15, 0.85) * w
            ry, rx = rng.uniform(0.04, 0.22) * h, rng.uniform(0.04, 0.22) * w
            d = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2
            amp = rng.uniform(0.2, 0.8)
            img += amp * (d < 1.0) - 0.5 * amp * ((d >= 1.0) & (d < 1.7))
        img += 0.03 * rng.standard_normal((h, w)).astype(np.float32)
        out[i, :, :, 0] = np.clip(img, 0.0, 1.0)
    return out


def uniform_slices(n, h, w, seed=1):
    """SURVEY.md 8(d) config 2 input: default_rng(seed).random(float32) in [0,1)."""
    return np.random.default_rng(seed).random((n, h, w, 1), dtype=np.float32)
