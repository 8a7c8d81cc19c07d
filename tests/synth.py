"""Synthetic frame generator of BASELINE/SURVEY Appendix C (the recipe the baseline numbers use)."""
import numpy as np


def gen(w, h, seed, nrect):
    rng = np.random.default_rng(seed)
    coarse = rng.uniform(60, 190, size=(h // 40 + 2, w // 40 + 2))
    img = np.kron(coarse, np.ones((40, 40)))[:h, :w]
    for _ in range(nrect):
        rw, rh = rng.integers(6, 60, 2)
        x = rng.integers(0, w - rw)
        y = rng.integers(0, h - rh)
        img[y:y + rh, x:x + rw] = rng.uniform(0, 255)
    p = np.pad(img, 1, mode="edge")
    img = sum(p[dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3)) / 9.0
    img += rng.normal(0, 2.0, img.shape)
    return np.clip(img + 0.5, 0, 255).astype(np.uint8)


def frame_1080p(seed):
    return gen(1920, 1080, seed, 300)


def frame_vga(seed=1):
    return gen(640, 480, seed, 60)


def frame_4k(seed=2):
    return gen(3840, 2160, seed, 1200)
