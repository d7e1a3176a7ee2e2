"""Case lists shared by the golden generator (which imports the reference) and the tests (which must not)."""
from __future__ import annotations

import numpy as np


def g9_cases(seed: int = 3, cases: int = 80):
    """The random small histogram-matching cases of g9 (shared with the tests): (n, h, w, dtype name, src seed, ref seed)."""
    rng = np.random.default_rng(seed)
    names = ["u8", "f16", "f32", "f64"]
    out = []
    for _ in range(cases):
        n = int(rng.integers(1, 5))
        h, w = int(rng.integers(4, 64)), int(rng.integers(4, 64))
        out.append((n, h, w, names[int(rng.integers(0, 4))], int(rng.integers(0, 1 << 20)), int(rng.integers(0, 1 << 20))))
    return out


def real_quadrants_512():
    """g11: (image index in g11_real_images.npz, y, x) of the twenty 512 x 512 quadrants of test_1..5 (image 0 is the target)."""
    return [(i, y, x) for i in range(1, 6) for y in (0, 512) for x in (0, 512)]


def real_crops_224():
    """g11: six 224 x 224 crops (the example's crop size), one per image at assorted offsets -- the target's included."""
    return [(0, 400, 400), (1, 100, 700), (2, 640, 80), (3, 333, 501), (4, 0, 0), (5, 800, 800)]
