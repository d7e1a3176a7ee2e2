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
