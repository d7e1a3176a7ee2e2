#!/usr/bin/env python3
"""Checks the restatement of torch's CPU float32 `sum()` over 256 elements (oracle._torch_sum_f32, the same order as
torch_sum_256 in csrc/histmatch.hip) against torch itself on random normalised histograms.  CPU only."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import stain_oracle as so  # noqa: E402

rng = np.random.default_rng(0)
bad = 0
for _ in range(2000):
    counts = rng.integers(0, int(rng.integers(2, 5000)), 256).astype(np.float32)
    x = counts / (np.float32(counts.sum(dtype=np.float64)) + np.float32(1e-8))
    for v in (x, counts):
        bad += int(np.float32(torch.from_numpy(v).sum().item()) != so._torch_sum_f32(v))
print(f"torch {torch.__version__} ({torch.backends.cpu.get_cpu_capability()}): {bad} mismatches in 4000 sums")
sys.exit(1 if bad else 0)
