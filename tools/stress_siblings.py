#!/usr/bin/env python3
"""Randomised parity sweep of Reinhard and histogram matching against the CPU oracle (run on a GPU box after kernel
changes; not part of the test suite)."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import stain_oracle as so  # noqa: E402
from stainx_amd import synth  # noqa: E402
from stainx_amd.backends.torch_hip_backend import HistogramMatchingHIP, ReinhardHIP  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dtypes = [torch.uint8, torch.float16, torch.float32, torch.float64]
bad = 0
for case in range(cases):
    n = int(rng.integers(1, 5))
    h, w = int(rng.integers(4, 200)), int(rng.integers(4, 200))
    dt = dtypes[int(rng.integers(0, len(dtypes)))]
    src_u8 = synth.noise_u8((n, 3, h, w), int(rng.integers(0, 1 << 20)))
    ref_u8 = synth.noise_u8((1, 3, h, w), int(rng.integers(0, 1 << 20)))
    x, ref = synth.as_dtype(src_u8, dt), synth.as_dtype(ref_u8, dt)
    # histogram matching, both layouts
    last = bool(rng.integers(0, 2))
    hb = HistogramMatchingHIP(dev, channel_axis=-1 if last else 1)
    xin = x.permute(0, 2, 3, 1).contiguous() if last else x
    rin = ref.permute(0, 2, 3, 1).contiguous() if last else ref
    hists = hb.compute_reference_histograms(rin.to(dev))
    got = hb.transform(xin.to(dev), hists)
    got = got.permute(0, 3, 1, 2) if last else got
    want = so.hm_transform(x.numpy(), so.hm_fit(ref.numpy()))
    g = got.cpu().numpy()
    if g.dtype != want.dtype or not np.array_equal(g, want):
        diff = np.abs(g.astype(np.float64) - want.astype(np.float64)).max()
        if diff > (0 if dt == torch.uint8 else 1e-6):
            bad += 1
            print(f"HM MISMATCH case {case}: n={n} {h}x{w} {dt} last={last} max diff {diff}", flush=True)
    # Reinhard
    rb = ReinhardHIP(dev)
    mean, std = rb.compute_reference_mean_std(ref.to(dev))
    out = rb.transform(x.to(dev), mean, std).cpu().numpy()
    m_o, s_o = so.reinhard_fit(ref.numpy())
    want = so.reinhard_transform(x.numpy(), m_o, s_o)
    diff = np.abs(out.astype(np.float64) - want.astype(np.float64)).max()
    tol = 1.0 if dt == torch.uint8 else (2e-3 if dt == torch.float16 else 2e-4)
    if out.dtype != want.dtype or diff > tol:
        bad += 1
        print(f"REINHARD MISMATCH case {case}: n={n} {h}x{w} {dt} max diff {diff} (tol {tol}) dtypes {out.dtype} {want.dtype}", flush=True)
print(f"{cases} cases, {bad} mismatches", flush=True)
