#!/usr/bin/env python3
"""Randomised sweep of the two-pass Macenko transform against the four-pass form of the same library (bit for bit), counting
how often a speculation failed its proof (slow exact path).   python tools/stress_twopass.py [seed] [cases]"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from stainx_amd import _native, synth  # noqa: E402
from stainx_amd.backends.torch_hip_backend import MacenkoHIP  # noqa: E402

dev = torch.device("cuda:0")
be = MacenkoHIP(dev)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dtypes = [torch.uint8, torch.float16, torch.bfloat16, torch.float32, torch.float64]
sm = torch.tensor(synth.HE_REF).to(dev)
tmc = torch.tensor([1.9705, 1.0308]).to(dev)
slots = failed = mismatches = tiles = 0
cand = []
for case in range(cases):
    big = rng.random() < 0.25
    h, w = (int(rng.integers(300, 1025)), int(rng.integers(300, 1025))) if big else (int(rng.integers(24, 300)), int(rng.integers(24, 300)))
    n = int(rng.integers(1, 4 if big else 9))
    dt = dtypes[int(rng.integers(0, len(dtypes)))]
    unit, last = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    u8 = synth.he_batch(n, h, w, seed0=int(rng.integers(0, 1 << 20)), scale_step=float(rng.uniform(0, 0.2)))
    x = synth.as_dtype(u8, dt).to(dev)
    if last:
        x = x.permute(0, 2, 3, 1).contiguous()
    two = be.transform(x, sm, tmc, normalize_to_0_1=unit, channels_last=last, _extra_flags=_native.MACENKO_TWO_PASS)
    p2 = be.tile_params(n)
    one = be.transform(x, sm, tmc, normalize_to_0_1=unit, channels_last=last, _extra_flags=_native.MACENKO_CLASSIC)
    p1 = be.tile_params(n)
    same = torch.equal(two.view(torch.uint8), one.view(torch.uint8)) and all(torch.equal(p2[k], p1[k]) for k in ("n_kept", "vecs", "he", "max_c", "phi_lo", "phi_hi"))
    if not same:
        mismatches += 1
        print(f"MISMATCH case {case}: n={n} {h}x{w} {dt} unit={unit} last={last} max|d|={(two.double() - one.double()).abs().max().item():.3g}", flush=True)
    fb = p2["fell_back"] & 15
    why = p2["fell_back"] >> 8
    tiles += n
    slots += 4 * n
    nf = int(sum(bin(int(v)).count("1") for v in fb))
    failed += nf
    if nf:
        print(f"fallback case {case}: n={n} {h}x{w} {dt} fell_back={fb.tolist()} why={[hex(int(v)) for v in why]} cand%={[round(float(v), 1) for v in (p2['n_candidates'].double().sum(0) / (n * h * w) * 100)]}", flush=True)
    cand.append(float(p2["n_candidates"].double().sum() / (n * h * w) * 100))
print(f"{cases} cases, {tiles} tiles: {mismatches} mismatches, {failed} of {slots} slots took the slow path; candidates (all four slots) mean {np.mean(cand):.1f} % of the pixels, max {np.max(cand):.1f} %", flush=True)
