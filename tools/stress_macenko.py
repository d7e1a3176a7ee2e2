#!/usr/bin/env python3
"""Randomised parity sweep of the Macenko transform and fit against the CPU oracle (not part of the test suite:
run it on a GPU box after kernel changes).  Random batch sizes, odd tile sizes, all dtypes, both layouts, fused /255."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import stain_oracle as so  # noqa: E402
from stainx_amd import synth  # noqa: E402
from stainx_amd.backends.torch_hip_backend import MacenkoHIP  # noqa: E402

dev = torch.device("cuda:0")
be = MacenkoHIP(dev)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 150
dtypes = [torch.uint8, torch.float16, torch.bfloat16, torch.float32, torch.float64]
ref_he, ref_mc = so.macenko_fit(synth.reference_tile(96, 96).numpy())
worst = {}
for case in range(cases):
    n = int(rng.integers(1, 6))
    h, w = int(rng.integers(8, 260)), int(rng.integers(8, 260))
    dt = dtypes[int(rng.integers(0, len(dtypes)))]
    unit = bool(rng.integers(0, 2))
    last = bool(rng.integers(0, 2))
    u8 = synth.he_batch(n, h, w, seed0=int(rng.integers(0, 1 << 20)), scale_step=float(rng.uniform(0, 0.2)))
    x = synth.as_dtype(u8, dt)
    want, params = so.macenko_transform(x.numpy() if dt != torch.bfloat16 else x.float().numpy(), ref_he, ref_mc, return_params=True, signs="positive_sum")
    if dt == torch.bfloat16:      # numpy has no bfloat16: the oracle ran on the same values in fp32, cast its result the same way
        want = torch.from_numpy(want).to(torch.bfloat16).float().numpy()
    if unit:
        want = so.apply_normalize_to_0_1(want.astype(np.float32) if dt == torch.bfloat16 else want)
        if dt == torch.bfloat16:
            want = (torch.from_numpy(np.asarray(want, dtype=np.float32) * 255.0).to(torch.bfloat16) / 255.0).float().numpy()
    xin = x.to(dev)
    if last:
        xin = xin.permute(0, 2, 3, 1).contiguous()
    out = be.transform(xin, torch.from_numpy(ref_he), torch.from_numpy(ref_mc), normalize_to_0_1=unit, channels_last=last)
    if dt == torch.uint8:      # the fused half-precision output is the cast of this output, bit for bit (any size, layout, /255)
        for od_t in (torch.bfloat16, torch.float16):
            half = be.transform(xin, torch.from_numpy(ref_he), torch.from_numpy(ref_mc), normalize_to_0_1=unit, channels_last=last, out_dtype=od_t)
            if not torch.equal(half, out.to(od_t)):
                print(f"MISMATCH case {case}: uint8 -> {od_t} differs from the cast: n={n} {h}x{w} unit={unit} last={last}", flush=True)
                worst["half-output mismatches"] = worst.get("half-output mismatches", 0) + 1
    if last:
        out = out.permute(0, 3, 1, 2)
    got = out.float().cpu().numpy() if out.dtype in (torch.bfloat16, torch.float16) else out.cpu().numpy()
    scale = 1.0 if unit else 255.0
    tol = {torch.uint8: (1.0 if not unit else 1.0 / 255 + 2e-7), torch.float16: 0.26 * scale / 255 * 4, torch.bfloat16: 2.1 * scale / 255 * 4, torch.float32: 2.55e-2 * scale / 255,
           torch.float64: 2.55e-2 * scale / 255}[dt]
    # (uint8 with /255: one grey level between two float32 quotients k/255: 1/255 give or take their roundings)
    # a tile whose two small covariance eigenvalues nearly coincide has no stable stain plane (in the reference either): its
    # middle eigenvector, and with it the output, moves with the last bits of the covariance -- such tiles are counted, not compared
    x_f = x.float().numpy() if dt == torch.bfloat16 else x.numpy()
    stable = np.ones(n, dtype=bool)
    for i in range(n):
        rows = so.optical_density(so.to_unit_float(x_f[i:i + 1]))[0].reshape(3, -1).T.astype(np.float64)
        kept = rows[rows.min(1) >= so.BETA]
        lam = np.linalg.eigvalsh(np.cov((kept if kept.shape[0] >= 3 else rows).T))
        stable[i] = (lam[1] - lam[0]) >= 1e-3 * lam[2]
    worst["ill-conditioned tiles skipped"] = worst.get("ill-conditioned tiles skipped", 0) + int((~stable).sum())
    if not stable.any():
        continue
    err = float(np.abs(got.astype(np.float64) - np.asarray(want, dtype=np.float64))[stable].max())
    worst[dt] = max(worst.get(dt, 0.0), err / tol)
    p = be.tile_params(n)
    if err > tol:
        print(f"MISMATCH case {case}: n={n} {h}x{w} {dt} unit={unit} last={last} err={err:.4g} tol={tol:.4g} fell_back={p['fell_back'].tolist()}", flush=True)
    # pooled fit of the same batch
    he, mc = be.compute_reference_stain_matrix(x.to(dev))
    x_np = x.numpy() if dt != torch.bfloat16 else x.float().numpy()
    he_o, mc_o = so.macenko_fit(x_np, signs="positive_sum")
    # the middle eigenvector is only defined as well as the two small eigenvalues are apart: a tile whose optical densities
    # lie (nearly) on a line has no stable stain plane, in the reference either -- such cases are counted, not compared
    od = so.optical_density(so.to_unit_float(x_np))
    rows = np.transpose(od, (1, 0, 2, 3)).reshape(3, -1).T.astype(np.float64)
    lam = np.linalg.eigvalsh(np.cov(rows[rows.min(1) >= so.BETA].T))
    if (lam[1] - lam[0]) < 1e-3 * lam[2]:
        worst["ill-conditioned fits skipped"] = worst.get("ill-conditioned fits skipped", 0) + 1
        continue
    if np.abs(he.cpu().numpy() - he_o).max() > 2e-4 or np.abs(mc.cpu().numpy() / mc_o - 1).max() > 2e-4:
        print(f"FIT MISMATCH case {case}: n={n} {h}x{w} {dt} he_err={np.abs(he.cpu().numpy() - he_o).max():.3g} mc={mc.cpu().numpy()} vs {mc_o}", flush=True)
print("worst error / tolerance per dtype:", {str(k): round(v, 3) for k, v in worst.items()}, flush=True)
