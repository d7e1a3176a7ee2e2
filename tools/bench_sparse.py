#!/usr/bin/env python3
"""Tiles with little or no tissue in a config-2 batch (64x3x512x512 fp32): time per call and which selection paths ran.
A tissue patch of side s in the corner of an otherwise white tile leaves s*s kept pixels; the 4096-pixel sample sees s*s/64 of them."""
import json
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import stain_oracle as so  # noqa: E402
from stainx_amd import Macenko, synth  # noqa: E402
from stainx_amd.backends.torch_hip_backend import MacenkoHIP  # noqa: E402

dev = torch.device("cuda:0")
norm = Macenko(device=dev).fit(synth.reference_tile(512, 512).to(dev))
sm, tmc = norm._stain_matrix, norm._target_max_conc
be = MacenkoHIP(dev)
base = synth.he_batch(64, 512, 512)
rng = np.random.default_rng(3)


def sparse(tile_u8: torch.Tensor, side: int) -> torch.Tensor:
    out = torch.from_numpy(rng.integers(236, 250, size=tuple(tile_u8.shape), dtype=np.uint8))      # background: OD < 0.15 in every channel
    if side:
        r0, c0 = int(rng.integers(0, 512 - side)), int(rng.integers(0, 512 - side))
        out[:, r0:r0 + side, c0:c0 + side] = tile_u8[:, r0:r0 + side, c0:c0 + side]
    return out


for label, sides in (("all tissue", []), ("1 blank", [0]), ("8 blank", [0] * 8), ("1 tile 16x16 tissue", [16]), ("1 tile 48x48", [48]), ("1 tile 128x128", [128]), ("8 tiles 64x64", [64] * 8),
                     ("32 tiles mixed", [0, 8, 16, 24, 32, 48, 64, 96] * 4)):
    u8 = base.clone()
    for k, side in enumerate(sides):
        u8[2 * k] = sparse(base[2 * k], side)
    x = synth.as_dtype(u8, torch.float32).to(dev)
    for _ in range(5):
        out = be.transform(x, sm, tmc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        out = be.transform(x, sm, tmc)
    e1.record()
    torch.cuda.synchronize()
    p = be.tile_params(64)
    fb = p["fell_back"]
    # parity of the odd tiles against the oracle (first modified tile)
    err = None
    if sides:
        want = so.macenko_transform(x[0:1].cpu().numpy(), sm.cpu().numpy(), tmc.cpu().numpy(), signs="positive_sum")
        err = float(np.abs(out[0:1].cpu().numpy() - want).max())
    print(json.dumps({"case": label, "ms": round(e0.elapsed_time(e1) / 30, 4), "tiles_fell_back": int((fb != 0).sum()), "fell_back_bits": sorted(set(int(v) for v in fb if v)),
                      "n_kept_tile0": int(p["n_kept"][0]), "use_all_tiles": int(p["use_all"].sum()), "max_abs_tile0_vs_oracle": err}), flush=True)
