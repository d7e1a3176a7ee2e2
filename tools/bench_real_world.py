#!/usr/bin/env python3
"""Config-2-shaped uint8 / fp32 batches with the artefacts of real slides: saturated white background (exact 255), black
pen marks / borders (exact 0), both as large tie groups of identical pixels.  Time per call, selection paths, parity."""
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


def paint(u8: torch.Tensor, white: float, black: float) -> torch.Tensor:
    out = u8.clone()
    h = out.shape[-2]
    out[..., : int(h * white), :] = 255          # top rows: saturated background
    if black:
        out[..., h - int(h * black):, :] = 0     # bottom rows: pen mark / border
    return out


for label, white, black in (("tissue only", 0.0, 0.0), ("40% white", 0.4, 0.0), ("40% white + 0.5% black", 0.4, 0.005), ("40% white + 5% black", 0.4, 0.05), ("20% white + 30% black", 0.2, 0.3),
                            ("90% white", 0.9, 0.0)):
    u8 = paint(base, white, black)
    for dt in (torch.uint8, torch.float32):
        x = synth.as_dtype(u8, dt).to(dev)
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
        want, params = so.macenko_transform(x[:1].cpu().numpy(), sm.cpu().numpy(), tmc.cpu().numpy(), return_params=True, signs="positive_sum")
        err = float(np.abs(out[:1].cpu().numpy().astype(np.float64) - want.astype(np.float64)).max())
        print(json.dumps({"case": label, "dtype": str(dt).split(".")[1], "ms": round(e0.elapsed_time(e1) / 30, 4), "fell_back_bits": sorted(set(int(v) for v in p["fell_back"] if v)),
                          "n_candidates_tile0": [int(v) for v in p["n_candidates"][0]], "max_abs_tile0_vs_oracle": err,
                          "max_c_rel": float(np.abs(p["max_c"][0].numpy() / params[0]["max_c"] - 1).max())}), flush=True)
