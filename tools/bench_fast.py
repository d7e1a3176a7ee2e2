#!/usr/bin/env python3
"""Macenko(precision="sampled") against the exact path: time and error on the config-2 batch (64x3x512x512 fp32)."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from stainx_amd import Macenko, synth  # noqa: E402

dev = torch.device("cuda:0")
ref = synth.reference_tile(512, 512).to(dev)
x = synth.as_dtype(synth.he_batch(64, 512, 512), torch.float32).to(dev)
res = {}
for precision in ("stable", "sampled"):
    norm = Macenko(device=dev, precision=precision).fit(ref)
    for _ in range(10):
        out = norm.transform(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        out = norm.transform(x)
    e1.record()
    torch.cuda.synchronize()
    res[precision] = (out, e0.elapsed_time(e1) / 50)
d = (res["sampled"][0] - res["stable"][0]).abs()
print(json.dumps({"stable_ms": round(res["stable"][1], 4), "fast_ms": round(res["sampled"][1], 4), "fast_megapixels_per_s": round(64 * 512 * 512 / 1e3 / res["sampled"][1], 1),
                  "fast_vs_stable_max_abs_0_255": round(float(d.max()), 3), "mean_abs_0_255": round(float(d.mean()), 4),
                  "per_tile_mean_abs_max": round(float(d.reshape(64, -1).mean(1).max()), 4)}))
