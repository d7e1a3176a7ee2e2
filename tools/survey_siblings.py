#!/usr/bin/env python3
"""Time per call of the sibling paths over element types, layouts and shapes: Macenko fit (single tile, pooled batch) and
precision="sampled", Reinhard fit / transform, histogram matching fit / transform (planar and channels-last).  A row whose
microseconds per megapixel are out of line with its neighbours is a path to look at."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from stainx_amd import synth  # noqa: E402
from stainx_amd.backends.torch_hip_backend import HistogramMatchingHIP, MacenkoHIP, ReinhardHIP  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, reps=20):
    """Median of three runs of `reps` calls (one stall of the box inside a single run once printed 3.9 ms for a 116 us fit)."""
    for _ in range(3):
        fn()
    runs = []
    for _ in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        runs.append(e0.elapsed_time(e1) / reps * 1e3)
    return round(sorted(runs)[1], 1)      # us


mac, mac_fast, rei = MacenkoHIP(dev), MacenkoHIP(dev, precision="sampled"), ReinhardHIP(dev)
hm, hm_last = HistogramMatchingHIP(dev), HistogramMatchingHIP(dev, channel_axis=-1)
ref = synth.reference_tile(256, 256).to(dev)
he, mc = mac.compute_reference_stain_matrix(ref)
r_mean, r_std = rei.compute_reference_mean_std(synth.as_dtype(ref.cpu(), torch.float32).to(dev))
r_hist = hm.compute_reference_histograms(ref)
for n, h, w in [(64, 512, 512), (256, 224, 224), (16, 1024, 1024), (4, 2048, 2048), (1, 512, 512), (1, 2048, 2048), (5, 321, 199)]:
    src = synth.he_batch(min(n, 8), h, w)
    src = src.repeat((n + src.shape[0] - 1) // src.shape[0], 1, 1, 1)[:n]
    mp = n * h * w / 1e6
    for name in ("uint8", "bfloat16", "float32"):
        x = synth.as_dtype(src, getattr(torch, name)).to(dev)
        xl = x.permute(0, 2, 3, 1).contiguous()
        row = {"shape": [n, h, w], "dtype": name,
               "macenko_fit": timed(lambda: mac.compute_reference_stain_matrix(x)),
               "macenko_fast": timed(lambda: mac_fast.transform(x, he, mc)),
               "reinhard_fit": timed(lambda: rei.compute_reference_mean_std(x)),
               "reinhard": timed(lambda: rei.transform(x, r_mean, r_std)),
               "hm_fit": timed(lambda: hm.compute_reference_histograms(x)),
               "hm": timed(lambda: hm.transform(x, r_hist)),
               "hm_nhwc": timed(lambda: hm_last.transform(xl, r_hist))}
        row["us_per_MP"] = {k: round(v / mp, 1) for k, v in row.items() if k not in ("shape", "dtype")}
        print(json.dumps(row), flush=True)
