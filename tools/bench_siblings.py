#!/usr/bin/env python3
"""Timing of the sibling paths on the BASELINE.json shapes (not the headline bench):
   histogram matching 64x3x1024x1024 uint8 (configs[2]), Reinhard 64x3x512x512 fp32 and 1x3x512x512 fp32
   (configs[0]), Macenko 256x3x224x224 bf16 through StainNormalizerTransform (configs[4], one GPU's share).
Prints one JSON object per workload with megapixels/s and the fraction of the 8 TB/s HBM peak on the
algorithmic bytes (one read + one write of every pixel)."""
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from stainx_amd import HistogramMatching, Reinhard, StainNormalizerTransform, synth  # noqa: E402

dev = torch.device("cuda:0")
steps, warmup = 50, 10


def timed(fn):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record()
    for i in range(steps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
    return sum(ms) / len(ms), min(ms)


def report(name, pixels, bytes_per_px, ms_mean, ms_min):
    gbs = pixels * bytes_per_px / (ms_mean * 1e-3) / 1e9
    print(json.dumps({"workload": name, "megapixels_per_s": round(pixels / 1e6 / (ms_mean * 1e-3), 1), "ms_per_call": round(ms_mean, 4), "ms_min": round(ms_min, 4),
                      "algorithmic_bytes_per_px": bytes_per_px, "achieved_GBs": round(gbs, 1), "frac_of_8TBs": round(gbs / 8000, 4)}), flush=True)


# histogram matching, config 3
g = torch.Generator().manual_seed(43)
src = (torch.rand(64, 3, 1024, 1024, generator=g) * 255).round().to(torch.uint8).to(dev)
ref = synth.noise_u8((1, 3, 1024, 1024), 42).to(dev)
hm = HistogramMatching(device=dev).fit(ref)
report("HistogramMatching.transform 64x3x1024x1024 u8", 64 * 1024 * 1024, 6, *timed(lambda: hm.transform(src)))
del src

# Reinhard
x = synth.as_dtype(synth.noise_u8((64, 3, 512, 512), 43), torch.float32).to(dev)
rn = Reinhard(device=dev).fit(synth.as_dtype(synth.noise_u8((1, 3, 512, 512), 42), torch.float32).to(dev))
report("Reinhard.transform 64x3x512x512 f32", 64 * 512 * 512, 24, *timed(lambda: rn.transform(x)))
x1 = x[:1].contiguous()
report("Reinhard.fit+transform 1x3x512x512 f32 (configs[0] shape)", 512 * 512, 24, *timed(lambda: rn.fit(x1).transform(x1)))
del x

# Macenko bf16 through the module, config 5 (one GPU's share of the batch)
tiles = synth.as_dtype(synth.he_batch(256, 224, 224), torch.bfloat16).to(dev)
t = StainNormalizerTransform(method="macenko", mode="reference", reference=synth.as_dtype(synth.reference_tile(224, 224), torch.bfloat16).to(dev))
report("StainNormalizerTransform(macenko, reference) 256x3x224x224 bf16", 256 * 224 * 224, 12, *timed(lambda: t(tiles)))

# Macenko on the config-2 shape for the other input dtypes (algorithmic bytes: u8 6 B/px, bf16/f16 12 B/px)
from stainx_amd import Macenko  # noqa: E402

u8 = synth.he_batch(64, 512, 512)
mk = Macenko(device=dev).fit(synth.reference_tile(512, 512).to(dev))
for name, dt, bpp in (("u8", torch.uint8, 6), ("bf16", torch.bfloat16, 12), ("f16", torch.float16, 12), ("f32", torch.float32, 24)):
    xin = synth.as_dtype(u8, dt).to(dev)
    report(f"Macenko.transform 64x3x512x512 {name}", 64 * 512 * 512, bpp, *timed(lambda: mk.transform(xin)))

# the same tiles as the decoder hands them over: (N,H,W,3) uint8, layout kept on the way out (C flag SX_MACENKO_CHANNELS_LAST)
from stainx_amd.backends.torch_hip_backend import MacenkoHIP  # noqa: E402

be = MacenkoHIP(dev)
he, max_c = be.compute_reference_stain_matrix(synth.reference_tile(512, 512).to(dev))
x_last = u8.to(dev).permute(0, 2, 3, 1).contiguous()
report("Macenko.transform 64x512x512x3 u8 (channels last)", 64 * 512 * 512, 6, *timed(lambda: be.transform(x_last, he, max_c, channels_last=True)))
