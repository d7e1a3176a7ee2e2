#!/usr/bin/env python3
"""Macenko.transform latency/throughput against the batch size (512x512 fp32 tiles)."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from stainx_amd import Macenko, synth  # noqa: E402

dev = torch.device("cuda:0")
norm = Macenko(device=dev).fit(synth.reference_tile(512, 512).to(dev))
src = synth.as_dtype(synth.he_batch(256, 512, 512), torch.float32).to(dev)
for n in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    x = src[:n].contiguous()
    for _ in range(5):
        norm.transform(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        norm.transform(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    print(json.dumps({"tiles": n, "ms_per_call": round(ms, 4), "megapixels_per_s": round(n * 512 * 512 / 1e3 / ms, 1)}), flush=True)
