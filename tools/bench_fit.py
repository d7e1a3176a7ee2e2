#!/usr/bin/env python3
"""Timing of the pooled batch fit (BASELINE configs[3] family: one stain estimate over all tiles of a batch):
   single-GPU sx_macenko_fit, the staged distributed fit run with world size 1, and fit_transform."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from stainx_amd import Macenko, synth  # noqa: E402
from stainx_amd.backends.torch_hip_backend import MacenkoHIP  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, steps=20, warmup=3):
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


x = synth.as_dtype(synth.he_batch(64, 512, 512), torch.float32).to(dev)
be = MacenkoHIP(dev)
mean, best = timed(lambda: be.compute_reference_stain_matrix(x))
print(json.dumps({"workload": "pooled fit 64x3x512x512 f32 (sx_macenko_fit, one GPU)", "ms_per_call": round(mean, 4), "ms_min": round(best, 4), "megapixels_per_s": round(64 * 512 * 512 / 1e3 / mean, 1)}))


def staged():
    st = be.dfit_begin(be.dfit_moments(x))
    for stage in (0, 1):
        for _ in range(4):
            be.dfit_advance(st, stage, be.dfit_histogram(x, st, stage))
    return be.dfit_result(st)


from stainx_amd import distributed as sxd  # noqa: E402

assert sxd._macenko_fit_pooled_brackets(x, None, be) is not None, "bracket form fell back to the radix rounds"
mean, best = timed(lambda: sxd.macenko_fit_pooled(x, steps=be))
print(json.dumps({"workload": "distributed pooled fit, bracket form (3 passes), world size 1 incl. host choreography", "ms_per_call": round(mean, 4), "ms_min": round(best, 4), "megapixels_per_s": round(64 * 512 * 512 / 1e3 / mean, 1)}))
mean, best = timed(staged)
print(json.dumps({"workload": "distributed pooled fit, radix form (9 passes), world size 1", "ms_per_call": round(mean, 4), "ms_min": round(best, 4), "megapixels_per_s": round(64 * 512 * 512 / 1e3 / mean, 1)}))
norm = Macenko(device=dev)
mean, best = timed(lambda: norm.fit(x).transform(x))
print(json.dumps({"workload": "fit + transform 64x3x512x512 f32 (pooled fit, per-tile transform)", "ms_per_call": round(mean, 4), "ms_min": round(best, 4), "megapixels_per_s": round(64 * 512 * 512 / 1e3 / mean, 1)}))
