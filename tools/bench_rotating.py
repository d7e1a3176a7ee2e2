#!/usr/bin/env python3
"""Config-2 transform over a ROTATION of input batches: with one buffer (what bench.py and the reference's harness time) the
201 MB input is still in the 256 MB Infinity Cache when the next call starts; with several buffers every call's first pass
reads its batch from HBM, as a pipeline that gets fresh tiles every step does.  Time per call for 1, 2, 4 buffers, per dtype."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from stainx_amd import synth  # noqa: E402
from stainx_amd.backends.torch_hip_backend import MacenkoHIP  # noqa: E402

dev = torch.device("cuda:0")
be = MacenkoHIP(dev)
he, mc = be.compute_reference_stain_matrix(synth.reference_tile(256, 256).to(dev))
src = synth.he_batch(64, 512, 512)
for name in sys.argv[1:] or ["float32", "uint8", "bfloat16"]:
    row = {"dtype": name}
    for buffers in (1, 2, 4):
        xs = [synth.as_dtype(src.roll(k, dims=0), getattr(torch, name)).to(dev) for k in range(buffers)]
        for i in range(12):
            be.transform(xs[i % buffers], he, mc)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(200):
            be.transform(xs[i % buffers], he, mc)
        e1.record()
        torch.cuda.synchronize()
        row[f"{buffers}_buffers_us"] = round(e0.elapsed_time(e1) / 200 * 1e3, 1)
        del xs
    print(json.dumps(row), flush=True)
