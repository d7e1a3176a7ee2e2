#!/usr/bin/env python3
"""uint8 in -> bf16 out (normalised to [0,1]) in one call against the two-step form (transform, then .to(bfloat16)):
config-2 shape as uint8 NHWC, and the config-5 shape (256 x 224 x 224)."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from stainx_amd import Macenko, synth  # noqa: E402
from stainx_amd.backends.torch_hip_backend import MacenkoHIP  # noqa: E402

dev = torch.device("cuda:0")
be = MacenkoHIP(dev)
for n, h, w in ((64, 512, 512), (256, 224, 224)):
    norm = Macenko(device=dev).fit(synth.reference_tile(h, w).to(dev))
    sm, tmc = norm._stain_matrix, norm._target_max_conc
    x = synth.he_batch(n, h, w).permute(0, 2, 3, 1).contiguous().to(dev)
    res = {}
    for label, fn in (("fused", lambda: be.transform(x, sm, tmc, normalize_to_0_1=True, channels_last=True, out_dtype=torch.bfloat16)),
                      ("two_step", lambda: be.transform(x, sm, tmc, normalize_to_0_1=True, channels_last=True).to(torch.bfloat16))):
        for _ in range(10):
            out = fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            out = fn()
        e1.record()
        torch.cuda.synchronize()
        res[label] = (out, e0.elapsed_time(e1) / 100)
    print(json.dumps({"shape": [n, h, w, 3], "fused_ms": round(res["fused"][1], 4), "two_step_ms": round(res["two_step"][1], 4),
                      "megapixels_per_s": round(n * h * w / 1e3 / res["fused"][1], 1), "bit_equal": bool(torch.equal(res["fused"][0], res["two_step"][0]))}), flush=True)
