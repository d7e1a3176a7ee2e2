#!/usr/bin/env python3
"""Time per call of the Macenko transform over a grid of element types, layouts, /255 fusion and shapes -- looks for paths that
are out of line with their neighbours (this is how the uint8 -> float32 store pattern of the reconstruct pass was found)."""
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
shapes = [(64, 512, 512), (256, 224, 224), (16, 1024, 1024), (4, 2048, 2048), (1024, 64, 64), (7, 321, 199), (1, 512, 512)]
for n, h, w in shapes:
    src = synth.he_batch(min(n, 16), h, w)
    src = src.repeat((n + src.shape[0] - 1) // src.shape[0], 1, 1, 1)[:n]
    for name in ("uint8", "float16", "bfloat16", "float32", "float64"):
        xp = synth.as_dtype(src, getattr(torch, name)).to(dev)
        row = {"shape": [n, h, w], "dtype": name}
        for last in (False, True):
            x = xp.permute(0, 2, 3, 1).contiguous() if last else xp
            for unit in (False, True):
                for _ in range(3):
                    be.transform(x, he, mc, channels_last=last, normalize_to_0_1=unit)
                runs = []
                for _ in range(3):      # median of three runs (a stall of the box inside one run is not a property of the path)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        be.transform(x, he, mc, channels_last=last, normalize_to_0_1=unit)
                    e1.record()
                    torch.cuda.synchronize()
                    runs.append(e0.elapsed_time(e1) / 20 * 1e3)
                row[("nhwc" if last else "nchw") + ("/255" if unit else "")] = round(sorted(runs)[1], 1)      # us
        vals = [v for k, v in row.items() if k not in ("shape", "dtype")]
        row["spread"] = round(max(vals) / min(vals), 2)
        print(json.dumps(row), flush=True)
