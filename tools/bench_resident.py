"""Per-call device time of the Macenko transform in its forms (resident / default / four-pass) for the shapes that matter.
    python tools/bench_resident.py [--real]
"""
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from stainx_amd import _native, synth  # noqa: E402
from stainx_amd.backends.torch_hip_backend import MacenkoHIP  # noqa: E402

dev = torch.device("cuda:0")
be = MacenkoHIP(dev)
SM = torch.tensor(synth.HE_REF, dtype=torch.float32)
TMC = torch.tensor([1.9705, 1.0308], dtype=torch.float32)


def timeit(fn, reps=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    t = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(reps)]) * 1e3
    return float(np.median(t)), float(t.mean()), float(t.min())


def real_batch(n=64):
    z = np.load(ROOT / "tests" / "golden" / "g11_real_images.npz")["images_u8"]
    tiles = []
    for i in range(6):
        for y in range(0, 1024, 512):
            for x in range(0, 1024, 512):
                tiles.append(z[i, :, y:y + 512, x:x + 512])
    rng = np.random.default_rng(0)
    while len(tiles) < n:
        i, y, x = int(rng.integers(0, 6)), int(rng.integers(0, 512)), int(rng.integers(0, 512))
        tiles.append(z[i, :, y:y + 512, x:x + 512])
    return torch.from_numpy(np.stack(tiles[:n]))


rows = []
cases = [("synthetic 64x512x512", synth.he_batch(64, 512, 512), [torch.float32, torch.uint8, torch.bfloat16], {}),
         ("real 64x512x512", real_batch(), [torch.float32, torch.uint8], {}),
         ("config5 256x224x224 unit", synth.he_batch(256, 224, 224, seed0=900), [torch.bfloat16, torch.uint8], {"normalize_to_0_1": True}),
         ("synthetic 16x1024x1024", synth.he_batch(16, 1024, 1024, seed0=50), [torch.float32], {}),
         ("synthetic 8x512x512", synth.he_batch(8, 512, 512, seed0=60), [torch.float32], {}),
         ("synthetic 1x512x512", synth.he_batch(1, 512, 512, seed0=61), [torch.float32], {})]
for name, src, dts, kw in cases:
    for dt in dts:
        xs = [synth.as_dtype(src, dt).to(dev), synth.as_dtype(src.flip(0), dt).to(dev)]
        row = {"case": name, "dtype": str(dt).split(".")[-1]}
        for form, flag in (("resident", _native.MACENKO_RESIDENT), ("default", 0), ("four_pass", _native.MACENKO_CLASSIC)):
            i = [0]

            def call():
                i[0] ^= 1
                be.transform(xs[i[0]], SM, TMC, _extra_flags=flag, **kw)

            med, mean, best = timeit(call)
            row[form + "_us"] = round(med, 1)
        rows.append(row)
        print(json.dumps(row), flush=True)
