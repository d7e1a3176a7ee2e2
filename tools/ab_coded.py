"""A/B of the four passes over float32 tiles with and without the 8-bit codes (diagnostic build: SX_MACENKO_NO_CODES), device time per call
over two rotating batches, synthetic config-2 tiles and crops of the reference's real images.
    STAINX_DIAG=1 python tools/ab_coded.py [calls]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from stainx_amd import _native, synth
from stainx_amd.backends.torch_hip_backend import MacenkoHIP

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0)
be = MacenkoHIP(dev, diag=True)
sm = torch.tensor([[0.5626, 0.2159], [0.7201, 0.8012], [0.4062, 0.5581]], dtype=torch.float32)
tmc = torch.tensor([1.9705, 1.0308], dtype=torch.float32)


def real_batches():
    d = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "g11_real_images.npz"))
    imgs = torch.from_numpy(d["images_u8"])
    out = []
    for b in range(2):
        tiles = []
        for i in range(64):
            im = imgs[(i + b) % imgs.shape[0]]
            y0, x0 = (37 * i + 101 * b) % 512, (53 * i + 71 * b) % 512
            tiles.append(im[:, y0:y0 + 512, x0:x0 + 512])
        out.append((torch.stack(tiles).to(torch.float32) / 255.0).to(dev))
    return out


def time_it(batches, flags):
    for i in range(20):
        be.transform(batches[i % 2], sm, tmc, _extra_flags=flags)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(calls)]
    for i in range(calls):
        ev[i][0].record()
        be.transform(batches[i % 2], sm, tmc, _extra_flags=flags)
        ev[i][1].record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2] * 1e3, t[0] * 1e3


C, N = _native.MACENKO_CLASSIC, _native.MACENKO_NO_CODES
for name, batches in (("synthetic 64x512x512 f32", [synth.as_dtype(synth.he_batch(64, 512, 512, seed0=1000 + 64 * b), torch.float32).to(dev) for b in range(2)]),
                      ("real crops 64x512x512 f32", real_batches()),
                      ("synthetic 256x224x224 f32", [synth.as_dtype(synth.he_batch(256, 224, 224, seed0=3000 + 256 * b), torch.float32).to(dev) for b in range(2)])):
    for label, flags in (("four passes, float pixels every pass", C | N), ("four passes, 8-bit codes behind the first", C), ("two-pass form, float pixels in both passes", _native.MACENKO_TWO_PASS | N), ("two-pass form, 8-bit codes behind pass A", _native.MACENKO_TWO_PASS)):
        if os.environ.get("AB_ONLY") and os.environ["AB_ONLY"] not in label:
            continue
        med, best = time_it(batches, flags)
        print(f"{name:28s} {label:44s} median {med:7.1f} us   min {best:7.1f} us", flush=True)
