"""Which slots of bench.py's real-tissue batches (150 overlapping 512 x 512 crops of the reference's six example images) leave the two-pass form's
speculative path, and why (1 preconditions, 2 frame / boundary check, 3 answer outside its bounds, 4 candidate records overflowed).
    STAINX_DIAG=1 [SX_SPEC_SIGMAS_CONC=8 ...] python tools/diag_real_batches.py"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from stainx_amd import _native, synth
from stainx_amd.backends.torch_hip_backend import MacenkoHIP

dev = torch.device("cuda", 0)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
imgs = torch.from_numpy(np.load(os.path.join(root, "tests", "golden", "g11_real_images.npz"))["images_u8"])
crops = torch.stack([imgs[i, :, y:y + 512, x:x + 512] for i in range(6) for y in range(0, 513, 128) for x in range(0, 513, 128)])      # 150 tiles
be = MacenkoHIP(dev, diag=True)
sm, tmc = be.compute_reference_stain_matrix(imgs[0:1].to(dev))
why_hist, slow_tiles, cand = collections.Counter(), 0, [[] for _ in range(4)]
for b0 in range(0, 150, 50):
    x = synth.as_dtype(crops[b0:b0 + 50], torch.float32).to(dev)
    be.transform(x, sm, tmc, _extra_flags=_native.MACENKO_TWO_PASS)
    p = be.tile_params(x.shape[0])
    for i in range(x.shape[0]):
        raw = int(p["fell_back"][i])
        if raw & 15:
            slow_tiles += 1
            for s in range(4):
                w = (raw >> (8 + 4 * s)) & 15
                if (raw >> s) & 1:
                    why_hist[(s, w)] += 1
        for s in range(4):
            cand[s].append(float(p["n_candidates"][i][s]) / 512 / 512 * 100)
print("tiles with a slow slot:", slow_tiles, "of 150;  (slot, why) counts:", dict(why_hist))
print("candidates per slot, % of the tile's pixels: mean", [round(float(np.mean(c)), 2) for c in cand], "max", [round(float(np.max(c)), 2) for c in cand])
