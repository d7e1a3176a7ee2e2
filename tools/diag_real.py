"""Why do slots of real tiles leave the two-pass form's speculative path?  Decodes the per-slot reason codes
(1 preconditions: prior gave up / rank outside the candidates, 2 frame or boundary check, 3 answer outside its bounds, 4 overflow)."""
import sys, json, numpy as np, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from stainx_amd import synth, _native
from stainx_amd.backends.torch_hip_backend import MacenkoHIP
from tests.golden.cases import real_quadrants_512
dev = torch.device("cuda:0")
root = __import__("pathlib").Path(__file__).resolve().parents[1]
imgs = torch.from_numpy(np.load(str(root / "tests/golden/g11_real_images.npz"))["images_u8"])
be = MacenkoHIP(dev, diag=True)
sm, tmc = be.compute_reference_stain_matrix(imgs[0:1].to(dev))
quads = torch.stack([imgs[i, :, y:y + 512, x:x + 512] for i, y, x in real_quadrants_512()]).contiguous()
x = synth.as_dtype(quads, torch.float32).to(dev)
be.transform(x, sm, tmc, _extra_flags=_native.MACENKO_TWO_PASS)
p = be.tile_params(len(quads))
raw = p["fell_back"]
for i in range(len(quads)):
    fb = int(raw[i]) & 15
    why = [(int(raw[i]) >> (8 + 4 * s)) & 15 for s in range(4)]
    white = float((quads[i].min(0).values > 235).float().mean())
    print(json.dumps({"tile": i, "image": real_quadrants_512()[i][0], "background_frac": round(white, 3), "n_kept_frac": round(float(p["n_kept"][i]) / 512 / 512, 3),
                      "slow_slots": fb, "why": why, "cand_pct": [round(float(v) / 512 / 512 * 100, 2) for v in p["n_candidates"][i]]}))
