"""Per-phase time stamps of the four-pass form's per-tile stages on REAL tiles against the synthetic batch (debug build:
STAINX_HIP_LIB=stainx_amd/_lib/libstainx_dbg.so).  Stamp slots: plane 0-4 (start, moments, eigen, keys, brackets), stain 6-10
(start, prefetch, resolve, vectors, brackets), scale 12-13."""
import sys, numpy as np, torch
root = __import__("pathlib").Path(__file__).resolve().parents[1]
sys.path.insert(0, str(root))
from stainx_amd import synth, _native
from stainx_amd.backends.torch_hip_backend import MacenkoHIP
dev = torch.device("cuda:0")
be = MacenkoHIP(dev, diag=True)
imgs = torch.from_numpy(np.load(str(root / "tests/golden/g11_real_images.npz"))["images_u8"])
crops = torch.stack([imgs[i, :, y:y + 512, x:x + 512] for i in range(6) for y in range(0, 513, 128) for x in range(0, 513, 128)])
real = crops[torch.arange(0, 150, 150 / 64).long()]
torch.set_printoptions(precision=1, linewidth=250, sci_mode=False)
for name, tiles in (("synthetic", synth.he_batch(64, 512, 512)), ("real", real)):
    x = synth.as_dtype(tiles, torch.float32).to(dev)
    sm, tmc = be.compute_reference_stain_matrix(tiles[:1].to(dev))
    for _ in range(3):
        be.transform(x, sm, tmc, _extra_flags=_native.MACENKO_CLASSIC)
    torch.cuda.synchronize()
    p = be.tile_params(64)
    s = p["stamps_us"]
    d = lambda a, b: (s[:, b] - s[:, a])
    print(name, "plane phases (moments, eigen, keys, brackets) median:", [round(float(d(i, i + 1).median()), 1) for i in range(0, 4)], "max total", round(float(d(0, 4).max()), 1))
    print(name, "stain phases (6->7, 7->8, 8->10) median:", [round(float(d(6, 7).median()), 1), round(float(d(7, 8).median()), 1), round(float(d(8, 10).median()), 1)], "max total", round(float(d(6, 10).max()), 1),
          "per tile totals sorted", [round(float(v), 1) for v in d(6, 10).sort().values[::8]])
    print(name, "scale 12->13 median", round(float(d(12, 13).median()), 1), "max", round(float(d(12, 13).max()), 1))
    print(name, "candidates per slot median", p["n_candidates"].median(0).values.tolist(), "max", p["n_candidates"].max(0).values.tolist(), "fell_back", int((p["fell_back"] != 0).sum()))
    slow = d(6, 10).argsort(descending=True)[:4].tolist()      # the stain stage's slowest tiles, phase by phase
    for t in slow:
        print(name, "slow tile", t, "stain phases", [round(float(d(6, 7)[t]), 1), round(float(d(7, 8)[t]), 1), round(float(d(8, 10)[t]), 1)], "candidates", p["n_candidates"][t].tolist(),
              "fell_back bits", hex(int(p["fell_back"][t])), "scale", round(float(d(12, 13)[t]), 1), "plane", round(float(d(0, 4)[t]), 1))
