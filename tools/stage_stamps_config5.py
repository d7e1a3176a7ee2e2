"""Per-phase time stamps of the four-pass form's per-tile stages on the configs[4] shape (256 x 224 x 224 bf16), debug build
(STAINX_HIP_LIB=stainx_amd/_lib/libstainx_dbg.so).  Stamp slots: plane 0-4 (start, moments, eigen, keys, brackets), stain 6-10
(start, prefetch, resolve, vectors, brackets), scale 12-13."""
import sys, torch
root = __import__("pathlib").Path(__file__).resolve().parents[1]
sys.path.insert(0, str(root))
from stainx_amd import synth, _native
from stainx_amd.backends.torch_hip_backend import MacenkoHIP
dev = torch.device("cuda:0")
be = MacenkoHIP(dev)
n, h, w = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (256, 224, 224)
tiles = synth.he_batch(n, h, w)
x = synth.as_dtype(tiles, torch.bfloat16).to(dev)
sm, tmc = be.compute_reference_stain_matrix(synth.reference_tile(h, w).to(dev))
for _ in range(3):
    be.transform(x, sm, tmc, _extra_flags=_native.MACENKO_CLASSIC)
torch.cuda.synchronize()
p = be.tile_params(n)
s = p["stamps_us"]
d = lambda a, b: (s[:, b] - s[:, a])
med = lambda t: round(float(t.median()), 1)
print("plane phases (moments, eigen, keys, brackets) median:", [med(d(i, i + 1)) for i in range(0, 4)], "total median", med(d(0, 4)), "max", round(float(d(0, 4).max()), 1))
print("stain phases (6->7 prefetch, 7->8 resolve, 8->10 vectors+brackets) median:", [med(d(6, 7)), med(d(7, 8)), med(d(8, 10))], "total median", med(d(6, 10)), "max", round(float(d(6, 10).max()), 1))
print("scale 12->13 median", med(d(12, 13)), "max", round(float(d(12, 13).max()), 1))
print("candidates per slot median", p["n_candidates"].median(0).values.tolist())
