"""Kernel trace target: the four-pass Macenko transform on REAL tiles (64 crops of 512x512 from tests/golden/g11_real_images.npz) or on
the synthetic batch, float32 or uint8.    rocprofv3 --kernel-trace --stats --output-format csv -d out -o kt -- python3 tools/prof_real.py [real|synth] [f32|u8]"""
import sys, numpy as np, torch
root = __import__("pathlib").Path(__file__).resolve().parents[1]
sys.path.insert(0, str(root))
from stainx_amd import synth, _native
from stainx_amd.backends.torch_hip_backend import MacenkoHIP
dev = torch.device("cuda:0")
which, dt = (sys.argv[1] if len(sys.argv) > 1 else "real"), {"f32": torch.float32, "u8": torch.uint8}[sys.argv[2] if len(sys.argv) > 2 else "f32"]
be = MacenkoHIP(dev)
if which == "real":
    imgs = torch.from_numpy(np.load(str(root / "tests/golden/g11_real_images.npz"))["images_u8"])
    crops = torch.stack([imgs[i, :, y:y + 512, x:x + 512] for i in range(6) for y in range(0, 513, 128) for x in range(0, 513, 128)])
    tiles = crops[torch.arange(0, 150, 150 / 64).long()]
    sm, tmc = be.compute_reference_stain_matrix(imgs[0:1].to(dev))
else:
    tiles = synth.he_batch(64, 512, 512)
    sm, tmc = be.compute_reference_stain_matrix(synth.reference_tile(512, 512).to(dev))
x = synth.as_dtype(tiles, dt).to(dev)
for _ in range(120):
    be.transform(x, sm, tmc, _extra_flags=_native.MACENKO_CLASSIC)
torch.cuda.synchronize()
