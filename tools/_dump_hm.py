import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import stain_oracle as so
from stainx_amd import synth
from stainx_amd.backends.torch_hip_backend import HistogramMatchingHIP
dev = torch.device("cuda:0")
rng = np.random.default_rng(3)
dtypes = [torch.uint8, torch.float16, torch.float32, torch.float64]
saved = {}
for case in range(150):
    n = int(rng.integers(1, 5)); h, w = int(rng.integers(4, 200)), int(rng.integers(4, 200))
    dt = dtypes[int(rng.integers(0, len(dtypes)))]
    s1, s2 = int(rng.integers(0, 1 << 20)), int(rng.integers(0, 1 << 20))
    last = bool(rng.integers(0, 2))
    if case in (1, 8, 45, 108):
        src_u8 = synth.noise_u8((n, 3, h, w), s1); ref_u8 = synth.noise_u8((1, 3, h, w), s2)
        x, ref = synth.as_dtype(src_u8, dt), synth.as_dtype(ref_u8, dt)
        hb = HistogramMatchingHIP(dev, channel_axis=1)
        hists = hb.compute_reference_histograms(ref.to(dev))
        got = hb.transform(x.to(dev), hists).cpu().numpy()
        t = hb.tables()
        want = so.hm_transform(x.numpy(), so.hm_fit(ref.numpy()))
        saved[f"c{case}_x"] = x.numpy(); saved[f"c{case}_ref"] = ref.numpy(); saved[f"c{case}_gpu"] = got; saved[f"c{case}_oracle"] = want
        saved[f"c{case}_lut"] = t["lut"].numpy(); saved[f"c{case}_counts"] = t["counts"].numpy()
        saved[f"c{case}_hists"] = np.stack([hh.cpu().numpy() for hh in hists]) if isinstance(hists, (list, tuple)) else hists.cpu().numpy()
np.savez_compressed("gpurun_out/hm_cases.npz", **saved)
print("saved", list(saved)[:6])
