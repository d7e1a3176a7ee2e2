"""Reinhard fp32 against the CPU oracle: max and mean absolute error of one build (STAINX_HIP_LIB selects it).
    python tools/check_reinhard_error.py"""
import sys, json, numpy as np, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from oracle import stain_oracle as so
from stainx_amd import Reinhard, synth
dev = torch.device("cuda:0")
x = synth.as_dtype(synth.noise_u8((4, 3, 256, 256), 43), torch.float32)
ref = synth.as_dtype(synth.noise_u8((1, 3, 256, 256), 42), torch.float32)
rn = Reinhard(device=dev).fit(ref.to(dev))
out = rn.transform(x.to(dev)).cpu().numpy().astype(np.float64)
fn = [n for n in dir(so) if "reinhard" in n.lower()]
m, s = so.reinhard_fit(ref.numpy())
want = so.reinhard_transform(x.numpy(), m, s).astype(np.float64)
d = np.abs(out - want)
print(json.dumps({"max_abs": float(d.max()), "mean_abs": float(d.mean()), "oracle_functions": fn}))
