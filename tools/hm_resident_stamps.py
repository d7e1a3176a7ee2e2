"""Phase stamps of the one-launch histogram matching (diagnostic build): STAINX_DIAG=1 python tools/hm_resident_stamps.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from stainx_amd import synth
from stainx_amd.backends.torch_hip_backend import HistogramMatchingHIP

dev = torch.device("cuda", 0)
be = HistogramMatchingHIP(dev, diag=True)
xs = [synth.he_batch(64, 1024, 1024, seed0=1000 + 64 * b).to(dev) for b in range(2)]
h = torch.rand(256) + 0.01
ref = [(h / h.sum()).to(dev)] * 3
for i in range(10):
    be.transform(xs[i % 2], ref)
torch.cuda.synchronize()
ws = be.last_workspace
off = int(be._lib.sx_debug_hm_stamp_offset())
st = ws[off:off + 64].view(torch.int64).cpu().tolist()
names = ["start", "counted", "flushed", "grid sync", "tables", "kept packs written", "end"]
for i in range(1, 7):
    print(f"{names[i]:22s} +{(st[i] - st[i - 1]) / 100.0:8.1f} us   at {(st[i] - st[0]) / 100.0:8.1f} us")
