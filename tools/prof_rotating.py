"""Kernel-trace target: the two-pass transform over N rotating input batches (N = 1: one buffer, Infinity-Cache resident).
    ... -- python3 tools/prof_rotating.py [batches]"""
import sys, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from stainx_amd import synth, _native
from stainx_amd.backends.torch_hip_backend import MacenkoHIP
dev = torch.device("cuda:0")
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2
xs = [synth.as_dtype(synth.he_batch(64, 512, 512, seed0=1000 + 100 * b), torch.float32).to(dev) for b in range(nb)]
be = MacenkoHIP(dev)
sm = torch.tensor(synth.HE_REF).to(dev); tmc = torch.tensor([1.9705, 1.0308]).to(dev)
for i in range(300): be.transform(xs[i % nb], sm, tmc, _extra_flags=_native.MACENKO_TWO_PASS)
torch.cuda.synchronize()
