"""Kernel-trace target: StainNormalizerTransform in batch mode (the batch is its own reference: fit + transform per call) on the config-2 batch.
    ... -- python3 tools/prof_batch_mode.py [macenko|reinhard|histogram_matching]"""
import sys, time, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from stainx_amd import StainNormalizerTransform, synth
dev = torch.device("cuda:0")
method = sys.argv[1] if len(sys.argv) > 1 else "macenko"
x = synth.as_dtype(synth.he_batch(64, 512, 512), torch.float32).to(dev)
t = StainNormalizerTransform(method=method, mode="batch")
for _ in range(10): t(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): t(x)
torch.cuda.synchronize()
print(f"{method} batch mode: {(time.perf_counter() - t0) * 1e4:.1f} us per call (wall)")
