"""Kernel-trace target: histogram matching configs[2] (64x3x1024x1024 uint8).   ... -- python3 tools/prof_hm.py"""
import sys, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from stainx_amd import HistogramMatching, synth
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(43)
src = (torch.rand(64, 3, 1024, 1024, generator=g) * 255).round().to(torch.uint8).to(dev)
hm = HistogramMatching(device=dev).fit(synth.noise_u8((1, 3, 1024, 1024), 42).to(dev))
for _ in range(100): hm.transform(src)
torch.cuda.synchronize()
