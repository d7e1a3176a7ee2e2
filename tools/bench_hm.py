"""Histogram matching configs[2] (64x3x1024x1024 uint8) alone: time per call.   python tools/bench_hm.py"""
import sys, json, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from stainx_amd import HistogramMatching, synth
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(43)
src = (torch.rand(64, 3, 1024, 1024, generator=g) * 255).round().to(torch.uint8).to(dev)
hm = HistogramMatching(device=dev).fit(synth.noise_u8((1, 3, 1024, 1024), 42).to(dev))
for _ in range(10): out = hm.transform(src)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): out = hm.transform(src)
e1.record(); torch.cuda.synchronize()
print(json.dumps({"hm_config3_us": round(e0.elapsed_time(e1) * 10, 1), "checksum": int(out.to(torch.int64).sum())}))
