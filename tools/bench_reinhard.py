"""Reinhard 64x3x512x512 fp32 / uint8 alone: time per call and a checksum of the output (to compare two builds bit for bit).
    python tools/bench_reinhard.py"""
import sys, json, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from stainx_amd import Reinhard, synth
dev = torch.device("cuda:0")
for dt in (torch.float32, torch.uint8, torch.bfloat16):
    x = synth.as_dtype(synth.noise_u8((64, 3, 512, 512), 43), dt).to(dev)
    rn = Reinhard(device=dev).fit(synth.as_dtype(synth.noise_u8((1, 3, 512, 512), 42), dt).to(dev))
    for _ in range(10): out = rn.transform(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): out = rn.transform(x)
    e1.record(); torch.cuda.synchronize()
    import hashlib
    h = hashlib.sha256(out.cpu().contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()[:16]
    print(json.dumps({"dtype": str(dt), "reinhard_us": round(e0.elapsed_time(e1) * 10, 1), "sha": h}))
