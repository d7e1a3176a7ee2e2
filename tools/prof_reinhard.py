"""Kernel-trace target: Reinhard 64x3x512x512 of one element type.   ... -- python3 tools/prof_reinhard.py f32|u8|bf16"""
import sys, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from stainx_amd import Reinhard, synth
dev = torch.device("cuda:0")
dt = {"u8": torch.uint8, "bf16": torch.bfloat16, "f32": torch.float32}[sys.argv[1] if len(sys.argv) > 1 else "f32"]
x = synth.as_dtype(synth.noise_u8((64, 3, 512, 512), 43), dt).to(dev)
rn = Reinhard(device=dev).fit(synth.as_dtype(synth.noise_u8((1, 3, 512, 512), 42), dt).to(dev))
for _ in range(100): rn.transform(x)
torch.cuda.synchronize()
