"""Kernel-trace target: the two-pass form forced on a batch of one element type (64x3x512x512, or N H W given).
   ... -- python3 tools/prof_twopass_dtype.py u8|bf16|f32 [N H W]"""
import sys, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from stainx_amd import synth, _native
from stainx_amd.backends.torch_hip_backend import MacenkoHIP
dev = torch.device("cuda:0")
dt = {"u8": torch.uint8, "bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[sys.argv[1] if len(sys.argv) > 1 else "u8"]
n, h, w = (int(v) for v in sys.argv[2:5]) if len(sys.argv) >= 5 else (64, 512, 512)
x = synth.as_dtype(synth.he_batch(n, h, w), dt).to(dev)
be = MacenkoHIP(dev)
sm = torch.tensor(synth.HE_REF).to(dev); tmc = torch.tensor([1.9705, 1.0308]).to(dev)
for _ in range(100): be.transform(x, sm, tmc, _extra_flags=_native.MACENKO_TWO_PASS)
torch.cuda.synchronize()
