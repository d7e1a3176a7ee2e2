"""A few resident-form transforms of one shape, for rocprofv3 (kernel trace / PMC passes): python3 tools/prof_resident.py [n hw dtype steps]"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from stainx_amd import _native, synth  # noqa: E402
from stainx_amd.backends.torch_hip_backend import MacenkoHIP  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dt = {"f32": torch.float32, "u8": torch.uint8, "bf16": torch.bfloat16}[sys.argv[3] if len(sys.argv) > 3 else "f32"]
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda:0")
be = MacenkoHIP(dev)
SM = torch.tensor(synth.HE_REF, dtype=torch.float32)
TMC = torch.tensor([1.9705, 1.0308], dtype=torch.float32)
x = synth.as_dtype(synth.he_batch(n, hw, hw, seed0=5), dt).to(dev)
for _ in range(steps):
    be.transform(x, SM, TMC, _extra_flags=_native.MACENKO_RESIDENT)
torch.cuda.synchronize()
print("done")
