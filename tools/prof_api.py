"""Kernel-trace target: the PUBLIC classes (not the backend objects) on the config-2 batch -- shows every kernel a user's call launches,
including any torch op the wrappers put in front of the library's.   ... -- python3 tools/prof_api.py macenko|reinhard|hm [dtype]"""
import sys, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from stainx_amd import HistogramMatching, Macenko, Reinhard, synth
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "macenko"
dt = {"u8": torch.uint8, "bf16": torch.bfloat16, "f32": torch.float32}[sys.argv[2] if len(sys.argv) > 2 else "f32"]
x = synth.as_dtype(synth.he_batch(64, 512, 512), dt).to(dev)
ref = synth.as_dtype(synth.reference_tile(512, 512), dt).to(dev)
norm = {"macenko": Macenko, "reinhard": Reinhard, "hm": HistogramMatching}[which](device=dev).fit(ref)
for _ in range(100): norm.transform(x)
torch.cuda.synchronize()
