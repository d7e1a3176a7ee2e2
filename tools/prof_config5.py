"""Kernel trace target: configs[4]'s shape (256x3x224x224 bf16 through the module) and 64x3x512x512 uint8, nothing else.
   rocprofv3 --kernel-trace --stats --output-format csv -d out -o kt -- python3 tools/prof_config5.py [bf16|u8]"""
import sys, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from stainx_amd import StainNormalizerTransform, synth
from stainx_amd.backends.torch_hip_backend import MacenkoHIP
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "bf16"
if which == "bf16":
    tiles = synth.as_dtype(synth.he_batch(256, 224, 224), torch.bfloat16).to(dev)
    t = StainNormalizerTransform(method="macenko", mode="reference", reference=synth.as_dtype(synth.reference_tile(224, 224), torch.bfloat16).to(dev))
    fn = lambda: t(tiles)
else:
    be = MacenkoHIP(dev)
    he, mc = be.compute_reference_stain_matrix(synth.reference_tile(512, 512).to(dev))
    x = synth.he_batch(64, 512, 512).to(dev)
    fn = lambda: be.transform(x, he, mc)
for _ in range(100):
    fn()
torch.cuda.synchronize()
