import sys, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from stainx_amd import synth
from stainx_amd.backends.torch_hip_backend import MacenkoHIP
dev = torch.device("cuda:0")
be = MacenkoHIP(dev)
he, mc = be.compute_reference_stain_matrix(synth.reference_tile(512, 512).to(dev))
src = synth.he_batch(64, 512, 512)
for dt in (torch.uint8, torch.bfloat16):
    x = synth.as_dtype(src, dt).to(dev)
    for _ in range(20):
        out = be.transform(x, he, mc)
    torch.cuda.synchronize()
