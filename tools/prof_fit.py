import sys, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from stainx_amd import synth
from stainx_amd.backends.torch_hip_backend import MacenkoHIP
dev = torch.device("cuda:0")
be = MacenkoHIP(dev)
x = synth.as_dtype(synth.he_batch(64, 512, 512), torch.float32).to(dev)
for _ in range(5):
    he, mc = be.compute_reference_stain_matrix(x)
torch.cuda.synchronize()
p = be.tile_params(1)
print("fell_back", p["fell_back"], "ncand", p["n_candidates"])
