import sys, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from stainx_amd import synth
from stainx_amd.backends.torch_hip_backend import MacenkoHIP
dev = torch.device("cuda:0")
be = MacenkoHIP(dev)
he, mc = be.compute_reference_stain_matrix(synth.reference_tile(256, 256).to(dev))
n, h = int(sys.argv[1]), int(sys.argv[2])
x = synth.as_dtype(synth.he_batch(n, h, h), torch.float32).to(dev)
for _ in range(12):
    out = be.transform(x, he, mc)
torch.cuda.synchronize()
p = be.tile_params(n)
print("ncand", p["n_candidates"][0].tolist(), "fell_back", p["fell_back"].tolist())
