import sys, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from stainx_amd import synth
from stainx_amd.backends.torch_hip_backend import MacenkoHIP
dev = torch.device("cuda:0")
be = MacenkoHIP(dev)
x = synth.as_dtype(synth.he_batch(64, 512, 512), torch.float32).to(dev)
he, mc = be.compute_reference_stain_matrix(synth.reference_tile(512, 512).to(dev))
for _ in range(3):
    out = be.transform(x, he, mc)
torch.cuda.synchronize()
p = be.tile_params(64)
torch.set_printoptions(precision=1, linewidth=250, sci_mode=False)
print("stamps us (tile 0, 1, 63):")
print(p["stamps_us"][[0, 1, 63]])
print("n_candidates", p["n_candidates"][[0, 1, 63]], "max", p["n_candidates"].max(0).values)
