import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import stain_oracle as so
from stainx_amd import synth
from stainx_amd.backends.torch_hip_backend import MacenkoHIP
dev=torch.device('cuda:0')
be=MacenkoHIP(dev)
for n in (16, 32, 64):
    tiles=synth.he_batch(n,512,512)
    for dt in (torch.uint8, torch.float32):
        x=synth.as_dtype(tiles,dt).to(dev)
        he,mc=be.compute_reference_stain_matrix(x)
        p=be.tile_params(1)
        print(n, dt, he.cpu().numpy().round(5).tolist(), mc.cpu().numpy().round(5).tolist(), 'fell_back', int(p['fell_back'][0]), 'n_kept', int(p['n_kept'][0]))
    ho,mo=so.macenko_fit(tiles.numpy(), signs="positive_sum")
    print(n,'oracle', ho.round(5).tolist(), mo.round(5).tolist())
