import sys, json, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from stainx_amd import synth, _native
from stainx_amd.backends.torch_hip_backend import MacenkoHIP
dev = torch.device("cuda:0")
x = synth.as_dtype(synth.he_batch(64, 512, 512), torch.float32).to(dev)
be = MacenkoHIP(dev)
sm = torch.tensor(synth.HE_REF).to(dev); tmc = torch.tensor([1.9705, 1.0308]).to(dev)
for _ in range(20): be.transform(x, sm, tmc, _extra_flags=_native.MACENKO_TWO_PASS)
torch.cuda.synchronize()
p = be.tile_params(64)
s = p["stamps_us"]
torch.set_printoptions(linewidth=250, precision=1, sci_mode=False)
print("cols: 8 9 10 11 12 | 1 barrier 2 prefix-issued 3 poll done 4 pinv done | 13 14 15")
print(torch.cat([s[:, 8:13], s[:, 1:5], s[:, 13:16]], 1)[::3])
