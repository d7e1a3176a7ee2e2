"""How often does a concentration slot's cone (built from the sample's extreme, macenko_twopass.hpp prior_kernel) miss the exact
stain vector?  Many different 512x512 and 256x256 tiles through the two-pass form, counting slots on the slow path and why.
    python tools/sweep_cones.py [seeds]"""
import sys, json, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from stainx_amd import synth, _native
from stainx_amd.backends.torch_hip_backend import MacenkoHIP

dev = torch.device("cuda:0")
be = MacenkoHIP(dev)
sm = torch.tensor(synth.HE_REF).to(dev); tmc = torch.tensor([1.9705, 1.0308]).to(dev)
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
tiles = slow = 0
why = {}
cand = []
for seed in range(seeds):
    for (n, hw) in ((32, 512), (64, 256)):
        x = synth.as_dtype(synth.he_batch(n, hw, hw, seed0=100000 + 1000 * seed, scale_step=0.002 * (seed % 5)), torch.float32).to(dev)
        be.transform(x, sm, tmc, _extra_flags=_native.MACENKO_TWO_PASS)
        p = be.tile_params(n)
        fb = p["fell_back"].to(torch.int64)
        tiles += n
        for t in range(n):
            if int(fb[t]) & 15:
                slow += bin(int(fb[t]) & 15).count("1")
                code = hex(int(fb[t]) >> 8)
                why[code] = why.get(code, 0) + 1
        cand.append((p["n_candidates"].double() / (hw * hw) * 100).mean(0))
print(json.dumps({"tiles": tiles, "slots": 4 * tiles, "slow_slots": slow, "why": why, "candidates_pct_per_slot": [round(float(v), 2) for v in torch.stack(cand).mean(0)]}))
