"""Slow-path slots of the two-pass form against the stain density of the tiles (he_scale of stainx_amd.synth.he_tile): dark,
partly saturated tiles are where the presample's frame stops being trustworthy.   python tools/sweep_scale.py"""
import sys, json, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from stainx_amd import synth, _native
from stainx_amd.backends.torch_hip_backend import MacenkoHIP

dev = torch.device("cuda:0")
be = MacenkoHIP(dev)
sm = torch.tensor(synth.HE_REF).to(dev); tmc = torch.tensor([1.9705, 1.0308]).to(dev)
for scale in (0.5, 1.0, 1.5, 2.0, 2.5, 3.0, 4.0, 6.0):
    u8 = torch.cat([synth.he_tile(512, 512, 7000 + i, scale) for i in range(32)], 0)
    x = synth.as_dtype(u8, torch.float32).to(dev)
    two = be.transform(x, sm, tmc, _extra_flags=_native.MACENKO_TWO_PASS)
    p = be.tile_params(32)
    classic = be.transform(x, sm, tmc, _extra_flags=_native.MACENKO_CLASSIC)
    fb = p["fell_back"].to(torch.int64)
    why = {}
    for t in range(32):
        if int(fb[t]) & 15:
            why[hex(int(fb[t]) >> 8)] = why.get(hex(int(fb[t]) >> 8), 0) + 1
    print(json.dumps({"he_scale": scale, "black_px_pct": round(float((u8 == 0).float().mean()) * 100, 2), "slow_slots": int(sum(bin(int(v) & 15).count("1") for v in fb)), "of": 128, "why": why,
                      "cand_pct": [round(float(v), 2) for v in (p["n_candidates"].double().mean(0) / (512 * 512) * 100)], "bit_equal": bool(torch.equal(two.view(torch.uint8), classic.view(torch.uint8)))}))
