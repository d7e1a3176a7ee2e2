"""Sweep of the two-pass speculation's knobs on real tissue and on the synthetic batch (diagnostic build: tools/build_debug.sh reads
SX_SPEC_* from the environment at every call).  Per setting: slots that left the speculative path, candidates per slot, time per call.
    STAINX_HIP_LIB=stainx_amd/_lib/libstainx_dbg.so python tools/tune_spec.py [out.jsonl]"""
import itertools, json, os, sys
import numpy as np, torch
root = __import__("pathlib").Path(__file__).resolve().parents[1]
sys.path.insert(0, str(root))
from stainx_amd import synth, _native
from stainx_amd.backends.torch_hip_backend import MacenkoHIP

dev = torch.device("cuda:0")
imgs = torch.from_numpy(np.load(str(root / "tests/golden/g11_real_images.npz"))["images_u8"])
be = MacenkoHIP(dev)
sm, tmc = be.compute_reference_stain_matrix(imgs[0:1].to(dev))


def crops(size, stride, images):
    return torch.stack([imgs[i, :, y:y + size, x:x + size] for i in images for y in range(0, 1024 - size + 1, stride) for x in range(0, 1024 - size + 1, stride)])


all512 = crops(512, 128, range(6))
batches = {"real64": synth.as_dtype(all512[torch.arange(0, 150, 150 / 64).long()], torch.float32).to(dev),
           "real_rest": synth.as_dtype(all512[[i for i in range(150) if i not in set(torch.arange(0, 150, 150 / 64).long().tolist())][:64]], torch.float32).to(dev),
           "synth64": synth.as_dtype(synth.he_batch(64, 512, 512), torch.float32).to(dev)}
TP = _native.MACENKO_TWO_PASS


def timed(x, flags, steps=60, warm=10):
    for _ in range(warm): be.transform(x, sm, tmc, _extra_flags=flags)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): be.transform(x, sm, tmc, _extra_flags=flags)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


grid = [dict(zip(("SX_SPEC_KW", "SX_SPEC_ROT", "SX_SPEC_EFF_FAR", "SX_SPEC_SIGMAS", "SX_SPEC_SIGMAS_CONC", "SX_SPEC_TSCALE"), v)) for v in
        [(0.10, 0.3, 2.0, 5.0, 5.0, 1.0), (0.10, 0.3, 2.0, 5.0, 8.0, 1.0), (0.10, 0.3, 2.0, 5.0, 12.0, 1.0), (0.10, 0.3, 2.0, 5.0, 5.0, 0.97), (0.10, 0.3, 2.0, 5.0, 5.0, 0.94),
         (0.10, 0.3, 2.0, 5.0, 8.0, 0.94), (0.10, 0.3, 2.0, 4.0, 8.0, 0.94), (0.12, 0.3, 2.0, 4.0, 12.0, 0.90)]]
out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
for knobs in grid:
    for k, v in knobs.items(): os.environ[k] = str(v)
    os.environ["SX_SPEC_EFF_NEAR"] = str(knobs["SX_SPEC_EFF_FAR"] / 2)
    row = dict(knobs)
    for name, x in batches.items():
        be.transform(x, sm, tmc, _extra_flags=TP)
        p = be.tile_params(x.shape[0])
        fb = p["fell_back"] & 15
        why = [[int(((p["fell_back"] >> (8 + 4 * s)) & 15).eq(w).sum()) for w in (1, 2, 3, 4)] for s in range(4)]
        pct = p["n_candidates"].double() / (512 * 512) * 100
        row[name] = {"slow_slots": int(sum(int(((fb >> s) & 1).sum()) for s in range(4))), "slow_tiles": int((fb != 0).sum()), "why(1,2,3,4)_per_slot": why,
                     "cand_pct_median": [round(float(pct[:, s].median()), 2) for s in range(4)], "cand_pct_max": [round(float(pct[:, s].max()), 2) for s in range(4)],
                     "two_pass_us": round(timed(x, TP), 1)}
    print(json.dumps(row), flush=True)
    if out: out.write(json.dumps(row) + "\n")
