"""Real-tissue speculation sweep: batches cut from the reference's six example images
(tests/golden/g11_real_images.npz): for the two-pass forms, how many slots leave the speculative path, candidates per slot, time
per call against the four-pass form -- one input buffer and rotating over two different batches -- and bitwise equality.
    python tools/sweep_real.py [out.jsonl]"""
import json, sys
import numpy as np, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from stainx_amd import synth, _native
from stainx_amd.backends.torch_hip_backend import MacenkoHIP

dev = torch.device("cuda:0")
imgs = torch.from_numpy(np.load(str(__import__("pathlib").Path(__file__).resolve().parents[1] / "tests/golden/g11_real_images.npz"))["images_u8"])
be = MacenkoHIP(dev)
fit = be.compute_reference_stain_matrix(imgs[0:1].to(dev))
sm, tmc = fit[0], fit[1]
out_path = sys.argv[1] if len(sys.argv) > 1 else None
rows = []


def crops(size, stride, images):
    t = []
    for i in images:
        for y in range(0, 1024 - size + 1, stride):
            for x in range(0, 1024 - size + 1, stride):
                t.append(imgs[i, :, y:y + size, x:x + size])
    return torch.stack(t)


def timed(batches, flags, steps=200, warm=20, **kw):
    for i in range(warm): be.transform(batches[i % len(batches)], sm, tmc, _extra_flags=flags, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps): out = be.transform(batches[i % len(batches)], sm, tmc, _extra_flags=flags, **kw)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


def run(tag, tiles, dt, forms):
    n, _, h, w = tiles.shape
    a = synth.as_dtype(tiles, dt).to(dev)
    b = synth.as_dtype(tiles.flip(0).flip(3).contiguous(), dt).to(dev)      # a second, different batch for the rotation
    row = {"case": tag, "shape": [n, 3, h, w], "dtype": str(dt).replace("torch.", "")}
    ref = be.transform(a, sm, tmc, _extra_flags=_native.MACENKO_CLASSIC)
    for name, flags in forms.items():
        out = be.transform(a, sm, tmc, _extra_flags=flags)
        p = be.tile_params(n)
        row[f"{name}_bitwise_equal_to_four_pass"] = bool(torch.equal(out.view(torch.uint8), ref.view(torch.uint8)))
        if name != "four_pass":
            fb = p["fell_back"] & 15
            row[f"{name}_slow_slots"] = int(sum(int(((fb >> s) & 1).sum()) for s in range(4)))
            row[f"{name}_slow_tiles"] = (fb != 0).nonzero().flatten().tolist()
            pct = p["n_candidates"].double() / (h * w) * 100
            row[f"{name}_candidates_pct_per_slot(min,median,max over tiles)"] = [[round(float(v), 2) for v in (pct[:, s].min(), pct[:, s].median(), pct[:, s].max())] for s in range(4)]
        row[f"{name}_us_one_buffer"] = round(timed([a], flags), 1)
        row[f"{name}_us_rotating"] = round(timed([a, b], flags), 1)
    print(json.dumps(row), flush=True)
    rows.append(row)


TP, FU, CL = _native.MACENKO_TWO_PASS, _native.MACENKO_TWO_PASS | _native.MACENKO_FUSE, _native.MACENKO_CLASSIC
all512 = crops(512, 128, range(6))                      # 150 overlapping tiles
pick = torch.arange(0, 150, 150 / 64).long()
run("real 64x512x512 (all six images)", all512[pick], torch.float32, {"four_pass": CL, "two_pass": TP, "fused": FU})
run("real 64x512x512 (all six images)", all512[pick], torch.uint8, {"four_pass": CL, "two_pass": TP})
run("real 50x512x512 tissue-rich (target, test_1, test_3)", crops(512, 128, (0, 1, 3))[:50], torch.float32, {"four_pass": CL, "two_pass": TP, "fused": FU})
run("real 50x512x512 background-heavy (test_4, test_5)", crops(512, 128, (4, 5)), torch.float32, {"four_pass": CL, "two_pass": TP, "fused": FU})
c224 = crops(224, 100, range(6))                        # 9 x 9 x 6 = 486 crops
run("real 256x224x224 (configs[4] shape)", c224[torch.arange(0, 486, 486 / 256).long()], torch.bfloat16, {"four_pass": CL, "two_pass": TP})
run("real 256x224x224 (configs[4] shape)", c224[torch.arange(0, 486, 486 / 256).long()], torch.float32, {"four_pass": CL, "two_pass": TP, "fused": FU})
synth_tiles = synth.he_batch(64, 512, 512)
run("synthetic 64x512x512 (bench.py's batch)", synth_tiles, torch.float32, {"four_pass": CL, "two_pass": TP, "fused": FU})
if out_path:
    with open(out_path, "w") as f:
        for r in rows: f.write(json.dumps(r) + "\n")
