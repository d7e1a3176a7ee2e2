"""Which form of the Macenko transform wins where: four launches of the two-pass form against the four passes (and the fused launch
where it can run) over element types and tile sizes, synthetic tiles, one buffer.   python tools/survey_forms.py [out.jsonl]"""
import json, sys, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from stainx_amd import synth, _native
from stainx_amd.backends.torch_hip_backend import MacenkoHIP
dev = torch.device("cuda:0")
be = MacenkoHIP(dev)
sm = torch.tensor(synth.HE_REF).to(dev); tmc = torch.tensor([1.9705, 1.0308]).to(dev)
DT = {"f32": torch.float32, "u8": torch.uint8, "bf16": torch.bfloat16, "f16": torch.float16, "f64": torch.float64}


def timed(x, flags, steps=100, warm=20, reps=3):
    out = []      # (median of three: the boxes stall a process for ~0.1 s now and then, which is 0.5 ms per call in a 150-call loop)
    for _ in range(reps):
        for _ in range(warm): be.transform(x, sm, tmc, _extra_flags=flags)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps): be.transform(x, sm, tmc, _extra_flags=flags)
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / steps * 1e3)
    return round(sorted(out)[1], 1)


rows = []
for n, h, w in ((64, 512, 512), (114, 384, 384), (164, 320, 320), (256, 224, 224), (1024, 128, 128), (36, 724, 724), (16, 1024, 1024)):
    for name in ("f32", "u8", "bf16"):
        x = synth.as_dtype(synth.he_batch(n, h, w), DT[name]).to(dev)
        lib = _native.require()
        row = {"shape": [n, 3, h, w], "dtype": name, "default_form": lib.sx_macenko_form(_native.DTYPE_CODES[DT[name]], n, h, w, 0),
               "four_pass_us": timed(x, _native.MACENKO_CLASSIC), "two_pass_us": timed(x, _native.MACENKO_TWO_PASS)}
        if lib.sx_macenko_form(_native.DTYPE_CODES[DT[name]], n, h, w, _native.MACENKO_TWO_PASS | _native.MACENKO_FUSE) == 2:
            row["fused_us"] = timed(x, _native.MACENKO_TWO_PASS | _native.MACENKO_FUSE)
        row["default_is_the_faster"] = (row["default_form"] != 0) == (row["two_pass_us"] < row["four_pass_us"])
        print(json.dumps(row), flush=True)
        rows.append(row)
        del x
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        for r in rows: f.write(json.dumps(r) + "\n")
