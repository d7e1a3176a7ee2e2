"""A/B of the two-pass and the four-pass (SX_MACENKO_CLASSIC) transform: HIP-event time per call, candidates, fallbacks.
    python tools/bench_twopass.py [tiles H W dtype]"""
import sys, json, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from stainx_amd import synth, _native
from stainx_amd.backends.torch_hip_backend import MacenkoHIP

dev = torch.device("cuda:0")
n, h, w = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 512, 512)
dt = {"f32": torch.float32, "u8": torch.uint8, "bf16": torch.bfloat16, "f16": torch.float16}[sys.argv[4] if len(sys.argv) > 4 else "f32"]
x = synth.as_dtype(synth.he_batch(n, h, w), dt).to(dev)
be = MacenkoHIP(dev)
sm = torch.tensor(synth.HE_REF).to(dev); tmc = torch.tensor([1.9705, 1.0308]).to(dev)

def timed(flags, steps=300, warm=30):
    for _ in range(warm): be.transform(x, sm, tmc, _extra_flags=flags)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): out = be.transform(x, sm, tmc, _extra_flags=flags)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3, out

tf, of = timed(_native.MACENKO_TWO_PASS | _native.MACENKO_FUSE)         # the fused launch where it can run (opt-in)
form = _native.require().sx_macenko_form(_native.DTYPE_CODES[dt], n, h, w, _native.MACENKO_TWO_PASS | _native.MACENKO_FUSE)
t2, o2 = timed(_native.MACENKO_TWO_PASS)                                # the two-pass form as four launches
p2 = be.tile_params(n)
t1, o1 = timed(_native.MACENKO_CLASSIC)
same = torch.equal(o1.view(torch.uint8), o2.view(torch.uint8)) and torch.equal(o1.view(torch.uint8), of.view(torch.uint8))
px = n * h * w
print(json.dumps({"shape": [n, 3, h, w], "dtype": str(dt), "fused_form": form, "fused_us": round(tf, 1), "two_pass_us": round(t2, 1), "four_pass_us": round(t1, 1), "bitwise_equal": same,
                  "two_pass_MPs": round(px / t2, 0), "fell_back_tiles": int((p2["fell_back"] != 0).sum()),
                  "candidates_pct_per_slot": [round(float(v), 2) for v in (p2["n_candidates"].double().mean(0) / (h * w) * 100)],
                  "stamps_us_median(prior 0-7, phi 8-11, conc 12-15)": [round(float(v), 1) for v in p2["stamps_us"].median(0).values],
                  "conc_us_per_tile(sorted)": [round(float(v), 1) for v in (p2["stamps_us"][:, 15] - p2["stamps_us"][:, 12]).sort().values[::8]],
                  "phi_us_per_tile(sorted)": [round(float(v), 1) for v in (p2["stamps_us"][:, 11] - p2["stamps_us"][:, 8]).sort().values[::8]]}))
