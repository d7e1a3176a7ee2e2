"""A/B of two builds (STAINX_HIP_LIB) over ROTATING input batches for the paths whose last pass now loads its input non-temporally:
Reinhard f32, histogram matching u8, Macenko u8 / bf16 (four-pass) and f32.  One buffer against a rotation of three batches.
    python tools/ab_rotating_siblings.py   (run once per library)"""
import json, sys, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from stainx_amd import HistogramMatching, Macenko, Reinhard, synth
dev = torch.device("cuda:0")


def timed(fn, batches, steps=200, warm=30):
    for i in range(warm): fn(batches[i % len(batches)])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps): fn(batches[i % len(batches)])
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / steps * 1e3, 1)


rows = {}
for name, make, fit, shape in (("reinhard f32 64x512x512", lambda s: synth.as_dtype(synth.noise_u8((64, 3, 512, 512), s), torch.float32), lambda: Reinhard(device=dev).fit(synth.as_dtype(synth.noise_u8((1, 3, 512, 512), 1), torch.float32).to(dev)), None),
                               ("hm u8 64x1024x1024", lambda s: synth.noise_u8((64, 3, 1024, 1024), s), lambda: HistogramMatching(device=dev).fit(synth.noise_u8((1, 3, 1024, 1024), 1).to(dev)), None),
                               ("macenko u8 64x512x512", lambda s: synth.he_batch(64, 512, 512, seed0=1000 * s), lambda: Macenko(device=dev).fit(synth.reference_tile(512, 512).to(dev)), None),
                               ("macenko bf16 256x224x224", lambda s: synth.as_dtype(synth.he_batch(256, 224, 224, seed0=1000 * s), torch.bfloat16), lambda: Macenko(device=dev).fit(synth.reference_tile(224, 224).to(dev)), None),
                               ("macenko f32 64x512x512", lambda s: synth.as_dtype(synth.he_batch(64, 512, 512, seed0=1000 * s), torch.float32), lambda: Macenko(device=dev).fit(synth.reference_tile(512, 512).to(dev)), None)):
    norm = fit()
    batches = [make(s).to(dev) for s in (2, 3, 4)]
    rows[name] = {"one_buffer_us": timed(norm.transform, batches[:1]), "rotating3_us": timed(norm.transform, batches)}
    del batches
print(json.dumps(rows))
