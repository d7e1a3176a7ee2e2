"""Work items per tile of the transform on a debug build (SX_ITEMS; tools/build_debug.sh): 64 x 512 x 512 f32, rotating two batches
and over one buffer, the result checked bit for bit against the four-pass form at the library's own split.
    STAINX_HIP_LIB=stainx_amd/_lib/libstainx_dbg.so python tools/ab_items.py 16 20 16 20"""
import os, sys, torch
sys.path.insert(0, '.')
from stainx_amd import synth, _native
from stainx_amd.backends.torch_hip_backend import MacenkoHIP
dev = torch.device("cuda:0")
be = MacenkoHIP(dev)
sm = torch.tensor(synth.HE_REF).to(dev); tmc = torch.tensor([1.9705, 1.0308]).to(dev)
a = synth.as_dtype(synth.he_batch(64, 512, 512), torch.float32).to(dev)
b = synth.as_dtype(synth.he_batch(64, 512, 512, seed0=5000), torch.float32).to(dev)
def timed(batches, steps=300, warm=30):
    for i in range(warm): be.transform(batches[i % len(batches)], sm, tmc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps): be.transform(batches[i % len(batches)], sm, tmc)
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / steps * 1e3, 1)
for items in sys.argv[1:] or ("16", "20"):
    os.environ["SX_ITEMS"] = items
    two = be.transform(a, sm, tmc)
    four = be.transform(a, sm, tmc, _extra_flags=_native.MACENKO_CLASSIC)
    same = torch.equal(two.view(torch.int32), four.view(torch.int32))
    print(items, "rotating", timed([a, b]), "one buffer", timed([a]), "two-pass == four-pass:", same, flush=True)
