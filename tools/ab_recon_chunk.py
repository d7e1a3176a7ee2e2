import os, sys, torch, json
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
for chunk in ("0", "2048", "1024", "2048", "1024", "0"):
    os.environ["SX_RECON_CHUNK"] = chunk
    print(chunk, "rotating", timed([a, b]), "one buffer", timed([a]), flush=True)
