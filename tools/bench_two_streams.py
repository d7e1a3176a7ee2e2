"""Throughput of the Macenko transform when ONE host thread feeds TWO streams with half a batch each (no dependency between the
streams: the tiles are independent), against one stream with whole batches: do the latency-bound launches of one half (prior,
per-tile stage) hide behind the streaming launches of the other?   64 x 3 x 512 x 512 float32, rotating input batches.
    python tools/bench_two_streams.py"""
import json, sys, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from stainx_amd import synth
from stainx_amd.backends.torch_hip_backend import MacenkoHIP
dev = torch.device("cuda:0")
sm = torch.tensor(synth.HE_REF).to(dev); tmc = torch.tensor([1.9705, 1.0308]).to(dev)
batches = [synth.as_dtype(synth.he_batch(64, 512, 512, seed0=1000 * s), torch.float32).to(dev) for s in (1, 2)]


def run(n_streams, steps=300, warm=30):
    be = MacenkoHIP(dev)
    streams = [torch.cuda.Stream(dev) for _ in range(n_streams)]
    per = 64 // n_streams
    parts = [[b[i * per:(i + 1) * per].contiguous() for i in range(n_streams)] for b in batches]

    def step(k):
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                be.transform(parts[k % 2][i], sm, tmc)

    for k in range(warm): step(k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()      # (on the default stream: nothing else is queued there; the synchronize above and below bracket the work)
    import time
    t0 = time.perf_counter()
    for k in range(steps): step(k)
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / steps * 1e6, 1)


for n in (1, 2, 4, 1, 2, 4):
    print(json.dumps({"streams": n, "tiles_per_call": 64 // n, "us_per_64_tiles": run(n)}), flush=True)
