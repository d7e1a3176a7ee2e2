"""Does a HIP graph shorten the call?  The two-pass transform (4 launches) eager vs replayed from a captured graph.
    python tools/bench_graph.py"""
import sys, json, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from stainx_amd import synth, _native
from stainx_amd.backends.torch_hip_backend import MacenkoHIP

dev = torch.device("cuda:0")
x = synth.as_dtype(synth.he_batch(64, 512, 512), torch.float32).to(dev)
be = MacenkoHIP(dev)
sm = torch.tensor(synth.HE_REF).to(dev); tmc = torch.tensor([1.9705, 1.0308]).to(dev)
F = _native.MACENKO_TWO_PASS

def timed(fn, steps=300, warm=30):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3

eager = timed(lambda: be.transform(x, sm, tmc, _extra_flags=F))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): out = be.transform(x, sm, tmc, _extra_flags=F)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    out = be.transform(x, sm, tmc, _extra_flags=F)
graph = timed(g.replay)
ref = be.transform(x, sm, tmc, _extra_flags=F)
print(json.dumps({"eager_us": round(eager, 1), "graph_replay_us": round(graph, 1), "same_bits": bool(torch.equal(ref.view(torch.uint8), out.view(torch.uint8)))}))
