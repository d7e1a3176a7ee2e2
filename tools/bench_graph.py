"""Does a HIP graph shorten the call?  The transform eager vs replayed from a captured graph: the two-pass form (4 launches) on
64 x 512 x 512, or the default form of any N H W given (small batches: seven launches of the four-pass form).
    python tools/bench_graph.py [N H W]"""
import sys, json, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from stainx_amd import synth, _native
from stainx_amd.backends.torch_hip_backend import MacenkoHIP

dev = torch.device("cuda:0")
shape = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (64, 512, 512)
x = synth.as_dtype(synth.he_batch(*shape), torch.float32).to(dev)
be = MacenkoHIP(dev)
sm = torch.tensor(synth.HE_REF).to(dev); tmc = torch.tensor([1.9705, 1.0308]).to(dev)
F = _native.MACENKO_TWO_PASS if len(sys.argv) < 4 else _native.MACENKO_CLASSIC

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
print(json.dumps({"shape": list(shape), "eager_us": round(eager, 1), "graph_replay_us": round(graph, 1), "same_bits": bool(torch.equal(ref.view(torch.uint8), out.view(torch.uint8)))}))
