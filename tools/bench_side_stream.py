"""What one cross-stream dependency per call costs: the config-2 transform in a loop, plain, and with a trivial kernel on a side
stream whose completion event the main stream waits for before every call (the shape of running the prior stage of call n+1
beside the tail of call n)."""
import sys, json, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from stainx_amd import synth, _native
from stainx_amd.backends.torch_hip_backend import MacenkoHIP
dev = torch.device("cuda:0")
x = synth.as_dtype(synth.he_batch(64, 512, 512), torch.float32).to(dev)
be = MacenkoHIP(dev)
sm = torch.tensor(synth.HE_REF).to(dev); tmc = torch.tensor([1.9705, 1.0308]).to(dev)
side = torch.cuda.Stream(dev)
small = torch.zeros(1024, device=dev)
def loop(mode, steps=300):
    for _ in range(30): be.transform(x, sm, tmc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        if mode:
            with torch.cuda.stream(side):
                small.add_(1.0)
                ev = torch.cuda.Event(); ev.record(side)
            torch.cuda.current_stream(dev).wait_event(ev)
        be.transform(x, sm, tmc)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3
print(json.dumps({"plain_us": round(loop(0), 1), "with_side_stream_dependency_us": round(loop(1), 1), "plain_again_us": round(loop(0), 1)}))
