"""Host time of one pooled fit_transform step (one rank, collectives forced): how long the Python choreography takes to ENQUEUE a step
against how long the GPU takes to run it -- a step whose enqueue time exceeds its device time is host-bound whatever the kernels do.
    python tools/host_time_pooled.py [steps]          (prints the two times and cProfile's top functions by own time)"""
import cProfile
import os
import pstats
import sys
import time

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
os.environ["STAINX_FORCE_COLLECTIVES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
from stainx_amd import distributed as sxd, synth
from stainx_amd.backends.torch_hip_backend import MacenkoHIP

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
be = MacenkoHIP(dev)
batches = [synth.as_dtype(synth.he_batch(64, 512, 512, seed0=1000 + 64 * b), torch.float32).to(dev) for b in range(2)]


def run(k):
    for i in range(k):
        sxd.macenko_fit_transform_pooled(batches[i % 2], steps=be)


run(30)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(steps)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e6 * (t1 - t0) / steps:.1f} us/step, until the device is done {1e6 * (t2 - t0) / steps:.1f} us/step")
pr = cProfile.Profile()
pr.enable()
run(steps)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
dist.destroy_process_group()
