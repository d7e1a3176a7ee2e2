"""Reinhard transform of a float32 batch with and without the tiles' 8-bit codes (workspace with / without room for them), device time per call.
    python tools/ab_reinhard_coded.py [calls]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from stainx_amd import _native, synth

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0)
lib = _native.require()
mean = torch.tensor([170.0, 150.0, 120.0], dtype=torch.float32, device=dev)
std = torch.tensor([40.0, 12.0, 9.0], dtype=torch.float32, device=dev)
f32 = _native.DTYPE_CODES[torch.float32]
for (n, h, w) in ((64, 512, 512), (256, 224, 224)):
    batches = [synth.as_dtype(synth.he_batch(n, h, w, seed0=1000 + n * b), torch.float32).to(dev) for b in range(2)]
    out = torch.empty_like(batches[0])
    for label, nbytes in (("float pixels in both passes", int(lib.sx_reinhard_workspace_bytes(n, h, w))), ("8-bit codes behind the statistics pass", int(lib.sx_reinhard_workspace_bytes_for(f32, n, h, w)))):
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        stream = _native.stream_ptr(dev)

        def call(i):
            x = batches[i % 2]
            rc = lib.sx_reinhard_transform_ready(x.data_ptr(), out.data_ptr(), f32, n, h, w, mean.data_ptr(), std.data_ptr(), ws.data_ptr(), ws.numel(), stream)
            assert rc == 0

        for i in range(20):
            call(i)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(calls)]
        for i in range(calls):
            ev[i][0].record()
            call(i)
            ev[i][1].record()
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) for a, b in ev)
        print(f"{n} x {h} x {w} f32  {label:42s} median {t[len(t) // 2] * 1e3:7.1f} us   min {t[0] * 1e3:7.1f} us", flush=True)
