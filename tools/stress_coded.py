#!/usr/bin/env python3
"""Randomised sweep of the coded passes (float32 tiles as 8-bit codes behind the first pass) against the same passes reading floats,
bit for bit (diagnostic build: SX_MACENKO_NO_CODES): random batch shapes around the sizes where the codes switch on, tiles that are grey
levels / are not / mix, unaligned views, both forms, /255; and the Reinhard transform with / without room for the codes.
    python tools/stress_coded.py [seed] [cases]"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from stainx_amd import _native, synth  # noqa: E402
from stainx_amd.backends.torch_hip_backend import MacenkoHIP  # noqa: E402

dev = torch.device("cuda:0")
be = MacenkoHIP(dev, diag=True)
lib = _native.require()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
sm = torch.tensor([[0.5626, 0.2159], [0.7201, 0.8012], [0.4062, 0.5581]], dtype=torch.float32, device=dev)
tmc = torch.tensor([1.9705, 1.0308], dtype=torch.float32, device=dev)
mean = torch.tensor([170.0, 150.0, 120.0], dtype=torch.float32, device=dev)
std = torch.tensor([40.0, 12.0, 9.0], dtype=torch.float32, device=dev)
C, TP, NC = _native.MACENKO_CLASSIC, _native.MACENKO_TWO_PASS, _native.MACENKO_NO_CODES
f32 = _native.DTYPE_CODES[torch.float32]
bad = 0
for case in range(cases):
    h = int(rng.choice([64, 96, 128, 200, 224, 256, 300, 384, 512])) + int(rng.choice([0, 0, 0, 4, 8, 16, 1]))
    w = int(rng.choice([64, 128, 224, 256, 320, 512])) + int(rng.choice([0, 0, 0, 4, 16, 3]))
    n = max(1, int(rng.integers(1 << 19, 3 << 20) // (h * w)))
    n = min(n, 96)
    x = synth.as_dtype(synth.he_batch(n, h, w, seed0=int(rng.integers(0, 1 << 20)), scale_step=float(rng.uniform(0, 0.1))), torch.float32)
    kind = int(rng.integers(0, 4))
    if kind == 1:      # one tile off the grey levels
        t = int(rng.integers(0, n))
        x[t] = (x[t] + 3e-4).clamp(0.0, 1.0)
    elif kind == 2:    # a few odd elements
        for _ in range(int(rng.integers(1, 5))):
            x[int(rng.integers(0, n)), int(rng.integers(0, 3)), int(rng.integers(0, h)), int(rng.integers(0, w))] = float(rng.uniform(0, 1))
    elif kind == 3:    # nothing is a grey level
        x = (x * 0.983).contiguous()
    unit = bool(rng.integers(0, 2))
    xin = x.to(dev)
    if rng.integers(0, 4) == 0:      # an unaligned view (offset by one element)
        buf = torch.empty(xin.numel() + 1, dtype=torch.float32, device=dev)
        buf[1:].copy_(xin.flatten())
        xin = buf[1:].view_as(xin)
    res = {}
    for name, flags in (("four", C), ("four_plain", C | NC), ("two", TP), ("two_plain", TP | NC)):
        res[name] = be.transform(xin, sm, tmc, normalize_to_0_1=unit, _extra_flags=flags)
    ok = torch.equal(res["four"], res["four_plain"]) and torch.equal(res["two"], res["two_plain"]) and torch.equal(res["four"], res["two"])
    # Reinhard through the C ABI, workspace with and without room for the codes
    outs = []
    xr = xin.contiguous()
    for nbytes in (int(lib.sx_reinhard_workspace_bytes_for(f32, n, h, w)), int(lib.sx_reinhard_workspace_bytes(n, h, w))):
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        out = torch.empty_like(xr)
        rc = lib.sx_reinhard_transform(xr.data_ptr(), out.data_ptr(), f32, n, h, w, mean.data_ptr(), std.data_ptr(), ws.data_ptr(), ws.numel(), _native.stream_ptr(dev))
        assert rc == 0
        outs.append(out)
    torch.cuda.synchronize()
    ok_r = torch.equal(outs[0], outs[1])
    if not (ok and ok_r):
        bad += 1
        print("MISMATCH", dict(case=case, n=n, h=h, w=w, kind=kind, unit=unit, macenko=ok, reinhard=ok_r), flush=True)
print(f"{cases} cases, {bad} mismatches", flush=True)
