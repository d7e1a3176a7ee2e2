"""the small-batch rows of the round-2 survey were 5-14 % slower than round 1's.  A/B on ONE box of the round-1,
round-2 and current libraries through the C ABI alone (ctypes, no Python backend: isolates the kernels + the library's host code),
and of the current Python backend on top (isolates the host routing).  HIP-event time per call, median of 5 repeats of 200 calls.
    python tools/ab_small_batches.py [out.jsonl]   (libraries: stainx_amd/_lib/libstainx_{r01,r02,hip}.so)"""
import ctypes, json, sys, statistics
import torch
root = __import__("pathlib").Path(__file__).resolve().parents[1]
sys.path.insert(0, str(root))
from stainx_amd import synth, _native
from stainx_amd.backends.torch_hip_backend import MacenkoHIP

dev = torch.device("cuda:0")
DT = {"f32": (torch.float32, 3), "f64": (torch.float64, 4), "u8": (torch.uint8, 0), "bf16": (torch.bfloat16, 2)}
sm = torch.tensor(synth.HE_REF).to(dev); tmc = torch.tensor([1.9705, 1.0308]).to(dev)
vp, i64 = ctypes.c_void_p, ctypes.c_int64


def load(name):
    p = root / "stainx_amd" / "_lib" / name
    if not p.exists():
        return None
    lib = ctypes.CDLL(str(p))
    lib.sx_macenko_workspace_bytes.restype = ctypes.c_size_t
    lib.sx_macenko_workspace_bytes.argtypes = [i64, i64, i64]
    lib.sx_macenko_transform.restype = ctypes.c_int
    lib.sx_macenko_transform.argtypes = [vp, vp, ctypes.c_int, i64, i64, i64, vp, vp, ctypes.c_uint, vp, ctypes.c_size_t, vp]
    return lib


libs = {k: load(v) for k, v in (("r01", "libstainx_r01.so"), ("r02", "libstainx_r02.so"), ("head", "libstainx_hip.so"))}


def timed(fn, steps=200, warm=30, reps=5):
    out = []
    for _ in range(reps):
        for _ in range(warm): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / steps * 1e3)
    return round(statistics.median(out), 1), round(max(out) / min(out), 2)


rows = []
for n, h, w, name, unit in ((1, 512, 512, "f32", 0), (7, 321, 199, "f32", 0), (7, 321, 199, "f64", 1), (2, 512, 512, "f32", 0), (1, 224, 224, "bf16", 1), (4, 512, 512, "u8", 0), (64, 512, 512, "u8", 0)):
    dt, code = DT[name]
    x = synth.as_dtype(synth.he_batch(n, h, w), dt).to(dev)
    out = torch.empty_like(x)
    row = {"shape": [n, 3, h, w], "dtype": name, "normalize_to_0_1": bool(unit)}
    stream = torch.cuda.current_stream(dev).cuda_stream
    for tag, lib in libs.items():
        if lib is None:
            continue
        ws = torch.empty(lib.sx_macenko_workspace_bytes(n, h, w), dtype=torch.uint8, device=dev)
        call = lambda: lib.sx_macenko_transform(x.data_ptr(), out.data_ptr(), code, n, h, w, sm.data_ptr(), tmc.data_ptr(), unit, ws.data_ptr(), ws.numel(), stream)
        assert call() == 0
        row[f"c_abi_{tag}_us"], row[f"c_abi_{tag}_spread"] = timed(call)
    be = MacenkoHIP(dev)
    row["python_backend_head_us"], row["python_backend_head_spread"] = timed(lambda: be.transform(x, sm, tmc, normalize_to_0_1=bool(unit)))
    print(json.dumps(row), flush=True)
    rows.append(row)
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        for r in rows: f.write(json.dumps(r) + "\n")
