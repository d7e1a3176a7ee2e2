"""Timeline of ONE fused launch from its per-unit stamps (debug build: tools/build_debug.sh, STAINX_HIP_LIB=.../libstainx_dbg.so):
when the pass-A items of each tile quarter finish, how long the stage jobs wait and work, when the reconstruct items start and end.
    STAINX_HIP_LIB=stainx_amd/_lib/libstainx_dbg.so python tools/fused_timeline.py [tiles H W]"""
import ctypes, json, sys, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from stainx_amd import synth, _native
from stainx_amd.backends.torch_hip_backend import MacenkoHIP

dev = torch.device("cuda:0")
n, h, w = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 512, 512)
x = synth.as_dtype(synth.he_batch(n, h, w), torch.float32).to(dev)
be = MacenkoHIP(dev)
sm = torch.tensor(synth.HE_REF).to(dev); tmc = torch.tensor([1.9705, 1.0308]).to(dev)
lib = ctypes.CDLL(str(_native.os.environ["STAINX_HIP_LIB"]))
lib.sx_debug_unit_stamp_offset.restype = ctypes.c_size_t
lib.sx_debug_unit_stamp_offset.argtypes = [ctypes.c_int64] * 3
off = lib.sx_debug_unit_stamp_offset(n, h, w)
for _ in range(20): be.transform(x, sm, tmc, _extra_flags=_native.MACENKO_TWO_PASS | _native.MACENKO_FUSE)
torch.cuda.synchronize()
bpt = (h * w + 16383) // 16384
units = 2 * n * bpt + 2 * n
st = be.last_workspace[off:off + units * 64].view(torch.int64).reshape(units, 8).cpu()
prior = be.tile_params(n)["stamps_us"]
info = st[:, 3]
xcc, queue, ticket = (info >> 60) & 7, (info >> 56) & 15, (info >> 32) & 0xFFFFFF
tiles_q = (n - queue + 7) // 8
is_a, is_s = ticket < tiles_q * bpt, (ticket >= tiles_q * bpt) & (ticket < tiles_q * (bpt + 2))
is_r = ~is_a & ~is_s
quarter = (ticket * 4) // (tiles_q * bpt)
t0 = int(st[:, 0].min())
us = lambda v: (v.double() - t0) * 0.01
T = us(st)
def mmm(v): return [round(float(x), 1) for x in (v.min(), v.median(), v.max())] if v.numel() else []
A, S, R = T[is_a], T[is_s], T[is_r]
out = {"shape": [n, 3, h, w], "prior_us(last stamp, tile median)": round(float(prior[:, 7].median()), 1),
       "passA_start(min,med,max)": mmm(A[:, 0]),
       "passA_end_by_quarter(med,max)": [[round(float(A[quarter[is_a] == i, 2].median()), 1), round(float(A[quarter[is_a] == i, 2].max()), 1)] for i in range(4) if (quarter[is_a] == i).any()],
       "stage_start": mmm(S[:, 0]), "stage_wait": mmm(S[:, 1] - S[:, 0]), "stage_work": mmm(S[:, 2] - S[:, 1]),
       "stage_phases(plane,keys,select,partner+pinv,conc)": [mmm(S[:, 4] - S[:, 1]), mmm(S[:, 5] - S[:, 4]), mmm(S[:, 6] - S[:, 5]), mmm(S[:, 7] - S[:, 6]), mmm(S[:, 2] - S[:, 7])],
       "stage_end": mmm(S[:, 2]),
       "recon_start": mmm(R[:, 0]), "recon_wait": mmm(R[:, 1] - R[:, 0]), "recon_work": mmm(R[:, 2] - R[:, 1]),
       "recon_first_work_start": round(float(R[:, 1].min()), 1), "recon_end(max)": round(float(R[:, 2].max()), 1),
       "units_per_xcc": torch.bincount(xcc, minlength=8).tolist(), "own_queue_fraction": round(float((xcc == queue).double().mean()), 3)}
print(json.dumps(out))
