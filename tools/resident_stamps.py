"""Phase stamps of the resident Macenko kernel (debug build: tools/build_debug.sh, STAINX_HIP_LIB=.../libstainx_dbg.so).
us from the start of the tile's first workgroup: 1 phase L, 2 plane, 3 angle sweep 0, 4 its exchange + pick, 5 collect sweep, 6 exchange + select,
7-10 the same for the concentrations, 11 before phase R, 12 after it."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from stainx_amd import _native, synth  # noqa: E402
from stainx_amd.backends.torch_hip_backend import MacenkoHIP  # noqa: E402

dev = torch.device("cuda:0")
be = MacenkoHIP(dev)
SM = torch.tensor(synth.HE_REF, dtype=torch.float32)
TMC = torch.tensor([1.9705, 1.0308], dtype=torch.float32)
for n, hw, dt in ((1, 512, torch.float32), (64, 512, torch.float32), (64, 512, torch.uint8), (256, 224, torch.bfloat16)):
    x = synth.as_dtype(synth.he_batch(n, hw, hw, seed0=5), dt).to(dev)
    for _ in range(3):
        be.transform(x, SM, TMC, _extra_flags=_native.MACENKO_RESIDENT)
    p = be.tile_params(n)
    st = p["stamps_us"]
    med = st.median(0).values
    print(f"{n} x {hw} x {hw} {dt}: median over tiles " + " ".join(f"{i}:{float(med[i]):.1f}" for i in range(13)), flush=True)
    print("   slowest tile           " + " ".join(f"{i}:{float(st[:, i].max()):.1f}" for i in range(13)), flush=True)
