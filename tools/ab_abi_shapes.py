"""The Macenko transform through the C ABI alone (ctypes: any build of the library, old ones included) on a few shapes, for A/B on one box:
    python tools/ab_abi_shapes.py libstainx_s0.so libstainx_hip.so libstainx_s0.so libstainx_hip.so"""
import sys, torch, json, ctypes
root = __import__("pathlib").Path(__file__).resolve().parents[1]
sys.path.insert(0, str(root))
from stainx_amd import synth, _native
dev = torch.device("cuda:0")
vp, i64 = ctypes.c_void_p, ctypes.c_int64


def load(name):
    lib = ctypes.CDLL(str(root / "stainx_amd" / "_lib" / name))
    lib.sx_macenko_workspace_bytes.restype = ctypes.c_size_t
    lib.sx_macenko_workspace_bytes.argtypes = [i64, i64, i64]
    lib.sx_macenko_transform.restype = ctypes.c_int
    lib.sx_macenko_transform.argtypes = [vp, vp, ctypes.c_int, i64, i64, i64, vp, vp, ctypes.c_uint, vp, ctypes.c_size_t, vp]
    return lib


sm = torch.tensor(synth.HE_REF).to(dev); tmc = torch.tensor([1.9705, 1.0308]).to(dev)
def survey(lib):
  rows = {}
  for name, dt, shape in (("f32 256x224x224", torch.float32, (256, 224, 224)), ("bf16 256x224x224", torch.bfloat16, (256, 224, 224)), ("u8 256x224x224", torch.uint8, (256, 224, 224)), ("f32 64x512x512", torch.float32, (64, 512, 512)), ("u8 64x512x512", torch.uint8, (64, 512, 512)), ("f32 114x384x384", torch.float32, (114, 384, 384)), ("u8 64x512x512 four-pass", torch.uint8, (64, 512, 512)), ("bf16 64x512x512 four-pass", torch.bfloat16, (64, 512, 512))):
      n, h, w = shape
      x = synth.as_dtype(synth.he_batch(n, h, w), dt).to(dev)
      out = torch.empty_like(x)
      nb = int(lib.sx_macenko_workspace_bytes(n, h, w))
      ws = torch.empty(nb, dtype=torch.uint8, device=dev)
      code = _native.DTYPE_CODES[dt]
      flags = _native.MACENKO_CLASSIC if name.endswith("four-pass") else 0
      def call():
          rc = lib.sx_macenko_transform(x.data_ptr(), out.data_ptr(), code, n, h, w, sm.data_ptr(), tmc.data_ptr(), flags, ws.data_ptr(), ws.numel(), _native.stream_ptr(dev))
          assert rc == 0
      for _ in range(20): call()
      torch.cuda.synchronize()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(200): call()
      e1.record(); torch.cuda.synchronize()
      rows[name] = round(e0.elapsed_time(e1) / 200 * 1e3, 1)
  return rows


for name in sys.argv[1:] or ['libstainx_hip.so']:
    print(name, json.dumps(survey(load(name))), flush=True)
