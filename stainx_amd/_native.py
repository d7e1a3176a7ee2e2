"""ctypes binding of ``libstainx_hip.so`` (C ABI declared in ``include/stainx_hip.h``).

The library is the product: nothing here falls back to torch ops or to the CPU.  If the shared
object is missing or does not match the header's ABI version, ``HIP_AVAILABLE`` is False and
:func:`require` raises ``ImportError`` -- the analogue of ``stainx_cuda_torch.FUNCTIONS_AVAILABLE``
(reference src/stainx_cuda_torch/__init__.py:31-49).
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

import torch

ABI_VERSION = 1
LIB_PATH = Path(__file__).resolve().parent / "_lib" / "libstainx_hip.so"
DIAG_LIB_PATH = Path(__file__).resolve().parent / "_lib" / "libstainx_diag.so"      # -DSX_DIAG build: diagnostic flags, design-study forms, stage stamps (tests / tools)

SX_OK, SX_ERR_BAD_ARG, SX_ERR_DTYPE, SX_ERR_WORKSPACE, SX_ERR_LAUNCH = range(5)
MACENKO_NORMALIZE_0_1 = 1
MACENKO_CHANNELS_LAST = 2
MACENKO_SAMPLED = 4
MACENKO_CLASSIC = 16
MACENKO_OUT_BF16 = 32
MACENKO_OUT_F16 = 64
# diagnostic build only (libstainx_diag.so; the product library refuses these bits)
MACENKO_NO_TIE_SHORTCUT = 8
MACENKO_SPEC_FAIL = 128
MACENKO_TWO_PASS = 256
MACENKO_FUSE = 512
MACENKO_RESIDENT = 1024
MACENKO_NO_CODES = 2048      # diagnostic build: the four passes over float32 tiles without the 8-bit codes
MACENKO_DIAG_BITS = MACENKO_NO_TIE_SHORTCUT | MACENKO_SPEC_FAIL | MACENKO_TWO_PASS | MACENKO_FUSE | MACENKO_RESIDENT | MACENKO_NO_CODES
MACENKO_PARAM_FLOATS = 48
PFIT_STATS_RECORD_BYTES = 49240
PFIT_SUMS = 1033
PFIT_COMPACT = 32768

DTYPE_CODES = {torch.uint8: 0, torch.float16: 1, torch.bfloat16: 2, torch.float32: 3, torch.float64: 4}

_c = ctypes
_vp, _i64, _int, _uint, _sz = _c.c_void_p, _c.c_int64, _c.c_int, _c.c_uint, _c.c_size_t

# name -> (restype, argtypes); every symbol include/stainx_hip.h declares
SIGNATURES = {
    "sx_version": (_int, []),
    "sx_last_error_string": (_c.c_char_p, []),
    "sx_macenko_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "sx_macenko_workspace_bytes_for": (_sz, [_int, _i64, _i64, _i64, _uint]),
    "sx_macenko_form": (_int, [_int, _i64, _i64, _i64, _uint]),
    "sx_macenko_transform": (_int, [_vp, _vp, _int, _i64, _i64, _i64, _vp, _vp, _uint, _vp, _sz, _vp]),
    "sx_macenko_fit": (_int, [_vp, _int, _i64, _i64, _i64, _vp, _vp, _vp, _sz, _vp]),
    "sx_macenko_tile_params": (_int, [_vp, _i64, _vp, _vp]),
    "sx_macenko_telemetry_offset": (_sz, []),
    "sx_macenko_takes_two_pass": (_int, [_int, _i64, _i64, _i64, _uint]),
    "sx_macenko_dfit_state_bytes": (_sz, []),
    "sx_macenko_dfit_moments": (_int, [_vp, _int, _i64, _i64, _i64, _vp, _vp, _sz, _vp]),
    "sx_macenko_dfit_begin": (_int, [_vp, _vp, _vp]),
    "sx_macenko_dfit_histogram": (_int, [_vp, _int, _i64, _i64, _i64, _vp, _int, _vp, _vp]),
    "sx_macenko_dfit_advance": (_int, [_vp, _int, _vp, _vp]),
    "sx_macenko_dfit_result": (_int, [_vp, _vp, _vp, _vp]),
    "sx_macenko_pfit_sample_count": (_int, [_i64, _i64, _i64]),
    "sx_macenko_pfit_stats": (_int, [_vp, _int, _i64, _i64, _i64, _vp, _vp, _vp, _sz, _vp]),
    "sx_macenko_pfit_plane": (_int, [_vp, _c.c_longlong, _vp, _int, _i64, _i64, _i64, _vp, _sz, _vp]),
    "sx_macenko_pfit_pass": (_int, [_vp, _int, _i64, _i64, _i64, _int, _c.c_longlong, _int, _vp, _vp, _sz, _vp]),
    "sx_macenko_pfit_gather": (_int, [_vp, _int, _c.c_longlong, _int, _i64, _i64, _i64, _int, _vp, _vp, _vp, _sz, _vp]),
    "sx_macenko_pfit_finish": (_int, [_vp, _vp, _int, _int, _int, _c.c_longlong, _int, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sx_macenko_pfit_stats_packed": (_int, [_vp, _int, _i64, _i64, _i64, _vp, _vp, _sz, _vp]),
    "sx_macenko_pfit_plane_packed": (_int, [_vp, _int, _vp, _vp, _vp, _c.c_longlong, _int, _i64, _i64, _i64, _vp, _sz, _vp]),
    "sx_macenko_pfit_gather_packed": (_int, [_vp, _int, _c.c_longlong, _int, _i64, _i64, _i64, _int, _vp, _vp, _vp, _sz, _vp]),
    "sx_macenko_pfit_finish_packed": (_int, [_vp, _int, _int, _int, _c.c_longlong, _int, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sx_reinhard_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "sx_reinhard_workspace_bytes_for": (_sz, [_int, _i64, _i64, _i64]),
    "sx_reinhard_sums": (_int, [_vp, _int, _i64, _i64, _i64, _vp, _vp, _sz, _vp]),
    "sx_reinhard_apply": (_int, [_vp, _vp, _int, _i64, _i64, _i64, _vp, _c.c_double, _vp, _vp, _vp, _sz, _vp]),
    "sx_hm_counts": (_int, [_vp, _int, _i64, _i64, _i64, _int, _vp, _vp, _sz, _vp]),
    "sx_hm_apply": (_int, [_vp, _vp, _int, _i64, _i64, _i64, _int, _vp, _c.c_double, _vp, _vp, _sz, _vp]),
    "sx_reinhard_fit": (_int, [_vp, _int, _i64, _i64, _i64, _vp, _vp, _vp, _sz, _vp]),
    "sx_reinhard_transform": (_int, [_vp, _vp, _int, _i64, _i64, _i64, _vp, _vp, _vp, _sz, _vp]),
    "sx_reinhard_workspace_init": (_int, [_vp, _sz, _vp]),
    "sx_reinhard_workspace_status_offset": (_sz, []),
    "sx_reinhard_transform_ready": (_int, [_vp, _vp, _int, _i64, _i64, _i64, _vp, _vp, _vp, _sz, _vp]),
    "sx_hm_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "sx_hm_fit": (_int, [_vp, _int, _i64, _i64, _i64, _int, _vp, _vp, _sz, _vp]),
    "sx_hm_transform": (_int, [_vp, _vp, _int, _i64, _i64, _i64, _int, _vp, _vp, _sz, _vp]),
    "sx_hm_workspace_init": (_int, [_vp, _sz, _vp]),
    "sx_hm_workspace_status_offset": (_sz, []),
    "sx_hm_fit_ready": (_int, [_vp, _int, _i64, _i64, _i64, _int, _vp, _vp, _sz, _vp]),
    "sx_hm_transform_ready": (_int, [_vp, _vp, _int, _i64, _i64, _i64, _int, _vp, _vp, _sz, _vp]),
    "sx_hm_counts_ready": (_int, [_vp, _int, _i64, _i64, _i64, _int, _vp, _vp, _sz, _vp]),
}

_lib = None
_load_error: str | None = None
_diag_lib = None


# entry points only the diagnostic build exports (include/stainx_hip.h, #ifdef SX_DIAG)
DIAG_SIGNATURES = {
    "sx_hm_workspace_parity_offset": (_sz, []),
    "sx_debug_hm_stamp_offset": (_sz, []),
}


def _open(path: Path):
    lib = ctypes.CDLL(str(path))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    for name, (res, args) in DIAG_SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype, fn.argtypes = res, args
    got = lib.sx_version()
    if got != ABI_VERSION:
        raise OSError(f"ABI version {got} != expected {ABI_VERSION}")
    return lib


def _load():
    global _lib, _load_error
    if _lib is not None or _load_error is not None:
        return _lib
    # STAINX_DIAG=1: the diagnostic build stands in for the product (tools that A/B the forms); STAINX_HIP_LIB names any other build
    default = DIAG_LIB_PATH if os.environ.get("STAINX_DIAG", "").strip().lower() in ("1", "true", "yes", "on") else LIB_PATH
    path = Path(os.environ.get("STAINX_HIP_LIB", default))
    try:
        _lib = _open(path)
    except (OSError, AttributeError) as exc:
        _load_error = f"{path}: {exc}"
    return _lib


def require_diag():
    """The diagnostic build (libstainx_diag.so): the product's code plus the diagnostic flags and the design-study forms.  Tests and tools."""
    global _diag_lib
    if _diag_lib is None:
        try:
            _diag_lib = _open(DIAG_LIB_PATH)
        except (OSError, AttributeError) as exc:
            raise ImportError(f"libstainx_diag.so is not built or not loadable ({DIAG_LIB_PATH}: {exc}); `python -c 'import __graft_entry__ as g; g.build()'` builds it next to the product") from None
    return _diag_lib


def library_available() -> bool:
    """The shared object loads and exports every declared symbol (no GPU needed)."""
    return _load() is not None


def hip_available() -> bool:
    """Library loaded AND a ROCm device is visible to torch."""
    return library_available() and torch.version.hip is not None and torch.cuda.is_available()


def check_arch(device: torch.device) -> None:
    """The code objects in the library are gfx950 (MI355X) only: say so instead of failing at the first launch."""
    if not torch.cuda.is_available():
        return
    arch = getattr(torch.cuda.get_device_properties(device), "gcnArchName", "") or ""
    if arch and not arch.split(":")[0].startswith("gfx950"):
        raise RuntimeError(f"libstainx_hip.so is built for gfx950 (MI355X); device {device} reports {arch}")


def require():
    lib = _load()
    if lib is None:
        raise ImportError(f"libstainx_hip.so is not built or not loadable ({_load_error}). Build it with "
                          "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
                          "There is no CPU fallback in this package.")
    return lib


def last_error(lib=None) -> str:
    return (lib or require()).sx_last_error_string().decode("utf-8", "replace")


def check(code: int, what: str, lib=None) -> None:
    if code == SX_OK:
        return
    msg = f"{what} failed (status {code}): {last_error(lib)}"
    raise RuntimeError(msg)      # the reference's TORCH_CHECK failures surface as RuntimeError too


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)      # the handle itself, without a Stream object (a step of the pooled fit asks ~20 times)


def device_index(device: torch.device) -> int:
    return device.index if device.index is not None else torch.cuda.current_device()


def stream_ptr(device: torch.device) -> int:
    """The HIP stream handle of the device's CURRENT torch stream (what every entry point takes as its last argument)."""
    if _raw_stream is not None:
        return int(_raw_stream(device_index(device)))
    return torch.cuda.current_stream(device).cuda_stream


class on_device:
    """`with on_device(dev):` -- torch.cuda.device(dev) when dev is not already the current device, nothing otherwise (the guard costs
    several microseconds a time; the host side of a pooled fit step takes it a dozen times)."""

    __slots__ = ("_guard",)

    def __init__(self, device: torch.device):
        self._guard = None if device_index(device) == torch.cuda.current_device() else torch.cuda.device(device)

    def __enter__(self):
        if self._guard is not None:
            self._guard.__enter__()
        return self

    def __exit__(self, *exc):
        if self._guard is not None:
            return self._guard.__exit__(*exc)
        return False


class Scratch:
    """Device workspaces owned by torch's caching allocator (the library never allocates): ONE PER STREAM, so that calls on
    different streams (or threads on different streams) never share selection state, and a buffer that has been handed out is
    never freed while this object lives -- a captured HIP graph, or a call still in flight, may hold its address (a superseded
    buffer is retired, not released).  Inside a stream capture the allocation comes from the graph's private pool."""

    def __init__(self, zeroed: bool = False):
        # zeroed: buffers are zero-filled when they are made (stream-ordered, once) -- the READY state of the histogram-matching
        # workspace (include/stainx_hip.h: sx_hm_*_ready); nobody but the library writes to them afterwards
        self._zeroed = zeroed
        self._bufs: dict[tuple[int, int], torch.Tensor] = {}
        self._retired: list[torch.Tensor] = []

    def get(self, nbytes: int, device: torch.device) -> torch.Tensor:
        key = (device_index(device), stream_ptr(device))
        buf = self._bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            if buf is not None:
                self._retired.append(buf)
            make = torch.zeros if self._zeroed else torch.empty
            buf = make(max(int(nbytes), 256), dtype=torch.uint8, device=device)
            self._bufs[key] = buf
        return buf

    def drop(self, device: torch.device) -> None:
        """Forget the current stream's buffer (after a failed call its state is unknown): the next get() makes a new one."""
        key = (device_index(device), stream_ptr(device))
        buf = self._bufs.pop(key, None)
        if buf is not None:
            self._retired.append(buf)
