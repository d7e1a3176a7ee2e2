"""``backend="torch_hip"``: torch.Tensor in, torch.Tensor out, arithmetic in libstainx_hip.so.

Same class contract as the reference's plugin surface (src/stainx/backends/torch_cuda_backend.py):
``cls(device, **kwargs)`` then ``impl.transform(images, *reference_params)``; exceptions keep the
reference's types (``ImportError`` when the native library is missing, ``ValueError`` for a non-GPU
device or bad shapes, ``RuntimeError`` from the native layer).  Unlike the reference, fit-time
statistics are computed on the device by the same library (reference: always torch on CPU,
torch_backend.py:463-466) -- there is no torch-op or CPU path in this package.
"""
from __future__ import annotations

import ctypes as _c
import os
import threading

import torch

from stainx_amd import _native

HIP_AVAILABLE = _native.library_available()


def _dtype_code(t: torch.Tensor) -> int:
    try:
        return _native.DTYPE_CODES[t.dtype]
    except KeyError:
        raise TypeError(f"unsupported image dtype {t.dtype}; supported: {sorted(str(d) for d in _native.DTYPE_CODES)}") from None


class TorchHIPBackendBase:
    """Common device checks (mirrors TorchCUDABackendBase, torch_cuda_backend.py:17-33)."""

    def __init__(self, device: str | torch.device | None = None):
        if not _native.library_available():
            _native.require()                      # raises ImportError with the build hint
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("no ROCm device is available on this system")
            self.device = torch.device("cuda", torch.cuda.current_device())
        else:
            self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError(f"HIP backend requires a CUDA (ROCm) device, got {self.device.type}")
        if self.device.index is None and torch.cuda.is_available():
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._lib = _native.require()
        _native.check_arch(self.device)
        self._scratch = _native.Scratch()
        self._pfit_bytes: dict[tuple[int, int, int], int] = {}

    def _f32(self, t: torch.Tensor) -> torch.Tensor:
        return t.to(device=self.device, dtype=torch.float32).contiguous()


class MacenkoHIP(TorchHIPBackendBase):
    """Macenko transform / fit on the GPU (numerics of MacenkoTorch, torch_backend.py:358-560)."""

    def __init__(self, device: str | torch.device | None = None, precision: str = "stable", diag: bool = False):
        super().__init__(device)
        if diag:      # tests / tools: the diagnostic build (its flags force the rare paths; the design-study forms live there)
            self._lib = _native.require_diag()
        if precision not in ("stable", "fast", "sampled"):
            raise ValueError(f"precision must be 'stable' or 'fast' (or the extension 'sampled'), got {precision!r}")
        # "stable" and "fast" run the SAME exact kernels (fp64 covariance, exact nearest-rank percentiles).  The reference's "fast"
        # (torch_cuda_backend.py:114-118; macenko.cu:116-191) keeps the exact percentiles and moves the big tensors to fp16 for a
        # 1.2-1.3x gain at MAE ~0.05 grey levels; here the exact path is the fast one, and its result lies inside that mode's
        # tolerance (tests/test_precision_modes_gpu.py holds it to the restated fp16 path).
        # "sampled" (an extension, opt-in, NOT a parity path): the percentiles of a 4096-pixel sample of each tile stand in for the
        # exact ones -- moments pass + one per-tile stage + reconstruct; mean error ~0.5, worst ~5 grey levels.  The fit is always exact.
        self._precision = precision
        self.last_workspace: torch.Tensor | None = None
        # Feedback for the choice between the two forms of the transform (see _route / _watch): the two-pass form speculates per
        # tile and pays a whole-tile exact select (0.1-0.5 ms) for a tile it cannot speculate on -- no tissue, no stable stain
        # plane.  The library counts those; the count is read back asynchronously and a batch stream that produces them is
        # switched to the four-pass form (same bits, no cliff), with an occasional probe.
        self._tele_offset = int(self._lib.sx_macenko_telemetry_offset())
        self._tele_host: torch.Tensor | None = None
        self._tele_event: torch.cuda.Event | None = None
        self._tele_stream: torch.cuda.Stream | None = None
        self._classic_left = 0
        self._classic_span = 32
        self._tele_age = 0
        self._tele_ptr = 0
        self._tele_seen: dict[int, tuple[int, int]] = {}      # workspace pointer -> (slow selections, waits that ran out) read there last
        self._tele_lock = threading.Lock()                     # (one object may be driven from several threads, each on its own stream)
        # A/B switch for benchmarks (it used to live inside the library): the four-pass form everywhere
        self._env_flags = _native.MACENKO_CLASSIC if os.environ.get("STAINX_MACENKO_CLASSIC", "").strip().lower() in ("1", "true", "yes", "on") else 0

    @staticmethod
    def _check_images(images: torch.Tensor, what: str) -> None:
        if images.dim() != 4:
            raise ValueError(f"Macenko expects NCHW images, got shape {tuple(images.shape)}")
        if images.shape[1] != 3:
            raise ValueError(f"Macenko {what} expects 3 channels in dim 1 (NCHW), got C={images.shape[1]} with shape {tuple(images.shape)}")

    def transform(self, images: torch.Tensor, stain_matrix: torch.Tensor, target_max_conc: torch.Tensor, *, normalize_to_0_1: bool = False,
                  channels_last: bool = False, out_dtype: torch.dtype | None = None, _extra_flags: int = 0) -> torch.Tensor:
        """``channels_last=True`` (an extension; the reference takes NCHW only): ``images`` is (N,H,W,3) as decoders and PIL
        hand tiles over, and so is the result -- the permute + copy a caller would otherwise do first is fused away.
        ``out_dtype=torch.bfloat16 / torch.float16`` (an extension, uint8 input only): the result of the call without it, cast
        with ``.to(out_dtype)``, written directly -- a decoder's uint8 tile becomes a model's half-precision input in one call."""
        images = images.to(self.device)
        if out_dtype is not None and out_dtype != images.dtype:
            if images.dtype != torch.uint8 or out_dtype not in (torch.bfloat16, torch.float16):
                raise ValueError(f"out_dtype is supported for uint8 input and bfloat16 / float16 output, got {images.dtype} -> {out_dtype}")
            _extra_flags = int(_extra_flags) | (_native.MACENKO_OUT_BF16 if out_dtype == torch.bfloat16 else _native.MACENKO_OUT_F16)
        else:
            out_dtype = None
        if tuple(stain_matrix.shape) != (3, 2):
            raise ValueError(f"stain_matrix must have shape (3, 2), got {stain_matrix.shape}")
        if channels_last:
            if images.dim() != 4 or images.shape[3] != 3:
                raise ValueError(f"Macenko transform with channels_last expects NHWC images with C=3, got shape {tuple(images.shape)}")
        else:
            self._check_images(images, "transform")
        sm = self._f32(stain_matrix)
        tmc = self._f32(target_max_conc).flatten()
        if tmc.numel() != 2:
            raise ValueError(f"target_max_conc must have 2 elements, got {tmc.numel()}")
        images = images.contiguous()
        n, h, w = (images.shape[0], images.shape[1], images.shape[2]) if channels_last else (images.shape[0], images.shape[2], images.shape[3])
        code = _dtype_code(images)
        if out_dtype is None:
            out_dtype = torch.float32 if (normalize_to_0_1 and images.dtype == torch.uint8) else images.dtype
        out = torch.empty(tuple(images.shape), dtype=out_dtype, device=self.device)
        if n == 0 or h * w == 0:
            return out
        with _native.on_device(self.device):
            flags = ((_native.MACENKO_NORMALIZE_0_1 if normalize_to_0_1 else 0) | (_native.MACENKO_CHANNELS_LAST if channels_last else 0)
                     | (_native.MACENKO_SAMPLED if self._precision == "sampled" else 0) | int(_extra_flags) | self._env_flags)
            # (only calls the library would run in its two-pass form take part in the feedback: for the others -- small batches,
            # narrow pixels -- the event, the side-stream copy and the stream bookkeeping are 7 us of host time per call for nothing)
            routed = (not (flags & (_native.MACENKO_CLASSIC | _native.MACENKO_TWO_PASS | _native.MACENKO_SAMPLED))
                      and self._lib.sx_macenko_takes_two_pass(code, n, h, w, flags) == 1)
            # the workspace of the form the library WOULD take: a routed call may be sent to the four-pass form below, whose
            # workspace is a prefix of it (one buffer per stream serves both)
            nbytes = self._lib.sx_macenko_workspace_bytes_for(code, n, h, w, flags)
            ws = self._scratch.get(nbytes, self.device)
            if routed:
                with self._tele_lock:
                    flags |= self._route()
            rc = self._lib.sx_macenko_transform(images.data_ptr(), out.data_ptr(), code, n, h, w, sm.data_ptr(), tmc.data_ptr(),
                                                flags, ws.data_ptr(), ws.numel(), _native.stream_ptr(self.device))
            _native.check(rc, "sx_macenko_transform", self._lib)
            if routed and not (flags & _native.MACENKO_CLASSIC):
                with self._tele_lock:
                    self._watch(ws)
        self.last_workspace = ws
        return out

    def _route(self) -> int:
        """Flag for this call: the four-pass form while a recent call reported tiles the two-pass form could not speculate on.
        Never synchronises the host with the device except to wait for an answer that is five calls old (see below)."""
        if torch.cuda.is_current_stream_capturing():
            return _native.MACENKO_CLASSIC      # a captured call is replayed on data nobody has seen: the form without a cliff (and no event may be queried here)
        if self._tele_event is not None:
            self._tele_age += 1
            if self._tele_age > 4 and not self._tele_event.query():
                # The host runs ahead of the device (a loop that never synchronises queues its calls long before they run): an
                # answer that is five calls old is waited for, so that a batch the two-pass form cannot speculate on costs a few
                # slow calls, not every call until the host happens to synchronise.  Five calls are still queued behind the one
                # waited for: the device does not run dry.
                self._tele_event.synchronize()
        if self._tele_event is not None and self._tele_event.query():
            # running counts (the library only adds to them): what is new since the last look at THIS workspace.  The first look
            # at a workspace only sets its base (fresh memory holds anything) -- taken from this same asynchronous read-back, so
            # a call on a new stream or a grown scratch costs no host synchronisation.
            now = (int(self._tele_host[0]) & 0xFFFFFFFF, int(self._tele_host[1]) & 0xFFFFFFFF)
            seen = self._tele_seen.get(self._tele_ptr)
            self._tele_seen[self._tele_ptr] = now
            self._tele_event = None
            slow = ((now[0] - seen[0]) & 0xFFFFFFFF) if seen is not None else 0
            if seen is not None and ((now[1] - seen[1]) & 0xFFFFFFFF) != 0:
                raise RuntimeError("sx_macenko_transform: a bounded wait inside the fused launch ran out on an earlier call (the device did not run its "
                                   "workgroups to completion); that call's output is incomplete")
            if slow > 0:
                self._classic_left = self._classic_span
                self._classic_span = min(self._classic_span * 2, 4096)      # probe again, less and less often
            else:
                self._classic_span = 32
                self._classic_left = 0      # (a probe that came back clean ends the provisional four-pass calls behind it)
        if self._classic_left > 0:
            self._classic_left -= 1
            return _native.MACENKO_CLASSIC
        if self._classic_span > 32 and self._tele_event is None:
            # A PROBE: the data was hard not long ago, and this call tries the two-pass form again.  Only this one: the calls behind it
            # stay with the four passes until its answer is in (a probe used to be every call up to the answer -- up to five calls at
            # 0.7 ms on real tissue, 10-20 us per step averaged over a loop of 200).
            self._classic_left = 8
        return 0

    def _watch(self, ws: torch.Tensor) -> None:
        """Read the library's counts of slow selections (and of waits that ran out) back without making anything wait: a side stream
        copies eight bytes once the call's kernels are done.  (Not inside a stream capture; advisory only.)"""
        if self._tele_event is not None or torch.cuda.is_current_stream_capturing():
            return
        if self._tele_host is None:
            self._tele_host = torch.zeros(2, dtype=torch.int32).pin_memory()
            self._tele_stream = torch.cuda.Stream(self.device)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(self.device))
        self._tele_stream.wait_event(done)
        with torch.cuda.stream(self._tele_stream):
            self._tele_host.copy_(ws[self._tele_offset:self._tele_offset + 8].view(torch.int32), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._tele_stream)
        ws.record_stream(self._tele_stream)
        self._tele_event = ev
        self._tele_ptr = ws.data_ptr()
        self._tele_age = 0

    def compute_reference_stain_matrix(self, images: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        """Pooled stain estimate ``(HE (3,2), maxC (2,))`` (compute_reference_stain_matrix_torch, :463-519)."""
        images = images.to(self.device)
        if images.dim() != 4 or images.shape[1] != 3:
            raise ValueError(f"Macenko fit expects NCHW with C=3, got shape {tuple(images.shape)}")
        images = images.contiguous()
        n, _, h, w = images.shape
        he = torch.empty((3, 2), dtype=torch.float32, device=self.device)
        max_c = torch.empty((2,), dtype=torch.float32, device=self.device)
        with _native.on_device(self.device):
            nbytes = self._lib.sx_macenko_workspace_bytes_for(_dtype_code(images), n, h, w, _native.MACENKO_CLASSIC)
            ws = self._scratch.get(nbytes, self.device)
            rc = self._lib.sx_macenko_fit(images.data_ptr(), _dtype_code(images), n, h, w, he.data_ptr(), max_c.data_ptr(),
                                          ws.data_ptr(), ws.numel(), _native.stream_ptr(self.device))
        _native.check(rc, "sx_macenko_fit")
        self.last_workspace = ws
        return he, max_c

    # name kept so code written against the reference's torch class keeps working
    compute_reference_stain_matrix_torch = compute_reference_stain_matrix

    # ---- staged pooled fit for a batch sharded across ranks (see stainx_amd/distributed.py) ----------
    def dfit_moments(self, images: torch.Tensor) -> torch.Tensor:
        images = images.to(self.device)
        if images.dim() != 4 or images.shape[1] != 3:
            raise ValueError(f"Macenko fit expects NCHW with C=3, got shape {tuple(images.shape)}")
        images = images.contiguous()
        n, _, h, w = images.shape
        mom = torch.empty(20, dtype=torch.float64, device=self.device)
        with _native.on_device(self.device):
            ws = self._scratch.get(self._lib.sx_macenko_workspace_bytes_for(_dtype_code(images), n, h, w, _native.MACENKO_CLASSIC), self.device)
            rc = self._lib.sx_macenko_dfit_moments(images.data_ptr(), _dtype_code(images), n, h, w, mom.data_ptr(), ws.data_ptr(), ws.numel(),
                                                   _native.stream_ptr(self.device))
        _native.check(rc, "sx_macenko_dfit_moments")
        return mom

    def dfit_begin(self, moments: torch.Tensor) -> torch.Tensor:
        state = torch.zeros(self._lib.sx_macenko_dfit_state_bytes(), dtype=torch.uint8, device=self.device)
        moments = moments.to(self.device, torch.float64).contiguous()
        with _native.on_device(self.device):
            rc = self._lib.sx_macenko_dfit_begin(moments.data_ptr(), state.data_ptr(), _native.stream_ptr(self.device))
        _native.check(rc, "sx_macenko_dfit_begin")
        return state

    def dfit_histogram(self, images: torch.Tensor, state: torch.Tensor, stage: int) -> torch.Tensor:
        images = images.to(self.device).contiguous()
        n, _, h, w = images.shape
        hist = torch.empty((2, 256), dtype=torch.int64, device=self.device)
        with _native.on_device(self.device):
            rc = self._lib.sx_macenko_dfit_histogram(images.data_ptr(), _dtype_code(images), n, h, w, state.data_ptr(), int(stage), hist.data_ptr(),
                                                     _native.stream_ptr(self.device))
        _native.check(rc, "sx_macenko_dfit_histogram")
        return hist

    def dfit_advance(self, state: torch.Tensor, stage: int, hist: torch.Tensor) -> None:
        hist = hist.to(self.device, torch.int64).contiguous()
        with _native.on_device(self.device):
            rc = self._lib.sx_macenko_dfit_advance(state.data_ptr(), int(stage), hist.data_ptr(), _native.stream_ptr(self.device))
        _native.check(rc, "sx_macenko_dfit_advance")

    def dfit_result(self, state: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        he = torch.empty((3, 2), dtype=torch.float32, device=self.device)
        max_c = torch.empty((2,), dtype=torch.float32, device=self.device)
        with _native.on_device(self.device):
            rc = self._lib.sx_macenko_dfit_result(state.data_ptr(), he.data_ptr(), max_c.data_ptr(), _native.stream_ptr(self.device))
        _native.check(rc, "sx_macenko_dfit_result")
        return he, max_c

    # ---- the same pooled fit across ranks on the bracket machinery (three passes; see stainx_amd/distributed.py) ----
    def _pfit_ws(self, n: int, h: int, w: int) -> torch.Tensor:
        need = self._pfit_bytes.get((n, h, w))
        if need is None:
            need = self._pfit_bytes[(n, h, w)] = int(self._lib.sx_macenko_workspace_bytes_for(_native.DTYPE_CODES[torch.float32], n, h, w, _native.MACENKO_CLASSIC))
        return self._scratch.get(need, self.device)

    def pfit_sample_count(self, n: int, h: int, w: int) -> int:
        return int(self._lib.sx_macenko_pfit_sample_count(n, h, w))

    def pfit_stats(self, images: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        """Local raw moments (10 float64) and the local sample's optical density (3, 4096) float32."""
        images = images.to(self.device)
        if images.dim() != 4 or images.shape[1] != 3:
            raise ValueError(f"Macenko fit expects NCHW with C=3, got shape {tuple(images.shape)}")
        images = images.contiguous()
        n, _, h, w = images.shape
        mom = torch.empty(10, dtype=torch.float64, device=self.device)
        sample = torch.empty((3, 4096), dtype=torch.float32, device=self.device)
        with _native.on_device(self.device):
            ws = self._pfit_ws(n, h, w)
            rc = self._lib.sx_macenko_pfit_stats(images.data_ptr(), _dtype_code(images), n, h, w, mom.data_ptr(), sample.data_ptr(), ws.data_ptr(), ws.numel(),
                                                 _native.stream_ptr(self.device))
        _native.check(rc, "sx_macenko_pfit_stats")
        self.last_workspace = ws
        return mom, sample

    def pfit_plane(self, moments: torch.Tensor, n_all: int, sample_union: torch.Tensor, sample_count: int, shape: tuple[int, int, int]) -> None:
        n, h, w = shape
        moments = moments.to(self.device, torch.float64).contiguous()
        sample_union = sample_union.to(self.device, torch.float32).contiguous()
        with _native.on_device(self.device):
            ws = self._pfit_ws(n, h, w)
            rc = self._lib.sx_macenko_pfit_plane(moments.data_ptr(), int(n_all), sample_union.data_ptr(), int(sample_count), n, h, w, ws.data_ptr(), ws.numel(),
                                                 _native.stream_ptr(self.device))
        _native.check(rc, "sx_macenko_pfit_plane")

    def pfit_pass(self, images: torch.Tensor, stage: int, n_all: int, sample_count: int) -> torch.Tensor:
        images = images.to(self.device).contiguous()
        n, _, h, w = images.shape
        sums = torch.empty(_native.PFIT_SUMS, dtype=torch.int64, device=self.device)
        with _native.on_device(self.device):
            ws = self._pfit_ws(n, h, w)
            rc = self._lib.sx_macenko_pfit_pass(images.data_ptr(), _dtype_code(images), n, h, w, int(stage), int(n_all), int(sample_count), sums.data_ptr(), ws.data_ptr(),
                                                ws.numel(), _native.stream_ptr(self.device))
        _native.check(rc, "sx_macenko_pfit_pass")
        return sums

    def pfit_gather(self, sums_global: torch.Tensor, stage: int, n_all: int, sample_count: int, shape: tuple[int, int, int], share: int) -> tuple[torch.Tensor, torch.Tensor]:
        n, h, w = shape
        sums_global = sums_global.to(self.device, torch.int64).contiguous()
        compact = torch.empty((2, int(share)), dtype=torch.int32, device=self.device)      # (only the first `counts` entries of a row are written and read: no fill)
        counts = torch.empty(2, dtype=torch.int32, device=self.device)
        with _native.on_device(self.device):
            ws = self._pfit_ws(n, h, w)
            rc = self._lib.sx_macenko_pfit_gather(sums_global.data_ptr(), int(stage), int(n_all), int(sample_count), n, h, w, int(share), compact.data_ptr(), counts.data_ptr(),
                                                  ws.data_ptr(), ws.numel(), _native.stream_ptr(self.device))
        _native.check(rc, "sx_macenko_pfit_gather")
        return compact, counts

    def pfit_finish(self, gathered_compact: torch.Tensor, gathered_counts: torch.Tensor, stage: int, n_all: int, sample_count: int, shape: tuple[int, int, int]):
        """Stage 0: nothing returned.  Stage 1: (HE, maxC, status) -- status is a device int32, non-zero if a bracket missed."""
        n, h, w = shape
        world, share = int(gathered_counts.shape[0]), int(gathered_compact.shape[-1])
        gathered_compact = gathered_compact.to(self.device, torch.int32).contiguous()
        gathered_counts = gathered_counts.to(self.device, torch.int32).contiguous()
        he = torch.empty((3, 2), dtype=torch.float32, device=self.device)
        max_c = torch.empty((2,), dtype=torch.float32, device=self.device)
        status = torch.zeros(1, dtype=torch.int32, device=self.device)
        with _native.on_device(self.device):
            ws = self._pfit_ws(n, h, w)
            rc = self._lib.sx_macenko_pfit_finish(gathered_compact.data_ptr(), gathered_counts.data_ptr(), world, share, int(stage), int(n_all), int(sample_count), n, h, w,
                                                  he.data_ptr(), max_c.data_ptr(), status.data_ptr(), ws.data_ptr(), ws.numel(), _native.stream_ptr(self.device))
        _native.check(rc, "sx_macenko_pfit_finish")
        self.last_workspace = ws
        return (he, max_c, status) if stage == 1 else None

    # ---- the same steps with the records of the exchanges packed and unpacked by the library (include/stainx_hip.h) ----
    def pfit_stats_packed(self, images: torch.Tensor) -> torch.Tensor:
        """This rank's stats record [int64 tiles | 10 fp64 moments | 3 x 4096 fp32 sample] as bytes, ready for the all-gather."""
        images = images.to(self.device)
        if images.dim() != 4 or images.shape[1] != 3:
            raise ValueError(f"Macenko fit expects NCHW with C=3, got shape {tuple(images.shape)}")
        images = images.contiguous()
        n, _, h, w = images.shape
        record = torch.empty(_native.PFIT_STATS_RECORD_BYTES, dtype=torch.uint8, device=self.device)
        with _native.on_device(self.device):
            ws = self._pfit_ws(n, h, w)
            rc = self._lib.sx_macenko_pfit_stats_packed(images.data_ptr(), _dtype_code(images), n, h, w, record.data_ptr(), ws.data_ptr(), ws.numel(), _native.stream_ptr(self.device))
        _native.check(rc, "sx_macenko_pfit_stats_packed")
        self.last_workspace = ws
        return record

    def pfit_empty_record(self) -> torch.Tensor:
        """The record of a rank without tiles (it still takes part in the exchange that tells every rank so)."""
        return torch.zeros(_native.PFIT_STATS_RECORD_BYTES, dtype=torch.uint8, device=self.device)

    def pfit_plane_packed(self, gathered: torch.Tensor, sample_counts: list[int], expected_tiles: torch.Tensor | None, n_all: int, sample_count: int,
                          shape: tuple[int, int, int]) -> torch.Tensor:
        """Every rank's stats record (world, record bytes) -> plane and angle brackets; returns the one-element int32 "some rank's
        tile count is not the expected one" flag (zero without `expected_tiles`)."""
        n, h, w = shape
        world = int(gathered.shape[0])
        gathered = gathered.contiguous()
        counts = (_c.c_int * world)(*[int(v) for v in sample_counts])
        stale = torch.empty(1, dtype=torch.int32, device=self.device)
        with _native.on_device(self.device):
            ws = self._pfit_ws(n, h, w)
            rc = self._lib.sx_macenko_pfit_plane_packed(gathered.data_ptr(), world, _c.cast(counts, _c.c_void_p), expected_tiles.data_ptr() if expected_tiles is not None else None,
                                                        stale.data_ptr(), int(n_all), int(sample_count), n, h, w, ws.data_ptr(), ws.numel(), _native.stream_ptr(self.device))
        _native.check(rc, "sx_macenko_pfit_plane_packed")
        return stale

    def pfit_gather_packed(self, sums_global: torch.Tensor, stage: int, n_all: int, sample_count: int, shape: tuple[int, int, int], share: int, stale: torch.Tensor | None) -> torch.Tensor:
        n, h, w = shape
        sums_global = sums_global.to(self.device, torch.int64).contiguous()
        row = torch.empty(3 + 2 * int(share), dtype=torch.int32, device=self.device)      # (only the counted entries are written and read: no fill)
        with _native.on_device(self.device):
            ws = self._pfit_ws(n, h, w)
            rc = self._lib.sx_macenko_pfit_gather_packed(sums_global.data_ptr(), int(stage), int(n_all), int(sample_count), n, h, w, int(share),
                                                         stale.data_ptr() if stale is not None else None, row.data_ptr(), ws.data_ptr(), ws.numel(), _native.stream_ptr(self.device))
        _native.check(rc, "sx_macenko_pfit_gather_packed")
        return row

    def pfit_finish_packed(self, gathered_rows: torch.Tensor, stage: int, n_all: int, sample_count: int, shape: tuple[int, int, int], share: int):
        """Stage 0: nothing returned.  Stage 1: (HE, maxC, status) -- bits 0-3 of the device int32: a bracket missed; bit 4: a stale flag."""
        n, h, w = shape
        world = int(gathered_rows.shape[0])
        gathered_rows = gathered_rows.contiguous()
        he = torch.empty((3, 2), dtype=torch.float32, device=self.device)
        max_c = torch.empty((2,), dtype=torch.float32, device=self.device)
        status = torch.empty(1, dtype=torch.int32, device=self.device)
        with _native.on_device(self.device):
            ws = self._pfit_ws(n, h, w)
            rc = self._lib.sx_macenko_pfit_finish_packed(gathered_rows.data_ptr(), world, int(share), int(stage), int(n_all), int(sample_count), n, h, w, he.data_ptr(), max_c.data_ptr(),
                                                         status.data_ptr(), ws.data_ptr(), ws.numel(), _native.stream_ptr(self.device))
        _native.check(rc, "sx_macenko_pfit_finish_packed")
        self.last_workspace = ws
        return (he, max_c, status) if stage == 1 else None

    def tile_params(self, n_groups: int) -> dict[str, torch.Tensor]:
        """Intermediates of the last transform (per tile) or fit (one pooled group); used by tests."""
        if self.last_workspace is None:
            raise RuntimeError("no transform / fit has run on this backend yet")
        raw = torch.empty((n_groups, _native.MACENKO_PARAM_FLOATS), dtype=torch.float32, device=self.device)
        with _native.on_device(self.device):
            rc = self._lib.sx_macenko_tile_params(self.last_workspace.data_ptr(), n_groups, raw.data_ptr(), _native.stream_ptr(self.device))
        _native.check(rc, "sx_macenko_tile_params")
        raw = raw.cpu()
        return {"n_kept": raw[:, 0].long(), "use_all": raw[:, 1].long(), "vecs": raw[:, 2:8].reshape(-1, 3, 2), "phi_lo": raw[:, 8],
                "phi_hi": raw[:, 9], "he": raw[:, 10:16].reshape(-1, 3, 2), "max_c": raw[:, 16:18], "fell_back": raw[:, 18].long(),
                "n_candidates": raw[:, 19:23].long(), "cov": raw[:, 23:32].reshape(-1, 3, 3), "stamps_us": raw[:, 32:48]}


class ReinhardHIP(TorchHIPBackendBase):
    """Reinhard LAB statistics matching on the GPU (numerics of ReinhardTorch, torch_backend.py:304-355)."""

    @staticmethod
    def _check(images: torch.Tensor) -> None:
        if images.dim() != 4 or images.shape[1] != 3:
            raise ValueError(f"Reinhard expects NCHW images with C=3, got shape {tuple(images.shape)}")

    def _workspace(self, n: int, h: int, w: int, ready: bool = False, code: int | None = None) -> torch.Tensor:
        """The stream's workspace.  Its arrival counters lie where (n, h, w) puts them and are zero between calls OF THAT SHAPE: the
        transform uses the entry point that relies on it (include/stainx_hip.h: sx_reinhard_transform_ready) and has the workspace
        zero-filled first whenever the last call on it had another shape (or there was none)."""
        # (`code`: the transform's element type -- a float32 batch gets room for its tiles' 8-bit codes behind the workspace proper)
        base = int(self._lib.sx_reinhard_workspace_bytes(n, h, w))
        ws = self._scratch.get(base if code is None else int(self._lib.sx_reinhard_workspace_bytes_for(code, n, h, w)), self.device)
        shapes = self.__dict__.setdefault("_ws_shape", {})
        key = (int(n), int(h), int(w))
        if ready and shapes.get(ws.data_ptr()) != key:      # (the workspace proper: the codes behind it need no clearing)
            _native.check(self._lib.sx_reinhard_workspace_init(ws.data_ptr(), base, _native.stream_ptr(self.device)), "sx_reinhard_workspace_init")
        shapes[ws.data_ptr()] = key
        self.last_workspace = ws
        return ws

    def workspace_status(self) -> int:
        """Bit 0: a transform found the workspace not in its ready state (synchronises; for tests and diagnosis)."""
        ws = getattr(self, "last_workspace", None)
        off = int(self._lib.sx_reinhard_workspace_status_offset())
        return 0 if ws is None else int(ws[off:off + 4].view(torch.int32).item())

    def compute_reference_mean_std(self, images: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        images = images.to(self.device)
        self._check(images)
        images = images.contiguous()
        n, _, h, w = images.shape
        mean = torch.empty(3, dtype=torch.float32, device=self.device)
        std = torch.empty(3, dtype=torch.float32, device=self.device)
        with _native.on_device(self.device):
            ws = self._workspace(n, h, w)
            rc = self._lib.sx_reinhard_fit(images.data_ptr(), _dtype_code(images), n, h, w, mean.data_ptr(), std.data_ptr(),
                                           ws.data_ptr(), ws.numel(), _native.stream_ptr(self.device))
        _native.check(rc, "sx_reinhard_fit")
        return mean, std

    compute_reference_mean_std_torch = compute_reference_mean_std

    def transform(self, images: torch.Tensor, reference_mean: torch.Tensor, reference_std: torch.Tensor) -> torch.Tensor:
        images = images.to(self.device)
        self._check(images)
        images = images.contiguous()
        mean, std = self._f32(reference_mean).flatten(), self._f32(reference_std).flatten()
        if mean.numel() != 3 or std.numel() != 3:
            raise ValueError("reference_mean / reference_std must have 3 elements")
        n, _, h, w = images.shape
        out = torch.empty_like(images)
        if images.numel() == 0:
            return out
        with _native.on_device(self.device):
            ws = self._workspace(n, h, w, ready=True, code=_dtype_code(images))
            rc = self._lib.sx_reinhard_transform_ready(images.data_ptr(), out.data_ptr(), _dtype_code(images), n, h, w, mean.data_ptr(),
                                                       std.data_ptr(), ws.data_ptr(), ws.numel(), _native.stream_ptr(self.device))
        if rc != 0:
            self.__dict__.get("_ws_shape", {}).pop(ws.data_ptr(), None)      # (a failed call may have left counters behind)
        _native.check(rc, "sx_reinhard_transform_ready")
        return out


    # ---- batch statistics pooled across ranks (see stainx_amd/distributed.py) --------------------------
    def local_sums(self, images: torch.Tensor) -> torch.Tensor:
        """6 fp64 values: per channel sum and sum of squares of (LAB - 128) over this rank's pixels."""
        images = images.to(self.device)
        self._check(images)
        images = images.contiguous()
        n, _, h, w = images.shape
        sums = torch.empty(6, dtype=torch.float64, device=self.device)
        with _native.on_device(self.device):
            ws = self._workspace(n, h, w)
            rc = self._lib.sx_reinhard_sums(images.data_ptr(), _dtype_code(images), n, h, w, sums.data_ptr(), ws.data_ptr(), ws.numel(),
                                            _native.stream_ptr(self.device))
        _native.check(rc, "sx_reinhard_sums")
        return sums

    def apply_with_sums(self, images: torch.Tensor, sums: torch.Tensor, n_total_pixels: int, reference_mean: torch.Tensor, reference_std: torch.Tensor) -> torch.Tensor:
        images = images.to(self.device)
        self._check(images)
        images = images.contiguous()
        mean, std = self._f32(reference_mean).flatten(), self._f32(reference_std).flatten()
        sums = sums.to(self.device, torch.float64).contiguous()
        n, _, h, w = images.shape
        out = torch.empty_like(images)
        with _native.on_device(self.device):
            ws = self._workspace(n, h, w)
            rc = self._lib.sx_reinhard_apply(images.data_ptr(), out.data_ptr(), _dtype_code(images), n, h, w, sums.data_ptr(), float(n_total_pixels),
                                             mean.data_ptr(), std.data_ptr(), ws.data_ptr(), ws.numel(), _native.stream_ptr(self.device))
        _native.check(rc, "sx_reinhard_apply")
        return out


class HistogramMatchingHIP(TorchHIPBackendBase):
    """Histogram matching on the GPU (numerics of HistogramMatchingTorch, torch_backend.py:134-301)."""

    def __init__(self, device: str | torch.device | None = None, channel_axis: int = 1, diag: bool = False):
        super().__init__(device)
        if diag:      # tests / tools: the diagnostic build (the one-launch design study lives there)
            self._lib = _native.require_diag()
        self.channel_axis = channel_axis
        self.last_workspace: torch.Tensor | None = None
        # the workspace is zero-filled when it is made and every library call leaves it zeroed again: the sx_hm_*_ready entry
        # points, which have no clearing launch in front of the histogram pass (include/stainx_hip.h)
        self._scratch = _native.Scratch(zeroed=True)
        self._status_offset = int(self._lib.sx_hm_workspace_status_offset())

    def _check_ready_call(self, rc: int, what: str) -> None:
        if rc != 0:
            self._scratch.drop(self.device)      # (a failed call may have left counters behind)
        _native.check(rc, what)

    def workspace_status(self) -> int:
        """Bit 0: a call found the workspace not in its ready state (synchronises; for tests and diagnosis)."""
        ws = self.last_workspace
        return 0 if ws is None else int(ws[self._status_offset:self._status_offset + 4].view(torch.int32).item())

    def _channels_last(self, images: torch.Tensor) -> bool:
        return self.channel_axis == -1 or (self.channel_axis == 3 and images.ndim == 4)      # torch_backend.py:182

    def _dims(self, images: torch.Tensor) -> tuple[int, int, int, bool]:
        if images.dim() != 4:
            raise ValueError(f"HistogramMatching expects 4D images, got shape {tuple(images.shape)}")
        last = self._channels_last(images)
        chans = images.shape[-1] if last else images.shape[1]
        if chans != 3:
            raise ValueError(f"HistogramMatching expects 3 channels, got {chans} with shape {tuple(images.shape)}")
        n, h, w = (images.shape[0], images.shape[1], images.shape[2]) if last else (images.shape[0], images.shape[2], images.shape[3])
        return n, h, w, last

    def compute_reference_histograms(self, images: torch.Tensor) -> list[torch.Tensor]:
        """Three normalised 256-bin histograms, what ``transform`` receives (torch_backend.py:139-160)."""
        images = images.to(self.device).contiguous()
        n, h, w, last = self._dims(images)
        hists = torch.empty((3, 256), dtype=torch.float32, device=self.device)
        with _native.on_device(self.device):
            ws = self._scratch.get(self._lib.sx_hm_workspace_bytes(n, h, w), self.device)
            rc = self._lib.sx_hm_fit_ready(images.data_ptr(), _dtype_code(images), n, h, w, int(last), hists.data_ptr(), ws.data_ptr(), ws.numel(),
                                           _native.stream_ptr(self.device))
        self._check_ready_call(rc, "sx_hm_fit_ready")
        return [hists[c] for c in range(3)]

    def _stack_reference(self, reference_histogram, chans: int) -> torch.Tensor:
        # list -> (C,256): pad with the first histogram / trim (torch_cuda_backend.py:51-74).  The normaliser hands over the same
        # fitted tensors on every call: the stacked copy is kept while they are the same objects at the same version (the
        # torch.stack was a 5 us kernel and a launch gap in front of every transform).
        # Not cached: tensors made under torch.inference_mode() (they track no version: an in-place change would go unseen -- and
        # reading `_version` raises), and anything while a stream capture is on (the stacked copy would only be filled at replay).
        parts = tuple(reference_histogram) if isinstance(reference_histogram, (list, tuple)) else (reference_histogram,)
        cacheable = (all(isinstance(p, torch.Tensor) and not p.is_inference() for p in parts)
                     and not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()))
        if not cacheable:
            return self._stack_reference_uncached(reference_histogram, chans)
        stamp = tuple((p.data_ptr(), tuple(p.shape), p._version) for p in parts)
        cached = getattr(self, "_ref_cache", None)
        if (cached is not None and cached[0] == chans and len(cached[1]) == len(parts)
                and all(p is q for p, q in zip(parts, cached[1])) and cached[2] == stamp):
            return cached[3]
        ref = self._stack_reference_uncached(reference_histogram, chans)
        self._ref_cache = (chans, parts, stamp, ref)      # (holds the sources: their identities cannot be reused meanwhile)
        return ref

    def _stack_reference_uncached(self, reference_histogram, chans: int) -> torch.Tensor:
        if isinstance(reference_histogram, (list, tuple)):
            if len(reference_histogram) == 0:
                raise ValueError("reference_histogram list cannot be empty")
            for i, hist in enumerate(reference_histogram):
                if not isinstance(hist, torch.Tensor):
                    raise TypeError(f"reference_histogram[{i}] must be a torch.Tensor, got {type(hist)}")
                if hist.dim() != 1 or hist.size(0) != 256:
                    raise ValueError(f"Each histogram in reference_histogram list must be 1D with 256 elements. Got histogram at index {i} with shape {hist.shape}")
            rows = [h.to(self.device) for h in reference_histogram]
            while len(rows) < chans:
                rows.append(rows[0])
            ref = torch.stack(rows[:chans], dim=0)
        else:
            ref = reference_histogram.to(self.device)
            if ref.dim() != 1 or ref.size(0) != 256:
                raise ValueError(f"reference_histogram must be 1D with 256 elements. Got shape {ref.shape}")
            ref = ref.unsqueeze(0).expand(chans, 256)
        return ref.to(torch.float32).contiguous()

    def transform(self, images: torch.Tensor, reference_histogram) -> torch.Tensor:
        images = images.to(self.device).contiguous()
        n, h, w, last = self._dims(images)
        ref = self._stack_reference(reference_histogram, 3)
        out = torch.empty_like(images)
        if images.numel() == 0:
            return out
        with _native.on_device(self.device):
            ws = self._scratch.get(self._lib.sx_hm_workspace_bytes(n, h, w), self.device)
            rc = self._lib.sx_hm_transform_ready(images.data_ptr(), out.data_ptr(), _dtype_code(images), n, h, w, int(last), ref.data_ptr(),
                                                 ws.data_ptr(), ws.numel(), _native.stream_ptr(self.device))
        self._check_ready_call(rc, "sx_hm_transform_ready")
        self.last_workspace = ws
        return out

    # ---- source histogram pooled across ranks (see stainx_amd/distributed.py) --------------------------
    def local_counts(self, images: torch.Tensor) -> torch.Tensor:
        images = images.to(self.device).contiguous()
        n, h, w, last = self._dims(images)
        counts = torch.empty((3, 256), dtype=torch.int64, device=self.device)
        with _native.on_device(self.device):
            ws = self._scratch.get(self._lib.sx_hm_workspace_bytes(n, h, w), self.device)
            rc = self._lib.sx_hm_counts_ready(images.data_ptr(), _dtype_code(images), n, h, w, int(last), counts.data_ptr(), ws.data_ptr(), ws.numel(),
                                              _native.stream_ptr(self.device))
        self._check_ready_call(rc, "sx_hm_counts_ready")
        return counts

    def apply_with_counts(self, images: torch.Tensor, counts: torch.Tensor, n_total_pixels: int, reference_histogram) -> torch.Tensor:
        images = images.to(self.device).contiguous()
        n, h, w, last = self._dims(images)
        ref = self._stack_reference(reference_histogram, 3)
        counts = counts.to(self.device, torch.int64).contiguous()
        out = torch.empty_like(images)
        with _native.on_device(self.device):
            ws = self._scratch.get(self._lib.sx_hm_workspace_bytes(n, h, w), self.device)
            rc = self._lib.sx_hm_apply(images.data_ptr(), out.data_ptr(), _dtype_code(images), n, h, w, int(last), counts.data_ptr(), float(n_total_pixels),
                                       ref.data_ptr(), ws.data_ptr(), ws.numel(), _native.stream_ptr(self.device))
        _native.check(rc, "sx_hm_apply")
        self.last_workspace = ws
        return out

    def tables(self) -> dict[str, torch.Tensor]:
        """Pooled integer source histogram (3,256) and float LUT (3,256) of the last transform."""
        ws = self.last_workspace
        counts = ws[: 3 * 256 * 4].view(torch.int32).reshape(3, 256).cpu().long()
        lut = ws[3 * 256 * 4 : 2 * 3 * 256 * 4].view(torch.float32).reshape(3, 256).cpu()
        return {"counts": counts, "lut": lut}
