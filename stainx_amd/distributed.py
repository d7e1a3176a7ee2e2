"""Multi-GPU use of the hot path: one process per GPU, ``torch.distributed`` (backend ``"nccl"`` is RCCL
over xGMI on ROCm; ``"gloo"`` works too and is what the CPU tests use).

What shards and what has to talk:

* ``transform`` of all three normalisers treats tiles (Macenko) or the local batch (Reinhard, histogram
  matching -- like the reference run once per rank) independently: give every rank a contiguous slice of
  the batch (:func:`shard_bounds`) and call ``normalizer.transform`` -- **no collective**.
* A Macenko stain estimate pooled over a batch that is sharded across ranks
  (``compute_reference_stain_matrix_torch`` semantics, reference torch_backend.py:463-519) needs one
  small exchange per reduction stage.  Default ("brackets", three passes over the local tiles): all-reduce of
  10 fp64 moments, all-gather of the ranks' 48 KB pixel samples, then per percentile stage an all-reduce of
  ~8 KB of integer counts and an all-gather of <= 32 KB of candidate keys.  Fallback ("radix", nine passes):
  20 fp64 raw moments, then four radix rounds of 2 x 256 integer bins per stage -- 9 all-reduces of <= 4 KB.
  Both are latency-bound (tens of microseconds over xGMI); integer counts and an order-independent
  selection make the result independent of the sharding and identical on every rank.
* Reinhard / histogram-matching statistics pooled over the GLOBAL batch (single-process semantics
  across shards): one all-reduce of 6 fp64 sums / of 3 x 256 integer counts.

The local arithmetic goes through a *steps* object -- by default the HIP backend classes -- so that the
choreography (which buffers are reduced, in which order) can be exercised on CPU-only machines with a
stand-in provider in the tests.
"""
from __future__ import annotations

import os
from typing import Any

import torch
import torch.distributed as dist

# STAINX_FORCE_COLLECTIVES=1 (or force=True): a process group of ONE rank still goes through dist.all_reduce / all_gather, so that
# the RCCL path can be exercised -- and timed -- on a one-GPU box (tests/test_distributed_gpu.py, bench.py --workload fit_transform_pooled).
# Read at every call (not frozen at import); only "1" / "true" / "yes" / "on" switch it on -- "0" does not.
FORCE_COLLECTIVES: bool | None = None      # a test or a driver may set this module attribute to override the environment


def force_collectives() -> bool:
    if FORCE_COLLECTIVES is not None:
        return bool(FORCE_COLLECTIVES)
    return os.environ.get("STAINX_FORCE_COLLECTIVES", "").strip().lower() in ("1", "true", "yes", "on")


# bench.py --workload fit_transform_pooled: device time spent inside the collectives (HIP events around every dist call)
_TIMED: list | None = None
# _macenko_fit_pooled_brackets: the ranks' tile counts of the last call per (group, local shape): ([counts], the same on the device)
_TILE_COUNTS: dict = {}


class collective_timer:
    """``with collective_timer() as t: ...; t.ms()`` -- sum of the device time of every collective issued inside (synchronises)."""

    def __enter__(self):
        global _TIMED
        _TIMED = []
        return self

    def __exit__(self, *exc):
        global _TIMED
        self._pairs, _TIMED = _TIMED, None
        return False

    def ms(self) -> float:
        torch.cuda.synchronize()
        return float(sum(a.elapsed_time(b) for a, b in self._pairs))

    def count(self) -> int:
        return len(self._pairs)


def _timed(fn, *args, **kwargs):
    if _TIMED is None or not torch.cuda.is_available():
        return fn(*args, **kwargs)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    out = fn(*args, **kwargs)
    b.record()
    _TIMED.append((a, b))
    return out


def _skip_collective(size: int, force: bool | None) -> bool:
    if not (dist.is_available() and dist.is_initialized()):
        return True
    return size == 1 and not (force_collectives() if force is None else force)


def world(group=None) -> tuple[int, int]:
    """(rank, world_size); (0, 1) when torch.distributed is not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_bounds(n_items: int, rank: int, world_size: int) -> tuple[int, int]:
    """Contiguous, balanced [begin, end) of ``n_items`` units for ``rank`` (first ranks take the remainder)."""
    base, extra = divmod(int(n_items), int(world_size))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def all_reduce_sum(t: torch.Tensor, group=None, force: bool | None = None) -> torch.Tensor:
    """In-place SUM over ranks.  RCCL reduces device tensors directly; gloo is staged through the host."""
    _, size = world(group)
    if _skip_collective(size, force):
        return t
    if t.is_cuda and dist.get_backend(group) != "nccl":
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        t.copy_(host)
    else:
        _timed(dist.all_reduce, t, op=dist.ReduceOp.SUM, group=group)
    return t


def all_gather_stack(t: torch.Tensor, group=None, force: bool | None = None) -> torch.Tensor:
    """Every rank's ``t`` (same shape everywhere) stacked along a new leading axis, in rank order."""
    _, size = world(group)
    if _skip_collective(size, force):
        return t.unsqueeze(0)
    if t.is_cuda and dist.get_backend(group) != "nccl":
        host = t.cpu()
        parts = [torch.empty_like(host) for _ in range(size)]
        dist.all_gather(parts, host, group=group)
        return torch.stack(parts).to(t.device)
    if t.is_cuda:      # RCCL: straight into the stacked result (a list of parts and torch.stack was one more copy kernel per exchange)
        out = torch.empty((size, *t.shape), dtype=t.dtype, device=t.device)
        _timed(dist.all_gather_into_tensor, out, t.contiguous(), group=group)
        return out
    parts = [torch.empty_like(t) for _ in range(size)]      # (gloo has no all_gather_into_tensor)
    _timed(dist.all_gather, parts, t.contiguous(), group=group)
    return torch.stack(parts)


def tiles_per_rank(n_local: int, device, group=None) -> list[int]:
    """Every rank's tile count (one tiny all-gather).  Raises on EVERY rank if some rank holds no tile: that rank could not run
    its local steps and the others would wait in the next collective until it times out."""
    _, size = world(group)
    if _skip_collective(size, None):
        tiles = [int(n_local)]
    else:
        tiles = [int(v) for v in all_gather_stack(torch.tensor([int(n_local)], dtype=torch.int64, device=device), group).flatten().tolist()]
    if min(tiles) <= 0:
        raise ValueError(f"every rank needs at least one tile for a pooled statistic, got tiles per rank {tiles} (shard with shard_bounds over >= world_size tiles)")
    return tiles


def macenko_fit_transform_pooled(local_images: torch.Tensor, *, group=None, steps: Any | None = None, device=None, normalize_to_0_1: bool = False,
                                 method: str = "brackets") -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """``fit_transform`` of a batch sharded across ranks (BASELINE configs[3]; reference: base.py:51-61 on the pooled estimate of
    torch_backend.py:463-519): ONE stain estimate over the union of every rank's tiles becomes the target, then every rank
    transforms its own tiles to it -- the only exchanges are the fit's small statistics.  Returns ``(out, HE, maxC)``.

    The bracket form's "did every bracket hold" word is read back asynchronously and looked at AFTER the transform has been
    queued (the host never waits for the fit before it can launch the transform); in the rare case that it did not hold, the fit
    is repeated with the radix rounds and the transform with it."""
    if steps is None:
        from stainx_amd.backends.torch_hip_backend import MacenkoHIP

        steps = MacenkoHIP(device if device is not None else local_images.device)
    he, max_c, pending = _macenko_fit_pooled(local_images, group, steps, method, defer_status=True)
    out = steps.transform(local_images, he, max_c, normalize_to_0_1=normalize_to_0_1)
    if pending is not None and not pending():
        _TILE_COUNTS.clear()      # (a stale sharding may be why: the repeat reads the counts afresh)
        he, max_c, _ = _macenko_fit_pooled(local_images, group, steps, "radix", defer_status=False)
        out = steps.transform(local_images, he, max_c, normalize_to_0_1=normalize_to_0_1)
    return out, he, max_c


def macenko_fit_pooled(local_images: torch.Tensor, *, group=None, steps: Any | None = None, device=None, method: str = "brackets") -> tuple[torch.Tensor, torch.Tensor]:
    """``(HE (3,2), maxC (2,))`` pooled over the union of every rank's ``local_images`` (N_r,3,H,W).

    ``method="brackets"`` (default when the steps provider has the ``pfit_*`` steps): three passes over the local tiles and FIVE
    small exchanges -- one all-gather of every rank's {tile count, 10 fp64 moments, 48 KB pixel sample}, then per percentile stage
    one all-reduce of ~8 KB of integer counts and one all-gather of {2 counts, <= 32 KB of candidate keys}.  If a bracket does not
    hold (reported by the last step) the fit is repeated with ``method="radix"``: nine passes, nine all-reduces, always exact."""
    if steps is None:
        from stainx_amd.backends.torch_hip_backend import MacenkoHIP

        steps = MacenkoHIP(device if device is not None else local_images.device)
    he, max_c, _ = _macenko_fit_pooled(local_images, group, steps, method, defer_status=False)
    return he, max_c


def _macenko_fit_pooled(local_images, group, steps, method: str, defer_status: bool, _retried: bool = False):
    if method not in ("brackets", "radix"):
        raise ValueError(f"method must be 'brackets' or 'radix', got {method!r}")
    if method == "brackets" and hasattr(steps, "pfit_stats"):
        result = _macenko_fit_pooled_brackets(local_images, group, steps)      # (raises on EVERY rank if some rank holds no tile)
        he, max_c, status = result
        if defer_status and status.is_cuda:
            host = torch.ones(1, dtype=torch.int32).pin_memory()      # (non-zero until the copy lands: a premature look repeats the fit, never accepts it)
            with torch.cuda.device(status.device):
                host.copy_(status, non_blocking=True)
                done = torch.cuda.Event()
                done.record(torch.cuda.current_stream(status.device))      # (the stream the copy was queued on, whatever the current device is)

            def held() -> bool:
                done.synchronize()      # (a four-byte copy queued BEFORE the transform: long done by the time the host gets here)
                return int(host[0]) == 0

            return he, max_c, held
        # the same on every rank by construction (every rank ran the same selection on the same union), so the decision to
        # repeat with the radix rounds is collective without another exchange
        code = int(status.item())
        if code == 0:
            return he, max_c, None
        if code & 16 and not _retried:      # the cached tile counts were stale: once more with fresh ones (still the bracket form)
            _TILE_COUNTS.clear()
            return _macenko_fit_pooled(local_images, group, steps, method, defer_status, _retried=True)
        # (anything else -- a bracket that missed, a stale flag that survives fresh counts, a rank without tiles -- goes on to the radix rounds
        # below, behind the tile-count check EVERY rank takes part in: a rank without tiles fails there on every rank)
        tiles_per_rank(int(local_images.shape[0]), _exchange_device(steps, local_images), group)
    else:
        # (both methods: a rank without tiles must fail on EVERY rank here, not leave the others in the next collective)
        tiles_per_rank(int(local_images.shape[0]), _exchange_device(steps, local_images), group)
    moments = all_reduce_sum(steps.dfit_moments(local_images), group)
    state = steps.dfit_begin(moments)
    for stage in (0, 1):                      # 0: angle percentiles, 1: concentration percentiles
        for _ in range(4):                    # byte-wise radix rounds
            hist = all_reduce_sum(steps.dfit_histogram(local_images, state, stage), group)
            steps.dfit_advance(state, stage, hist)
    he, max_c = steps.dfit_result(state)
    return he, max_c, None


def _exchange_device(steps, local_images: torch.Tensor):
    """Where the small tensors that go through a collective live: the steps provider's device (RCCL has no CPU path; CPU-resident
    images are moved there by the steps anyway)."""
    return steps.device if hasattr(steps, "device") else local_images.device


_COUNT_BYTES: dict = {}


def _count_bytes(n: int, device) -> torch.Tensor:
    """The tile count as the eight bytes that lead a rank's record in the packed all-gather; kept per (count, device) -- a loop calls
    with the same count step after step, and a fresh torch.full is a fill kernel each time."""
    key = (int(n), str(device))
    t = _COUNT_BYTES.get(key)
    if t is None:
        if len(_COUNT_BYTES) > 64:
            _COUNT_BYTES.clear()
        t = _COUNT_BYTES[key] = torch.full((1,), int(n), dtype=torch.int64, device=device).view(torch.uint8)
    return t


def _macenko_fit_pooled_brackets(local_images: torch.Tensor, group, steps, _share: int | None = None):
    """-> (HE, maxC, status): status is a one-element int32 tensor, non-zero if a bracket missed (the caller then repeats the fit
    with the radix rounds)."""
    n, _, h, w = local_images.shape
    shape = (int(n), int(h), int(w))
    dev = _exchange_device(steps, local_images)
    skip = _skip_collective(world(group)[1], None)
    if int(n) <= 0 and skip:
        raise ValueError(f"every rank needs at least one tile for a pooled statistic, got tiles per rank {[int(n)]}")
    if not skip and hasattr(steps, "pfit_stats_packed"):
        return _macenko_fit_pooled_brackets_packed(local_images, group, steps, shape, _share)
    if int(n) > 0:
        moments, sample = steps.pfit_stats(local_images)
    else:      # a rank without tiles still has to take part in the exchange that tells every rank so
        moments, sample = torch.zeros(10, dtype=torch.float64, device=dev), torch.zeros((3, 4096), dtype=torch.float32, device=dev)
    if skip:
        tiles, union, sample_count = [int(n)], sample, steps.pfit_sample_count(int(n), int(h), int(w))
    else:
        # ONE all-gather carries what used to be three exchanges: tile count, moments, pixel sample -- as bytes
        mine = torch.cat([_count_bytes(int(n), moments.device), moments.contiguous().view(torch.uint8), sample.contiguous().view(torch.uint8).flatten()])
        got = all_gather_stack(mine, group)                                                   # (world, 8 + 80 + 49152)
        counts_dev = got[:, :8].contiguous().view(torch.int64).flatten()
        # The ranks' tile counts are needed on the HOST (pixel total, sample layout, the compact list's split), and reading them
        # stalls the host until the first pass has run -- with ~20 launches still to queue behind it.  A loop calls this with the same
        # sharding step after step: the counts of the last call with this (group, local shape) are taken on trust and CHECKED ON THE
        # DEVICE against what the all-gather brought (folded into the status word the caller reads after everything is queued; a
        # mismatch -- some rank's shard changed -- repeats the fit with fresh counts).
        key = (id(group) if group is not None else 0, int(n), int(h), int(w), int(counts_dev.numel()), str(counts_dev.device))
        cached = _TILE_COUNTS.get(key) if int(n) > 0 else None
        if cached is not None:
            tiles, stale = cached[0], (counts_dev != cached[1]).any().to(torch.int32).reshape(1)
        else:
            tiles, stale = [int(v) for v in counts_dev.tolist()], None      # (the one early host read: only the first pass is queued)
            if min(tiles) > 0:
                _TILE_COUNTS[key] = (tiles, counts_dev.clone())
        if min(tiles) <= 0:      # (known only to the ranks that read the counts afresh: see _sit_out)
            return _sit_out(group, counts_dev.device, len(tiles), int(_share) if _share else 32768 // len(tiles))
        size = len(tiles)
        moments = got[:, 8:88].contiguous().view(torch.float64).sum(dim=0)                   # the same sum on every rank
        samples = got[:, 88:].contiguous().view(torch.float32).reshape(size, 3, 4096)
        counts = [steps.pfit_sample_count(int(tiles[r]), int(h), int(w)) for r in range(size)]
        if len(set(counts)) == 1:      # equal shards: every world-th column of every rank's sample, one strided copy
            cat = samples[:, :, : counts[0] : size].permute(1, 0, 2).reshape(3, -1)[:, :4096]
        else:
            cat = torch.cat([samples[r][:, : counts[r] : size] for r in range(size)], dim=1)[:, :4096]
        sample_count = int(cat.shape[1])
        if sample_count == 4096:
            union = cat.contiguous()      # (the usual case: nothing to pad)
        else:
            union = torch.zeros((3, 4096), dtype=torch.float32, device=cat.device)
            union[:, :sample_count] = cat
    n_all = int(sum(tiles)) * int(h) * int(w)
    if n_all >= 1 << 32:
        raise ValueError(f"a pooled fit over {n_all} pixels exceeds the 2^32 the native counters hold; fit on a subset of the tiles")
    size = len(tiles)
    steps.pfit_plane(moments, n_all, union, sample_count, shape)
    share = int(_share) if _share else 32768 // size          # every rank's part of the compact candidate list (_share: tests mimic a larger world)
    out = None
    any_stale = None
    for stage in (0, 1):
        sums = all_reduce_sum(steps.pfit_pass(local_images, stage, n_all, sample_count), group)
        compact, counts = steps.pfit_gather(sums, stage, n_all, sample_count, shape, share)
        if skip:
            g_compact, g_counts = compact.unsqueeze(0), counts.unsqueeze(0)
        else:
            # the two counts travel in front of the keys: one exchange.  So does this rank's "my cached tile counts were stale" flag:
            # every rank sees every rank's, so the decision to repeat the fit is the same everywhere (a rank that repeated alone
            # would leave the others out of its collectives).
            flag = stale.to(counts.device) if stale is not None else torch.zeros(1, dtype=torch.int32, device=counts.device)
            got = all_gather_stack(torch.cat([counts.to(torch.int32).flatten(), flag.to(torch.int32), compact.to(torch.int32).flatten()]), group)
            g_counts, g_compact = got[:, :2], got[:, 3:].reshape(size, 2, share)
            if stage == 1:
                any_stale = got[:, 2].max().reshape(1)      # (the flag is the same in both stages' exchanges)
        out = steps.pfit_finish(g_compact, g_counts, stage, n_all, sample_count, shape)
    if any_stale is not None:
        he, max_c, status = out
        out = (he, max_c, torch.add(status, any_stale.to(device=status.device, dtype=status.dtype), alpha=16))      # (bit 4: some rank's cached tile counts were not this call's)
    return out


_ZERO_TILES = 16 | 32      # status of an attempt some rank had no tile for: "stale" (everybody repeats once with fresh counts) and "no tiles"


def _sit_out(group, device, size: int, share: int):
    """Some rank holds no tile.  A rank that has just READ the tile counts knows; a rank that took its cached counts on trust does not
    (its device-side check will say "stale" at the end of the attempt) and goes on into the attempt's remaining exchanges -- so the
    ranks that know take part in those with empty records instead of raising alone and leaving the others in a collective until it
    times out.  Everybody then meets again: the attempt's status is "stale | no tiles" on the ranks that sat out and
    "stale" on the others, all repeat once with fresh counts, all see the empty shard, all sit out, and all raise behind the radix
    path's tile-count exchange."""
    for _stage in (0, 1):
        all_reduce_sum(torch.zeros(1033, dtype=torch.int64, device=device), group)
        all_gather_stack(torch.zeros(3 + 2 * int(share), dtype=torch.int32, device=device), group)
    zeros = torch.zeros
    return zeros((3, 2), dtype=torch.float32, device=device), zeros(2, dtype=torch.float32, device=device), torch.full((1,), _ZERO_TILES, dtype=torch.int32, device=device)


def _macenko_fit_pooled_brackets_packed(local_images: torch.Tensor, group, steps, shape: tuple[int, int, int], _share: int | None):
    """The five exchanges with the records packed and unpacked by the steps provider (include/stainx_hip.h, sx_macenko_pfit_*_packed):
    the host moves buffers it never looks into -- except the ranks' tile counts, once per (group, local shape)."""
    n, h, w = shape
    record = steps.pfit_stats_packed(local_images) if n > 0 else steps.pfit_empty_record()
    got = all_gather_stack(record, group)                                                     # (world, record bytes)
    size = int(got.shape[0])
    # the ranks' tile counts are needed on the HOST (pixel total, sample layout, the compact list's split); a loop calls with the same
    # sharding step after step: the counts of the last call with this key are taken on trust and checked on the device against what
    # the all-gather brought -- the flag travels with the stage records and ends in the status word (bit 4)
    key = ("packed", id(group) if group is not None else 0, n, h, w, size, str(got.device))
    cached = _TILE_COUNTS.get(key) if n > 0 else None
    if cached is not None:
        tiles, expected = cached
    else:
        counts_dev = got[:, :8].contiguous().view(torch.int64).flatten()
        tiles, expected = [int(v) for v in counts_dev.tolist()], None      # (the one early host read: only the first pass is queued)
        if min(tiles) > 0:
            _TILE_COUNTS[key] = (tiles, counts_dev.clone())
    if min(tiles) <= 0:      # (known only to the ranks that read the counts afresh: see _sit_out)
        return _sit_out(group, got.device, size, int(_share) if _share else 32768 // size)
    n_all = int(sum(tiles)) * h * w
    if n_all >= 1 << 32:
        raise ValueError(f"a pooled fit over {n_all} pixels exceeds the 2^32 the native counters hold; fit on a subset of the tiles")
    counts = [steps.pfit_sample_count(int(tiles[r]), h, w) for r in range(size)]
    sample_count = min(4096, sum((c + size - 1) // size for c in counts))
    stale = steps.pfit_plane_packed(got, counts, expected, n_all, sample_count, shape)
    share = int(_share) if _share else 32768 // size
    out = None
    for stage in (0, 1):
        sums = all_reduce_sum(steps.pfit_pass(local_images, stage, n_all, sample_count), group)
        rows = all_gather_stack(steps.pfit_gather_packed(sums, stage, n_all, sample_count, shape, share, stale), group)
        out = steps.pfit_finish_packed(rows, stage, n_all, sample_count, shape, share)
    return out


def reinhard_transform_pooled(local_images: torch.Tensor, reference_mean, reference_std, *, group=None, steps: Any | None = None, device=None) -> torch.Tensor:
    """Reinhard transform whose source LAB statistics are pooled over ALL ranks' batches."""
    if steps is None:
        from stainx_amd.backends.torch_hip_backend import ReinhardHIP

        steps = ReinhardHIP(device if device is not None else local_images.device)
    tiles_per_rank(int(local_images.shape[0]), _exchange_device(steps, local_images), group)
    sums = all_reduce_sum(steps.local_sums(local_images), group)
    pixels = torch.tensor([local_images.shape[0] * local_images.shape[2] * local_images.shape[3]], dtype=torch.int64, device=sums.device)
    n_total = int(all_reduce_sum(pixels, group).item())
    return steps.apply_with_sums(local_images, sums, n_total, reference_mean, reference_std)


def hm_transform_pooled(local_images: torch.Tensor, reference_histogram, *, group=None, steps: Any | None = None, device=None, channel_axis: int = 1) -> torch.Tensor:
    """Histogram matching whose source histogram is pooled over ALL ranks' batches (exact integer counts)."""
    if steps is None:
        from stainx_amd.backends.torch_hip_backend import HistogramMatchingHIP

        steps = HistogramMatchingHIP(device if device is not None else local_images.device, channel_axis=channel_axis)
    tiles_per_rank(int(local_images.shape[0]), _exchange_device(steps, local_images), group)
    counts = all_reduce_sum(steps.local_counts(local_images), group)
    n_total = int(counts[0].sum().item())
    return steps.apply_with_counts(local_images, counts, n_total, reference_histogram)
