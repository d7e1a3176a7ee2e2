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

from typing import Any

import torch
import torch.distributed as dist


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


def all_reduce_sum(t: torch.Tensor, group=None) -> torch.Tensor:
    """In-place SUM over ranks.  RCCL reduces device tensors directly; gloo is staged through the host."""
    _, size = world(group)
    if size == 1:
        return t
    if t.is_cuda and dist.get_backend(group) != "nccl":
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def all_gather_stack(t: torch.Tensor, group=None) -> torch.Tensor:
    """Every rank's ``t`` (same shape everywhere) stacked along a new leading axis, in rank order."""
    _, size = world(group)
    if size == 1:
        return t.unsqueeze(0)
    if t.is_cuda and dist.get_backend(group) != "nccl":
        host = t.cpu()
        parts = [torch.empty_like(host) for _ in range(size)]
        dist.all_gather(parts, host, group=group)
        return torch.stack(parts).to(t.device)
    parts = [torch.empty_like(t) for _ in range(size)]
    dist.all_gather(parts, t.contiguous(), group=group)
    return torch.stack(parts)


def macenko_fit_pooled(local_images: torch.Tensor, *, group=None, steps: Any | None = None, device=None, method: str = "brackets") -> tuple[torch.Tensor, torch.Tensor]:
    """``(HE (3,2), maxC (2,))`` pooled over the union of every rank's ``local_images`` (N_r,3,H,W).

    ``method="brackets"`` (default when the steps provider has the ``pfit_*`` steps): three passes over the local tiles,
    one all-reduce of 10 moments, one all-gather of the ranks' pixel samples, and per percentile stage one all-reduce of
    ~8 KB of integer counts plus one all-gather of <= 32 KB of candidate keys.  If a bracket does not hold (reported by
    the last step) the fit is repeated with ``method="radix"``: nine passes, nine all-reduces, always exact."""
    if steps is None:
        from stainx_amd.backends.torch_hip_backend import MacenkoHIP

        steps = MacenkoHIP(device if device is not None else local_images.device)
    if method not in ("brackets", "radix"):
        raise ValueError(f"method must be 'brackets' or 'radix', got {method!r}")
    if method == "brackets" and hasattr(steps, "pfit_stats"):
        result = _macenko_fit_pooled_brackets(local_images, group, steps)
        if result is not None:
            return result
    moments = all_reduce_sum(steps.dfit_moments(local_images), group)
    state = steps.dfit_begin(moments)
    for stage in (0, 1):                      # 0: angle percentiles, 1: concentration percentiles
        for _ in range(4):                    # byte-wise radix rounds
            hist = all_reduce_sum(steps.dfit_histogram(local_images, state, stage), group)
            steps.dfit_advance(state, stage, hist)
    return steps.dfit_result(state)


def _macenko_fit_pooled_brackets(local_images: torch.Tensor, group, steps):
    rank, size = world(group)
    n, _, h, w = local_images.shape
    shape = (int(n), int(h), int(w))
    dev = steps.device if hasattr(steps, "device") else local_images.device
    if size == 1:
        tiles = [int(n)]
    else:      # tiles per rank: the sample union and the pixel total need them (shapes only, no pixel data)
        tiles = all_gather_stack(torch.tensor([n], dtype=torch.int64, device=dev), group).flatten().tolist()
    n_all = int(sum(tiles)) * int(h) * int(w)
    moments, sample = steps.pfit_stats(local_images)
    moments = all_reduce_sum(moments, group)
    if size == 1:
        union, sample_count = sample, steps.pfit_sample_count(int(n), int(h), int(w))
    else:
        samples = all_gather_stack(sample, group)                           # (world, 3, 4096)
        pieces = [samples[r][:, : steps.pfit_sample_count(int(tiles[r]), int(h), int(w)) : size] for r in range(size)]
        cat = torch.cat(pieces, dim=1)[:, :4096]
        sample_count = int(cat.shape[1])
        union = torch.zeros((3, 4096), dtype=torch.float32, device=cat.device)
        union[:, :sample_count] = cat
    steps.pfit_plane(moments, n_all, union, sample_count, shape)
    share = 32768 // size          # every rank's part of the compact candidate list
    out = None
    for stage in (0, 1):
        sums = all_reduce_sum(steps.pfit_pass(local_images, stage, n_all, sample_count), group)
        compact, counts = steps.pfit_gather(sums, stage, n_all, sample_count, shape, share)
        out = steps.pfit_finish(all_gather_stack(compact, group), all_gather_stack(counts, group), stage, n_all, sample_count, shape)
    he, max_c, status = out
    # the same on every rank by construction (every rank ran the same selection on the same union), so the decision to
    # repeat with the radix rounds is collective without another exchange; one host read of 4 bytes
    if int(status.item()) != 0:
        return None
    return he, max_c


def reinhard_transform_pooled(local_images: torch.Tensor, reference_mean, reference_std, *, group=None, steps: Any | None = None, device=None) -> torch.Tensor:
    """Reinhard transform whose source LAB statistics are pooled over ALL ranks' batches."""
    if steps is None:
        from stainx_amd.backends.torch_hip_backend import ReinhardHIP

        steps = ReinhardHIP(device if device is not None else local_images.device)
    sums = all_reduce_sum(steps.local_sums(local_images), group)
    pixels = torch.tensor([local_images.shape[0] * local_images.shape[2] * local_images.shape[3]], dtype=torch.int64, device=sums.device)
    n_total = int(all_reduce_sum(pixels, group).item())
    return steps.apply_with_sums(local_images, sums, n_total, reference_mean, reference_std)


def hm_transform_pooled(local_images: torch.Tensor, reference_histogram, *, group=None, steps: Any | None = None, device=None, channel_axis: int = 1) -> torch.Tensor:
    """Histogram matching whose source histogram is pooled over ALL ranks' batches (exact integer counts)."""
    if steps is None:
        from stainx_amd.backends.torch_hip_backend import HistogramMatchingHIP

        steps = HistogramMatchingHIP(device if device is not None else local_images.device, channel_axis=channel_axis)
    counts = all_reduce_sum(steps.local_counts(local_images), group)
    n_total = int(counts[0].sum().item())
    return steps.apply_with_counts(local_images, counts, n_total, reference_histogram)
