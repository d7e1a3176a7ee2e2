#!/usr/bin/env python3
"""The hot path in five lines each: the three normalisers, the nn.Module, uint8 HWC tiles as a decoder hands them over,
the sampled `precision="fast"` mode, and how a batch is split over GPUs (one process per GPU, no collective for
`transform`).  Run on a ROCm GPU:  python examples/normalize_tiles.py
Under torchrun (`python -m torch.distributed.run --nproc-per-node N examples/normalize_tiles.py`) every rank works on
its own slice of the batch and the last section pools a Macenko fit over all ranks."""
from __future__ import annotations

import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from stainx_amd import HistogramMatching, Macenko, Reinhard, StainNormalizerTransform, synth  # noqa: E402
from stainx_amd import distributed as sxd  # noqa: E402
from stainx_amd.backends.torch_hip_backend import MacenkoHIP  # noqa: E402


def main() -> None:
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    if world > 1:
        torch.distributed.init_process_group("nccl", device_id=dev)

    reference = synth.reference_tile(256, 256)                       # uint8 (1,3,H,W): the look every tile should get
    batch = synth.he_batch(16, 256, 256, seed0=2024)                 # uint8 (N,3,H,W) synthetic H&E tiles
    lo, hi = sxd.shard_bounds(batch.shape[0], rank, world)           # this rank's tiles: transform needs no communication
    tiles = batch[lo:hi].to(dev)

    # 1. the three normalisers -- same constructor / fit / transform as `from stainx import ...`
    for cls in (Macenko, Reinhard, HistogramMatching):
        out = cls(device=dev).fit(reference.to(dev)).transform(tiles)
        print(f"[rank {rank}] {cls.__name__:18s} {tuple(out.shape)} {out.dtype}  mean {out.float().mean().item():.2f}")

    # 2. as an nn.Module in a preprocessing pipeline (float tiles in [0,1], output in [0,1])
    module = StainNormalizerTransform(method="macenko", mode="reference", reference=synth.as_dtype(reference, torch.bfloat16).to(dev))
    out = module(synth.as_dtype(batch[lo:hi], torch.bfloat16).to(dev))
    print(f"[rank {rank}] module (bf16)        {tuple(out.shape)} {out.dtype}  range [{out.float().min().item():.3f}, {out.float().max().item():.3f}]")

    # 3. uint8 HWC tiles straight from a decoder: no permute / copy / float conversion before the call
    backend = MacenkoHIP(dev)
    he, max_c = backend.compute_reference_stain_matrix(reference.to(dev))
    hwc = tiles.permute(0, 2, 3, 1).contiguous()
    out = backend.transform(hwc, he, max_c, channels_last=True)
    # ... and straight to a model's input type: uint8 HWC in, bf16 in [0, 1] out, one call (== the line above / 255, .to(bfloat16))
    model_in = backend.transform(hwc, he, max_c, channels_last=True, normalize_to_0_1=True, out_dtype=torch.bfloat16)
    assert model_in.dtype == torch.bfloat16 and model_in.shape == hwc.shape
    print(f"[rank {rank}] uint8 NHWC          {tuple(out.shape)} {out.dtype}")

    # 4. sampled percentiles: about twice as fast, mean abs error ~0.5 grey levels against the exact transform
    exact = Macenko(device=dev).fit(reference.to(dev)).transform(tiles)
    fast = Macenko(device=dev, precision="fast").fit(reference.to(dev)).transform(tiles)
    print(f"[rank {rank}] precision='fast'    mean |fast - exact| = {(fast.float() - exact.float()).abs().mean().item():.3f} grey levels")

    # 5. one stain estimate pooled over the tiles of ALL ranks (a few small collectives), identical bits on every rank
    he_pooled, max_c_pooled = sxd.macenko_fit_pooled(tiles)
    print(f"[rank {rank}] pooled fit          HE[:,0] = {[round(v, 4) for v in he_pooled[:, 0].tolist()]}  maxC = {[round(v, 4) for v in max_c_pooled.tolist()]}")
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
