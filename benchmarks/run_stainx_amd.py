#!/usr/bin/env python3
"""Command-line timing of one stainx_amd method, with the flags of the reference's benchmarks/run_stainx.py:20-98
(positional method, --batch-size/--height/--width/--runs/--seed/--device) so documented invocations carry over:

    python benchmarks/run_stainx_amd.py macenko --batch-size 64 --height 512 --width 512 --runs 100

Differences: the device is always a ROCm GPU ("auto" and "cuda" both mean cuda:0; "cpu"/"mps" are refused: this
package has no CPU path), --warmup untimed calls precede the timed ones, --dtype picks the tile element type, and
--data he uses synthetic H&E tiles instead of the reference's uniform noise (Macenko on noise is numerically
degenerate).  The timed region is bracketed by HIP events on the launch stream and a device synchronise.
"""
from __future__ import annotations

import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

DTYPES = {"uint8": torch.uint8, "float16": torch.float16, "bfloat16": torch.bfloat16, "float32": torch.float32, "float64": torch.float64}


def main() -> int:
    ap = argparse.ArgumentParser(description="Timing of a single stainx_amd method on one MI355X")
    ap.add_argument("method", choices=["reinhard", "macenko", "histogram_matching"])
    ap.add_argument("--batch-size", type=int, default=128)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--channels", type=int, default=3)
    ap.add_argument("--device", default="auto", choices=["auto", "cpu", "mps", "cuda"])
    ap.add_argument("--runs", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--dtype", default="uint8", choices=sorted(DTYPES))
    ap.add_argument("--data", default="noise", choices=["noise", "he"], help="uniform noise (as the reference CLI) or synthetic H&E tiles")
    args = ap.parse_args()

    if args.device in ("cpu", "mps"):
        print(f"device '{args.device}' is not supported: stainx_amd runs on ROCm GPUs only", file=sys.stderr)
        return 2
    if not torch.cuda.is_available():
        print("no GPU visible", file=sys.stderr)
        return 2
    if args.channels != 3:
        print("stain normalisation works on 3-channel RGB tiles", file=sys.stderr)
        return 2
    from stainx_amd import HistogramMatching, Macenko, Reinhard, synth

    dev = torch.device("cuda", 0)
    n, h, w = args.batch_size, args.height, args.width
    if args.data == "he":
        ref = synth.as_dtype(synth.he_batch(n, h, w, seed0=args.seed), DTYPES[args.dtype]).to(dev)
        src = synth.as_dtype(synth.he_batch(n, h, w, seed0=args.seed + 1000), DTYPES[args.dtype]).to(dev)
    else:
        ref = synth.as_dtype(synth.noise_u8((n, 3, h, w), args.seed), DTYPES[args.dtype]).to(dev)
        src = synth.as_dtype(synth.noise_u8((n, 3, h, w), args.seed + 1), DTYPES[args.dtype]).to(dev)
    norm = {"reinhard": Reinhard, "macenko": Macenko, "histogram_matching": HistogramMatching}[args.method](device=dev)

    print(f"Device: {torch.cuda.get_device_name(0)}")
    print(f"Method: {args.method}   dtype: {args.dtype}   data: {args.data}")
    print(f"Batch size: {n}   Image size: {h}x{w}")
    norm.fit(ref)
    for _ in range(args.warmup):
        out = norm.transform(src)
    torch.cuda.synchronize(dev)
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(args.runs):
        out = norm.transform(src)
    stop.record()
    torch.cuda.synchronize(dev)
    ms = start.elapsed_time(stop) / args.runs
    print(f"Result shape: {tuple(out.shape)}   dtype: {out.dtype}")
    print(f"Time: {ms:.4f} ms per transform")
    print(f"Images per second: {n * 1e3 / ms:.1f}   megapixels per second: {n * h * w / 1e3 / ms:.1f}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
