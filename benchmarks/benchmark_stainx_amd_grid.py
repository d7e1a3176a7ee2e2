#!/usr/bin/env python3
"""Batch-size x image-size timing grid of one stainx_amd method, with the flags of the reference's
benchmarks/benchmark_stainx_backend.py:83-97 (--method --image-size ... --batch-size ... --warmup --runs --seed) so the
invocations its docs give (docs/benchmarks.md:16-21) carry over:

    python benchmarks/benchmark_stainx_amd_grid.py --method macenko --image-size 128 256 512 --batch-size 32 64 --runs 100

What differs, because this package has one backend and no CPU path: ``--backend1`` is always ``torch_hip`` (``torch_cuda`` is
accepted as its alias), ``--backend2`` may be ``oracle`` -- the numpy restatement of the reference's backend="torch" CPU path,
timed on ONE tile-batch sample so that a grid still finishes -- or ``none`` (default); the "Relative Error" column is then the
GPU result against the oracle on the first image, as in the reference's table.  ``--dtype`` picks the tile element type and
``--data he`` synthetic H&E tiles instead of uniform noise (Macenko on noise has no stable stain plane).  One JSON line per grid
cell on stdout, a table at the end; timing protocol as the reference's harness: warm-up calls, then ``runs`` timed calls between
device synchronisations (benchmarks/utils.py:51-74), here with HIP events on the launch stream."""
from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
DTYPES = {"uint8": torch.uint8, "float16": torch.float16, "bfloat16": torch.bfloat16, "float32": torch.float32, "float64": torch.float64}


def main() -> int:
    ap = argparse.ArgumentParser(description="stainx_amd backend benchmark grid (flags of the reference's benchmark_stainx_backend.py)")
    ap.add_argument("--method", required=True, choices=["reinhard", "macenko", "histogram_matching"])
    ap.add_argument("--image-size", nargs="+", type=int, default=[64, 128, 256, 512], help="square image sizes")
    ap.add_argument("--channels", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=25)
    ap.add_argument("--runs", type=int, default=100)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--batch-size", nargs="+", type=int, default=[32, 64, 128])
    ap.add_argument("--backend1", default="torch_hip", choices=["torch_hip", "torch_cuda"])
    ap.add_argument("--backend2", default="none", choices=["none", "oracle", "torch"], help="'oracle' / 'torch': the CPU restatement of the reference path")
    ap.add_argument("--dtype", default="uint8", choices=sorted(DTYPES))
    ap.add_argument("--data", default=None, choices=["noise", "he"], help="default: he for macenko, noise otherwise")
    args = ap.parse_args()
    if args.channels != 3:
        print("stain normalisation works on 3-channel RGB tiles", file=sys.stderr)
        return 2
    if not torch.cuda.is_available():
        print("Error: no ROCm GPU is visible. This benchmark requires one.", file=sys.stderr)
        return 1
    from stainx_amd import HistogramMatching, Macenko, Reinhard, synth

    dev = torch.device("cuda", 0)
    dtype = DTYPES[args.dtype]
    data = args.data or ("he" if args.method == "macenko" else "noise")
    cls = {"reinhard": Reinhard, "macenko": Macenko, "histogram_matching": HistogramMatching}[args.method]
    with_oracle = args.backend2 in ("oracle", "torch")
    if with_oracle:
        from oracle import stain_oracle as so
    rows = []
    print(f"# device {torch.cuda.get_device_name(0)}; method {args.method}; dtype {args.dtype}; data {data}; warmup {args.warmup}, runs {args.runs}", file=sys.stderr)
    for batch in args.batch_size:
        for size in args.image_size:
            if data == "he":
                ref_u8, src_u8 = synth.he_batch(1, size, size, seed0=args.seed), synth.he_batch(batch, size, size, seed0=args.seed + 1000)
            else:
                ref_u8, src_u8 = synth.noise_u8((1, 3, size, size), args.seed), synth.noise_u8((batch, 3, size, size), args.seed + 1)
            ref, src = synth.as_dtype(ref_u8, dtype), synth.as_dtype(src_u8, dtype)
            norm = cls(device=dev, backend=args.backend1).fit(ref.to(dev))
            x = src.to(dev)
            for _ in range(args.warmup):
                out = norm.transform(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.runs):
                out = norm.transform(x)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.runs
            cell = {"method": args.method, "batch": batch, "size": size, "dtype": args.dtype, "backend1": args.backend1, "ms_per_call": round(ms, 4),
                    "images_per_s": round(batch / ms * 1e3, 1), "megapixels_per_s": round(batch * size * size / ms / 1e3, 1)}
            if with_oracle:
                sample = src[:1].float().numpy() if dtype == torch.bfloat16 else src[:1].numpy()
                ref_np = ref.float().numpy() if dtype == torch.bfloat16 else ref.numpy()
                t0 = time.perf_counter()
                if args.method == "macenko":
                    want = so.macenko_transform(sample, *so.macenko_fit(ref_np))
                    # (per-tile: the first image alone is the first image of the batch)
                elif args.method == "reinhard":
                    want = so.reinhard_transform(src.float().numpy() if dtype == torch.bfloat16 else src.numpy(), *so.reinhard_fit(ref_np))[:1]
                else:
                    want = so.hm_transform(src.float().numpy() if dtype == torch.bfloat16 else src.numpy(), so.hm_fit(ref_np))[:1]
                cpu_s = time.perf_counter() - t0
                n_timed = 1 if args.method == "macenko" else batch      # Reinhard / HM pool their statistics over the batch: the whole batch ran
                got = out[:1].float().cpu().numpy().astype(np.float64)
                w64 = np.asarray(want, dtype=np.float64)
                cell["backend2_images_per_s"] = round(n_timed / cpu_s, 2)
                cell["speedup"] = round(cell["images_per_s"] / (n_timed / cpu_s), 1)
                cell["relative_error"] = float(np.abs(got - w64).sum() / max(np.abs(w64).sum(), 1e-12))
            rows.append(cell)
            print(json.dumps(cell), flush=True)
    head = ["batch", "size", "img/s", "MP/s"] + (["cpu img/s", "speedup", "rel. error"] if with_oracle else [])
    print("\n" + " | ".join(f"{h:>10s}" for h in head), file=sys.stderr)
    for c in rows:
        vals = [c["batch"], c["size"], c["images_per_s"], c["megapixels_per_s"]] + ([c["backend2_images_per_s"], c["speedup"], f"{c['relative_error']:.2e}"] if with_oracle else [])
        print(" | ".join(f"{str(v):>10s}" for v in vals), file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
