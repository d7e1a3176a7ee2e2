#!/usr/bin/env python3
"""Headline benchmark: megapixels/s of ``Macenko.transform`` on 64x3x512x512 fp32 tiles per GPU
(BASELINE.json configs[1]), inputs resident in HBM, reference-mode (fit once, excluded from timing).

    python bench.py --gpus N --steps K --warmup W

For N > 1 the driver launches it under ``torch.distributed.run`` (one rank per GPU, RCCL).  Tiles are
independent units, so every rank transforms its own 64-tile batch with no data-path collective
 and the job is weak-scaled: value = N * pixels_per_rank / max-over-ranks time.

One JSON line on stdout (rank 0).  Besides the contract fields it carries
  roofline     -- algorithmic bytes (24 B/px: one fp32 read + one fp32 write of every pixel)
                  of one transform call / its duration measured with HIP events on the launch stream,
                  against the 8 TB/s HBM3E peak of MI355X_MICROARCH.md.  A transform is several launches
                  (see `launches`); the figure prices ALL of them, not only the biggest kernel.
  cpu_baseline -- the CPU oracle (a numpy port of the reference's backend="torch" path) timed on this
                  box's host cores over a bounded sample of the same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
TRAFFIC_FILE = ROOT / "profiles" / "r01_macenko_cfg2_hbm_traffic.json"   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, see DESIGN.md
TILES, HEIGHT, WIDTH = 64, 512, 512
BYTES_PER_PIXEL = 24           # fp32 in + fp32 out, 3 channels


def parse() -> argparse.Namespace:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # A step is 0.2 ms.  The boxes this was measured on stall a process for ~0.1 s now and then (seen three times in some sixty
    # timed loops): inside 100 steps that multiplies ms_per_step by five, inside 1000 steps (0.2 s) it adds half.
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--cpu-tiles", type=int, default=64, help="tiles of the workload the CPU baseline is timed on")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    return ap.parse_args()


def cpu_baseline(n_tiles: int, x_cpu: torch.Tensor, he, max_c) -> dict:
    """Oracle (port of the reference CPU path) on the first n_tiles tiles of the workload."""
    from oracle import stain_oracle as so

    from threadpoolctl import threadpool_limits

    sample = x_cpu[:n_tiles].numpy()
    with threadpool_limits(limits=1):                    # one core, so that `cores` is what was really used
        so.macenko_transform(sample[:1], he, max_c)      # warm-up
        reps, t0 = 0, time.perf_counter()
        while True:                                      # repeat the sample until ~10 s of CPU work have been timed
            so.macenko_transform(sample, he, max_c)
            reps += 1
            dt = time.perf_counter() - t0
            if dt >= 10.0 or reps >= 64:
                break
    return {"value": round(reps * n_tiles * HEIGHT * WIDTH / 1e6 / dt, 3), "unit": "megapixels/s", "cores": 1, "kind": "port",
            "sample": f"{reps} x {n_tiles} of the {TILES} tiles (512x512 fp32), numpy oracle on one core (BLAS threads limited to 1), {dt:.1f} s; host has {os.cpu_count()} cores",
            "reference_note": "true reference (stainx 0.1.4 backend=torch, 8 cores, build container): 11.4 megapixels/s (BASELINE.md)"}


def measured_traffic():
    """HBM bytes of one transform call from the committed PMC run of this same command (rocprofv3 cannot be nested
    inside the timed process): FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, separate passes.  None if absent."""
    try:
        return int(json.loads(TRAFFIC_FILE.read_text())["total_bytes"])
    except (OSError, KeyError, ValueError):
        return None


def main() -> None:
    args = parse()
    # Only the result line may reach stdout: libraries underneath print there (RCCL's version banner at communicator
    # creation, for one), so file descriptor 1 points at stderr until the line is written.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or bool(os.environ.get("STAINX_BENCH_FORCE_DIST"))      # (the env switch lets a one-GPU box run the RCCL code path)
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist

        dist.init_process_group(backend="nccl", device_id=dev)

    import __graft_entry__ as entry

    if rank == 0:
        entry.build()
    if distributed:
        dist.barrier()

    from stainx_amd import Macenko, synth

    # synthetic Beer-Lambert tiles: rank r uses seeds 1000+64r ...
    src_u8 = synth.he_batch(TILES, HEIGHT, WIDTH, seed0=1000 + TILES * rank)
    x_cpu = synth.as_dtype(src_u8, torch.float32)
    x = x_cpu.to(dev)
    norm = Macenko(device=dev, backend="torch_hip")
    norm.fit(synth.reference_tile(HEIGHT, WIDTH).to(dev))          # reference mode: fit once, not timed

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        out = norm.transform(x)
    barrier()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        out = norm.transform(x)
        ev[i + 1].record()
    barrier()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    step_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]
    dev_ms = sum(step_ms) / len(step_ms)
    dev_std = (sum((v - dev_ms) ** 2 for v in step_ms) / max(len(step_ms) - 1, 1)) ** 0.5

    if rank == 0:
        pixels = TILES * HEIGHT * WIDTH
        ms_per_step = elapsed / args.steps * 1e3
        value = world * pixels / 1e6 / (elapsed / args.steps)
        achieved = pixels * BYTES_PER_PIXEL / (dev_ms * 1e-3) / 1e9
        line = {
            "metric": "megapixels/sec Macenko transform, 64x3x512x512 fp32; max-abs vs torch CPU",
            "value": round(value, 1), "unit": "megapixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Macenko reference-mode transform, 64x3x512x512 fp32 per GPU (BASELINE configs[1])",
                       "tiles_per_gpu": TILES, "height": HEIGHT, "width": WIDTH, "parallelism": f"tiles sharded over {world} GPU(s), no collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": measured_traffic(), "algorithmic_bytes_per_launch": pixels * BYTES_PER_PIXEL,
                         "kernel": "all 7 launches of one sx_macenko_transform call (stats, plane, bracket<phi>, stain, bracket<conc>, scale, reconstruct)",
                         "device_ms_per_call": round(dev_ms, 4), "device_ms_std": round(dev_std, 4), "device_ms_min": round(min(step_ms), 4),
                         "device_ms_median": round(sorted(step_ms)[len(step_ms) // 2], 4),
                         "dominant_kernel": {"name": "reconstruct_kernel", "algorithmic_bytes": pixels * BYTES_PER_PIXEL,
                                             "note": "the only launch that moves the full 24 B/px; its rocprofv3 average is in profiles/r01_final_macenko_cfg2_kernel_stats.csv"}},
        }
        if world == 1 and not args.no_cpu:
            he = norm._stain_matrix.cpu().numpy()
            max_c = norm._target_max_conc.cpu().numpy()
            line["cpu_baseline"] = cpu_baseline(args.cpu_tiles, x_cpu, he, max_c)
            # parity of the timed output against the oracle on the same sample (0-255 scale)
            from oracle import stain_oracle as so

            want = so.macenko_transform(x_cpu[:2].numpy(), he, max_c)
            line["max_abs_vs_oracle_0_255"] = float((out[:2].cpu() - torch.from_numpy(want)).abs().max())
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
