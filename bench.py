#!/usr/bin/env python3
"""Headline benchmark: megapixels/s of ``Macenko.transform`` on 64x3x512x512 fp32 tiles per GPU
(BASELINE.json configs[1]), inputs resident in HBM, reference-mode (fit once, excluded from timing).

    python bench.py --gpus N --steps K --warmup W [--workload transform|fit_transform_pooled|hm_config3|module_config5|real_tiles|reinhard_f32]

For N > 1 the job is one rank per GPU over RCCL: either the driver launches this file under ``torch.distributed.run`` (WORLD_SIZE /
RANK / LOCAL_RANK in the environment), or -- ``python bench.py --gpus N`` as it stands -- this process starts the N ranks ITSELF
(a child ``torch.distributed.run`` on 127.0.0.1, before anything here touches a GPU) and passes their one JSON line through; it
refuses loudly when the node has fewer than N GPUs.  Tiles are independent units, so every rank transforms its own tiles with no
data-path collective and the job is weak-scaled: value = N * pixels_per_rank / max-over-ranks time.

The timed loop ROTATES over two different input batches (2 x 201 MB > the 256 MiB Infinity Cache), so a call
reads its input from HBM as a pipeline that brings fresh tiles every step does; the one-buffer figure the
reference's harness would report (its input served by the Infinity Cache from the second call on) is kept as
``roofline.device_ms_hot``.

One JSON line on stdout (rank 0).  Besides the contract fields it carries
  roofline     -- algorithmic bytes (24 B/px: one fp32 read + one fp32 write of every pixel)
                  of one transform call / its duration measured with HIP events on the launch stream,
                  against the 8 TB/s HBM3E peak of MI355X_MICROARCH.md (and against its measured 6.29 TB/s
                  copy ceiling).  A transform is several launches (see `kernel`); the figure prices ALL of them.
                  `traffic`: HBM bytes per call from the committed rocprofv3 PMC passes of this command, cited
                  only while the kernel sources still hash to what those passes ran (else null).
  cpu_baseline -- the CPU oracle (a numpy port of the reference's backend="torch" path) timed on this
                  box's host cores over a bounded sample of the same workload (rank 0, N=1 only).

``--workload hm_config3`` (BASELINE configs[2]: HistogramMatching.transform, 64x3x1024x1024 uint8, 6 B/px), ``module_config5``
(configs[4]: StainNormalizerTransform(macenko, reference), 256x3x224x224 bf16 per GPU, 12 B/px) and ``real_tiles`` (the headline
transform on 64 crops of 512x512 from the reference's own example images, tests/golden/g11_real_images.npz, instead of synthetic
tiles) print the same kind of line for the other single-GPU configurations; replicas / weak scaling as for ``transform``.

``--workload fit_transform_pooled`` (BASELINE configs[3]): every rank holds 64 tiles; one step = ONE stain estimate
pooled over all ranks' tiles (small RCCL exchanges, ``stainx_amd.distributed``) + the transform of the local tiles to it;
``collective_ms`` is the device time inside the collectives per step.  At N = 1 the collectives are forced through RCCL.
"""
from __future__ import annotations

import argparse
import hashlib
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
COPY_CEILING_GBS = 6290.0      # MI355X_MICROARCH.md: measured float4 copy
TRAFFIC_FILE = ROOT / "profiles" / "r04_macenko_cfg2_hbm_traffic.json"   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, tools/hbm_traffic.py
KERNEL_STATS = "profiles/r04_macenko_cfg2_kernel_stats.txt"
TILES, HEIGHT, WIDTH = 64, 512, 512
BYTES_PER_PIXEL = 24           # fp32 in + fp32 out, 3 channels


def parse() -> argparse.Namespace:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # A step is 0.17 ms.  The boxes this was measured on stall a process for ~0.1 s now and then: inside 100 steps that
    # multiplies ms_per_step several times, inside 1000 steps it adds half.
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", choices=("transform", "fit_transform_pooled", "hm_config3", "module_config5", "real_tiles", "reinhard_f32"), default="transform")
    ap.add_argument("--batches", type=int, default=2, help="different input batches the timed loop rotates over (1: one buffer)")
    ap.add_argument("--cpu-tiles", type=int, default=64, help="tiles of the workload the CPU baseline is timed on")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    return ap.parse_args()


def source_hash() -> str:
    """Hash of the kernel sources: ties a committed profile to the code it was taken from (no .git on the GPU box)."""
    h = hashlib.sha256()
    for f in sorted((ROOT / "stainx_amd" / "csrc").glob("*")):
        if f.suffix in (".hip", ".hpp"):
            h.update(f.name.encode())
            h.update(f.read_bytes())
    return h.hexdigest()[:16]


def _cpu_worker(args):
    """One worker process of the CPU baseline: the oracle on its share of the tiles, BLAS pinned to one thread."""
    tiles, he, max_c, seconds = args
    from threadpoolctl import threadpool_limits

    from oracle import stain_oracle as so

    with threadpool_limits(limits=1):
        so.macenko_transform(tiles[:1], he, max_c)       # warm-up (imports, first-touch)
        reps, t0 = 0, time.perf_counter()
        while True:
            so.macenko_transform(tiles, he, max_c)
            reps += 1
            dt = time.perf_counter() - t0
            if dt >= seconds or reps >= 64:
                break
    return reps * tiles.shape[0], dt


def _cpu_worker_any(args):
    """One worker process of the CPU baseline of the other workloads: `kind` names the oracle call, BLAS pinned to one thread."""
    kind, sample, params, seconds = args
    from threadpoolctl import threadpool_limits

    from oracle import stain_oracle as so

    fn = ((lambda x: so.hm_transform(x, so.hm_fit(params[0]))) if kind == "hm" else (lambda x: so.reinhard_transform(x, params[0], params[1])) if kind == "reinhard"
          else (lambda x: so.macenko_transform(x, params[0], params[1])))
    with threadpool_limits(limits=1):
        fn(sample[:1])
        reps, t0 = 0, time.perf_counter()
        while True:
            fn(sample)
            reps += 1
            dt = time.perf_counter() - t0
            if dt >= seconds or reps >= 64:
                break
    return reps * sample.shape[0], dt


def cpu_baseline_any(kind: str, sample, params, h: int, w: int, what: str) -> dict:
    """The oracle on a bounded sample of a workload, one worker PROCESS per host core this job may use (16 on a one-GPU box), each on its
    own share of the sample for ~10 s: the same protocol as the headline's cpu_baseline, so the CPU column is comparable across lines.
    (Histogram matching pools its histogram over the batch it is given: every worker's share is its own small batch -- a throughput
    figure, not a parity run.)"""
    import multiprocessing as mp

    cores = max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16, sample.shape[0]))
    shares = [sample[i::cores] for i in range(cores)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(cores) as pool:
        done = pool.map(_cpu_worker_any, [(kind, sh, params, 10.0) for sh in shares])
    wall = time.perf_counter() - t0
    rate = sum(tiles * h * w / 1e6 / dt for tiles, dt in done)
    return {"value": round(rate, 3), "unit": "megapixels/s", "cores": cores, "kind": "port",
            "sample": f"{sum(t for t, _ in done)} tile transforms of {what} by {cores} worker processes side by side (numpy oracle, BLAS threads 1 each), ~10 s each, "
                      f"{wall:.1f} s wall with start-up; host has {os.cpu_count()} cores, this job may use {cores}"}


def cpu_baseline(n_tiles: int, x_cpu: torch.Tensor, he, max_c) -> dict:
    """Oracle (port of the reference CPU path: tiles are independent there too) on the first n_tiles tiles of the workload, one
    worker PROCESS per host core this job may use (the numpy port is single-threaded; the reference spreads its torch ops over
    the cores), each on its own share of the tiles for ~10 s."""
    import multiprocessing as mp

    cores = max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16, n_tiles))      # (a one-GPU job's share of the box's host cores is 16)
    sample = x_cpu[:n_tiles].numpy()
    shares = [sample[i::cores] for i in range(cores)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(cores) as pool:
        done = pool.map(_cpu_worker, [(sh, he, max_c, 10.0) for sh in shares])
    wall = time.perf_counter() - t0
    rate = sum(tiles * HEIGHT * WIDTH / 1e6 / dt for tiles, dt in done)      # workers run side by side: their rates add
    return {"value": round(rate, 3), "unit": "megapixels/s", "cores": cores, "kind": "port",
            "sample": f"{sum(t for t, _ in done)} tile transforms of the workload's first {n_tiles} tiles (512x512 fp32) by {cores} worker processes side by side (numpy oracle, BLAS threads 1 each), "
                      f"~10 s each, {wall:.1f} s wall with start-up; host has {os.cpu_count()} cores, this job may use {cores}",
            "reference_note": "true reference (stainx 0.1.4 backend=torch, 8 cores, build container): 11.4 megapixels/s (BASELINE.md)"}


def measured_traffic():
    """HBM bytes of one transform call from the committed PMC passes of this same command (rocprofv3 cannot be nested inside
    the timed process): FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, separate passes.  None if the file is absent or was
    taken from other kernel sources than the ones in this tree."""
    try:
        doc = json.loads(TRAFFIC_FILE.read_text())
        if doc.get("source_hash") != source_hash():
            return None, None
        return int(doc["total_bytes"]), doc
    except (OSError, KeyError, ValueError):
        return None, None


def timed_loop(fn, steps: int, barrier):
    """K calls of fn(i) bracketed by barrier + synchronize; wall seconds and the HIP-event time of every call (ms)."""
    barrier()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    out = None
    for i in range(steps):
        out = fn(i)
        ev[i + 1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    return elapsed, [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)], out


def other_workload(args, dev, rank, world, barrier, max_over_ranks, real_stdout) -> None:
    """hm_config3 / module_config5 / real_tiles: the same protocol (warm-up, K timed steps over rotating batches, barrier +
    synchronize on both sides, max over ranks), roofline on the workload's own algorithmic bytes, CPU oracle on a bounded sample."""
    import numpy as np

    from oracle import stain_oracle as so
    from stainx_amd import HistogramMatching, Macenko, Reinhard, StainNormalizerTransform, synth

    n_batches = max(1, args.batches)
    if args.workload == "hm_config3":
        n, h, w, bpp, dt_name = 64, 1024, 1024, 6, "u8"
        ref = synth.noise_u8((1, 3, h, w), 42)
        batches_cpu = [synth.noise_u8((n, 3, h, w), 43 + 7 * (rank + world * b)) for b in range(n_batches)]
        norm = HistogramMatching(device=dev, backend="torch_hip").fit(ref.to(dev))
        call = norm.transform
        metric = "megapixels/sec HistogramMatching transform, 64x3x1024x1024 uint8"
        workload = "HistogramMatching transform, 64x3x1024x1024 uint8 per GPU (BASELINE configs[2])"
        kernel = "histogram (LDS sub-histograms, integer counts) + LUT + apply: 3 R + 3 R + 3 W bytes per pixel moved for the 6 algorithmic"
        cpu_fn = lambda sample: so.hm_transform(sample, so.hm_fit(ref.numpy()))
        cpu_sample, exact = batches_cpu[0][:4].numpy(), True
        cpu_kind, cpu_params, cpu_many = "hm", (ref.numpy(),), batches_cpu[0][:16].numpy()
    elif args.workload == "module_config5":
        n, h, w, bpp, dt_name = 256, 224, 224, 12, "bf16"
        ref = synth.reference_tile(h, w)
        batches_cpu = [synth.as_dtype(synth.he_batch(n, h, w, seed0=1000 + n * (rank + world * b)), torch.bfloat16) for b in range(n_batches)]
        module = StainNormalizerTransform(method="macenko", mode="reference", reference=synth.as_dtype(ref, torch.bfloat16).to(dev), device=dev, backend="torch_hip")
        call = module
        metric = "megapixels/sec StainNormalizerTransform(macenko, reference), 256x3x224x224 bf16"
        workload = "StainNormalizerTransform(method=macenko, mode=reference) on 256x3x224x224 bf16 per GPU, data on the device (BASELINE configs[4]; one replica per GPU)"
        kernel = "all launches of one sx_macenko_transform call on bf16 tiles (four-pass form: stats, plane, bracket<phi>, stain, bracket<conc>, scale, reconstruct with the fused /255)"
        he, max_c = (t.float().cpu().numpy() for t in (module.normalizer._stain_matrix, module.normalizer._target_max_conc))
        cpu_fn = lambda sample: so.macenko_transform(sample, he, max_c)
        cpu_sample, exact = batches_cpu[0][:32].float().numpy(), False
        cpu_kind, cpu_params, cpu_many = "macenko", (he, max_c), batches_cpu[0][:256].float().numpy()
    elif args.workload == "reinhard_f32":
        n, h, w, bpp, dt_name = 64, 512, 512, 24, "f32"
        ref = synth.as_dtype(synth.reference_tile(h, w), torch.float32)
        batches_cpu = [synth.as_dtype(synth.he_batch(n, h, w, seed0=1000 + n * (rank + world * b)), torch.float32) for b in range(n_batches)]
        norm = Reinhard(device=dev, backend="torch_hip").fit(ref.to(dev))
        call = norm.transform
        metric = "megapixels/sec Reinhard transform, 64x3x512x512 fp32"
        workload = "Reinhard transform (LAB statistics pooled over the batch), 64x3x512x512 fp32 per GPU (the sibling of BASELINE configs[1])"
        kernel = "statistics pass (leaves the tiles' 8-bit codes) + apply pass (reads them): 12 R + 3 W + 3 R + 12 W bytes per pixel moved for the 24 algorithmic"
        r_mean, r_std = so.reinhard_fit(ref.numpy())
        cpu_fn = lambda sample: so.reinhard_transform(sample, r_mean, r_std)
        cpu_sample, exact = batches_cpu[0][:16].numpy(), False
        cpu_kind, cpu_params, cpu_many = "reinhard", (r_mean, r_std), batches_cpu[0][:16].numpy()
    else:
        n, h, w, bpp, dt_name = 64, 512, 512, 24, "f32"
        imgs = torch.from_numpy(np.load(str(ROOT / "tests" / "golden" / "g11_real_images.npz"))["images_u8"])
        crops = torch.stack([imgs[i, :, y:y + h, x:x + w] for i in range(6) for y in range(0, 513, 128) for x in range(0, 513, 128)])      # 150 overlapping tiles
        picks = [torch.arange(b, 150, 150 / n).long()[:n] for b in range(n_batches)]
        batches_cpu = [synth.as_dtype(crops[(pk + rank) % 150], torch.float32) for pk in picks]
        norm = Macenko(device=dev, backend="torch_hip").fit(imgs[0:1].to(dev))
        call = norm.transform
        metric = "megapixels/sec Macenko transform, 64x3x512x512 fp32, REAL tissue (crops of the reference's example images)"
        workload = "Macenko reference-mode transform, 64x3x512x512 fp32 per GPU, crops of the reference's six example tiles (fit on its target image)"
        kernel = "all launches of one sx_macenko_transform call (the form the backend's feedback settles on for this data)"
        he, max_c = norm._stain_matrix.cpu().numpy(), norm._target_max_conc.cpu().numpy()
        cpu_fn = lambda sample: so.macenko_transform(sample, he, max_c)
        cpu_sample, exact = batches_cpu[0][:16].numpy(), False
        cpu_kind, cpu_params, cpu_many = "macenko", (he, max_c), batches_cpu[0].numpy()
    batches = [b.to(dev) for b in batches_cpu]
    pixels = n * h * w
    for i in range(args.warmup):
        out = call(batches[i % n_batches])
    elapsed, step_ms, out = timed_loop(lambda i: call(batches[i % n_batches]), args.steps, barrier)
    elapsed = max_over_ranks(elapsed)
    if rank != 0:
        return
    dev_ms = sum(step_ms) / len(step_ms)
    achieved = pixels * bpp / (dev_ms * 1e-3) / 1e9
    line = {"metric": metric, "value": round(world * pixels / 1e6 / (elapsed / args.steps), 1), "unit": "megapixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dt_name,
            "data": "real (tests/golden/g11_real_images.npz)" if args.workload == "real_tiles" else "synthetic",
            "config": {"workload": workload, "tiles_per_gpu": n, "height": h, "width": w, "input_batches_rotated": n_batches, "parallelism": f"replicas / tiles sharded over {world} GPU(s), no collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                         "algorithmic_bytes_per_launch": pixels * bpp, "device_ms_per_call": round(dev_ms, 4), "device_ms_min": round(min(step_ms), 4), "kernel": kernel},
            "kernel_source_hash": source_hash()}
    if args.workload == "real_tiles":
        engine = norm._get_backend_impl()
        line["form"] = "four-pass (the two-pass form left its speculative path on some tiles; the backend's feedback switched)" if int(engine._classic_left) > 0 else "two-pass"
    if world == 1 and not args.no_cpu:
        want = cpu_fn(cpu_sample)
        k = cpu_sample.shape[0]
        line["cpu_baseline"] = cpu_baseline_any(cpu_kind, cpu_many, cpu_params, h, w, f"the workload's first {cpu_many.shape[0]} tiles ({h}x{w} {dt_name})")
        # parity of a GPU result against the oracle on the same sample (HM pools its histogram over the batch it is given: the sample alone)
        got = call(batches[0][:k].contiguous()).float().cpu()
        w_t = torch.from_numpy(np.asarray(want)).float() / (255.0 if args.workload == "module_config5" else 1.0)
        line["max_abs_vs_oracle"] = float((got - w_t).abs().max())
        line["max_abs_vs_oracle_note"] = "grey levels, bit-exact expected" if exact else ("[0, 1] scale (module default normalize_to_0_1), bf16 storage" if args.workload == "module_config5" else "[0, 1] scale (float tiles in, float tiles out)" if args.workload == "reinhard_f32" else "0-255 scale")
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    print(json.dumps(line), flush=True)
    os.dup2(2, 1)


def launcher_command(n_gpus: int, argv: list[str], port: int) -> list[str]:
    """The command that starts one rank per GPU of this node for `python bench.py --gpus N ...` (what the driver's own wrapper does)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
            str(Path(__file__).resolve()), *argv]


def launch_ranks(args) -> int:
    """`--gpus N` with N > 1 and no rank environment: start the N ranks as children (this process has not touched a GPU:
    torch.cuda.device_count() does not initialise one) and return their exit code; their rank 0 prints the line."""
    import socket
    import subprocess

    have = torch.cuda.device_count()
    if have < args.gpus and not os.environ.get("STAINX_BENCH_REHEARSE"):
        raise SystemExit(f"bench.py --gpus {args.gpus}: this node has {have} GPU(s); refusing to time fewer GPUs than asked for "
                         "(STAINX_BENCH_REHEARSE=1 puts all ranks on the GPUs there are: a rehearsal of the N > 1 path, not a measurement)")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    return subprocess.run(launcher_command(args.gpus, sys.argv[1:], port), check=False).returncode


def real_tiles_record(dev, steps: int, barrier) -> dict:
    """The headline transform on 64 crops of 512 x 512 from the reference's own example images (tests/golden/g11_real_images.npz) instead of
    synthetic tiles, rotating two batches: what `--workload real_tiles` prints as its own line, in short, for the default line."""
    import numpy as np

    from stainx_amd import Macenko, synth

    path = ROOT / "tests" / "golden" / "g11_real_images.npz"
    if not path.exists():
        return {"skipped": "tests/golden/g11_real_images.npz is not in this tree"}
    imgs = torch.from_numpy(np.load(str(path))["images_u8"])
    crops = torch.stack([imgs[i, :, y:y + 512, x:x + 512] for i in range(6) for y in range(0, 513, 128) for x in range(0, 513, 128)])
    batches = [synth.as_dtype(crops[torch.arange(b, 150, 150 / 64).long()[:64]], torch.float32).to(dev) for b in range(2)]
    norm = Macenko(device=dev, backend="torch_hip").fit(imgs[0:1].to(dev))
    first = []
    for i in range(8):      # the first calls on this data, one by one: is there a cliff before any feedback?
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        norm.transform(batches[i % 2])
        e1.record()
        torch.cuda.synchronize(dev)
        first.append(e0.elapsed_time(e1))
    for i in range(30):
        norm.transform(batches[i % 2])
    elapsed, step_ms, _ = timed_loop(lambda i: norm.transform(batches[i % 2]), steps, barrier)
    dev_ms = sum(step_ms) / len(step_ms)
    engine = norm._get_backend_impl()
    fell = engine.tile_params(64)["fell_back"]
    pixels = 64 * 512 * 512
    return {"ms_per_step": round(elapsed / steps * 1e3, 4), "device_ms_per_call": round(dev_ms, 4), "device_ms_max": round(max(step_ms), 4), "device_ms_median": round(sorted(step_ms)[len(step_ms) // 2], 4),
            "frac": round(pixels * BYTES_PER_PIXEL / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "first_calls_ms": [round(v, 4) for v in first],
            "slow_slots_last_call": int(sum(bin(int(v) & 15).count("1") for v in fell)), "steps": steps,
            "data": "64 crops (512 x 512) of the reference's six example images, float32, two batches rotated"}


def main() -> None:
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    # Only the result line may reach stdout: libraries underneath print there (RCCL's version banner at communicator
    # creation, for one), so file descriptor 1 points at stderr until the line is written.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    pooled = args.workload == "fit_transform_pooled"
    if pooled and world == 1:      # a one-GPU box still runs the RCCL path: a process group of one rank, collectives forced
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ["STAINX_FORCE_COLLECTIVES"] = "1"
    distributed = world > 1 or pooled or bool(os.environ.get("STAINX_BENCH_FORCE_DIST"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # STAINX_BENCH_REHEARSE=1: several ranks on ONE GPU with gloo for the barrier and the max over ranks -- a rehearsal of the
    # N > 1 code path (rank handling, aggregation, who prints) on a one-GPU box; RCCL refuses two ranks on one device.  Not a measurement.
    rehearse = bool(os.environ.get("STAINX_BENCH_REHEARSE")) and not pooled
    device_index = local_rank % torch.cuda.device_count() if rehearse else local_rank
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    if distributed:
        import torch.distributed as dist

        if rehearse:
            dist.init_process_group(backend="gloo")
        elif world == 1:
            dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    import __graft_entry__ as entry

    if rank == 0:
        entry.build()
    if distributed:
        dist.barrier()

    from stainx_amd import Macenko, synth
    from stainx_amd import distributed as sxd

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(seconds: float) -> float:
        t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if rehearse else dev)
        if distributed:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    if args.workload in ("hm_config3", "module_config5", "real_tiles", "reinhard_f32"):
        other_workload(args, dev, rank, world, barrier, max_over_ranks, real_stdout)
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        return

    # synthetic Beer-Lambert tiles: rank r, batch b uses seeds 1000 + 64 (r + world b) ...
    n_batches = max(1, args.batches)
    x_cpu = synth.as_dtype(synth.he_batch(TILES, HEIGHT, WIDTH, seed0=1000 + TILES * rank), torch.float32)
    batches = [x_cpu.to(dev)]
    for b in range(1, n_batches):
        batches.append(synth.as_dtype(synth.he_batch(TILES, HEIGHT, WIDTH, seed0=1000 + TILES * (rank + world * b)), torch.float32).to(dev))
    pixels = TILES * HEIGHT * WIDTH

    if pooled:
        from stainx_amd.backends.torch_hip_backend import MacenkoHIP

        be = MacenkoHIP(dev)

        def step(i):
            return sxd.macenko_fit_transform_pooled(batches[i % n_batches], steps=be)[0]

        for i in range(args.warmup):
            step(i)
        elapsed, step_ms, out = timed_loop(step, args.steps, barrier)
        elapsed = max_over_ranks(elapsed)
        probe = min(args.steps, 50)
        with sxd.collective_timer() as timer:
            for i in range(probe):
                step(i)
        collective_ms = timer.ms() / probe
        n_collectives = timer.count() // probe
        if rank == 0:
            dev_ms = sum(step_ms) / len(step_ms)
            line = {
                "metric": "megapixels/sec Macenko batch-mode fit_transform, 64x3x512x512 fp32 per GPU, stain estimate pooled over all GPUs",
                "value": round(world * pixels / 1e6 / (elapsed / args.steps), 1), "unit": "megapixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"Macenko batch-mode fit_transform, {world * TILES}x3x512x512 fp32 sharded over {world} GPU(s) with RCCL statistics exchanges (BASELINE configs[3] is this at 8 GPUs)",
                           "tiles_per_gpu": TILES, "height": HEIGHT, "width": WIDTH, "input_batches_rotated": n_batches,
                           "parallelism": f"tiles sharded over {world} GPU(s); pooled fit: 1 all-gather of (tile count, 10 fp64 moments, 48 KB sample) per rank, then per percentile stage 1 all-reduce of ~8 KB int64 counts + 1 all-gather of (2 counts, <= 32 KB candidate keys); transform local"},
                "collective_ms": round(collective_ms, 4), "collectives_per_step": n_collectives, "backend": dist.get_backend(),
                "roofline": {"bound": "hbm", "achieved": round(pixels * BYTES_PER_PIXEL / (dev_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(pixels * BYTES_PER_PIXEL / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                             "algorithmic_bytes_per_launch": pixels * BYTES_PER_PIXEL, "device_ms_per_call": round(dev_ms, 4),
                             "kernel": "pooled fit (stats, bracket passes, pool kernels, per-group stages; host choreography with five small collectives) + transform of the local tiles"},
            }
            sys.stdout.flush()
            os.dup2(real_stdout, 1)
            print(json.dumps(line), flush=True)
            os.dup2(2, 1)
        dist.barrier()
        dist.destroy_process_group()
        return

    norm = Macenko(device=dev, backend="torch_hip")
    norm.fit(synth.reference_tile(HEIGHT, WIDTH).to(dev))          # reference mode: fit once, not timed

    for i in range(args.warmup):
        out = norm.transform(batches[i % n_batches])
    elapsed, step_ms, out = timed_loop(lambda i: norm.transform(batches[i % n_batches]), args.steps, barrier)
    last_batch = (args.steps - 1) % n_batches
    elapsed = max_over_ranks(elapsed)
    # the one-buffer figure (input resident in the Infinity Cache between calls), a shorter loop
    hot_steps = min(args.steps, 300)
    for _ in range(20):
        norm.transform(batches[0])
    _, hot_ms, _ = timed_loop(lambda i: norm.transform(batches[0]), hot_steps, barrier)
    dev_ms = sum(step_ms) / len(step_ms)
    dev_std = (sum((v - dev_ms) ** 2 for v in step_ms) / max(len(step_ms) - 1, 1)) ** 0.5

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * pixels / 1e6 / (elapsed / args.steps)
        achieved = pixels * BYTES_PER_PIXEL / (dev_ms * 1e-3) / 1e9
        traffic, traffic_doc = measured_traffic()
        engine = norm._get_backend_impl()
        two_pass = bool(int(engine.tile_params(TILES)["n_candidates"].sum()) > 0 and int(engine._classic_left) == 0 and not os.environ.get("STAINX_MACENKO_CLASSIC"))
        line = {
            "metric": "megapixels/sec Macenko transform, 64x3x512x512 fp32; max-abs vs torch CPU",
            "value": round(value, 1), "unit": "megapixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Macenko reference-mode transform, 64x3x512x512 fp32 per GPU (BASELINE configs[1])",
                       "tiles_per_gpu": TILES, "height": HEIGHT, "width": WIDTH, "input_batches_rotated": n_batches,
                       "parallelism": f"tiles sharded over {world} GPU(s), no collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "frac_of_copy_ceiling": round(achieved / COPY_CEILING_GBS, 4), "copy_ceiling": COPY_CEILING_GBS,
                         "traffic": traffic, "traffic_source": (str(TRAFFIC_FILE.relative_to(ROOT)) + " (same kernel sources: " + source_hash() + ")") if traffic else None,
                         "algorithmic_bytes_per_launch": pixels * BYTES_PER_PIXEL,
                         "kernel": "all 4 launches of one sx_macenko_transform call in its two-pass form (prior, pass_a, estimate_stage, reconstruct)" if two_pass
                                   else "all 7 launches of one sx_macenko_transform call in its four-pass form (stats, plane, bracket<phi>, stain, bracket<conc>, scale, reconstruct)",
                         "device_ms_per_call": round(dev_ms, 4), "device_ms_std": round(dev_std, 4), "device_ms_min": round(min(step_ms), 4),
                         "device_ms_median": round(sorted(step_ms)[len(step_ms) // 2], 4),
                         "device_ms_hot": round(sum(hot_ms) / len(hot_ms), 4),
                         "device_ms_hot_note": "the same call over ONE input buffer (201 MB: it stays in the 256 MiB Infinity Cache between calls), what the reference's harness would time",
                         "dominant_kernel": {"name": "pass_a_kernel", "algorithmic_bytes": pixels * 12,
                                             "note": "the call's one pass over the fp32 input (12 B/px read; it also leaves the tiles as 8-bit codes, 3 B/px, which the "
                                                     f"reconstruct pass reads instead of the floats before it writes 12 B/px); rocprofv3 averages of all four launches in {KERNEL_STATS}"}},
            "kernel_source_hash": source_hash(),
        }
        if not os.environ.get("STAINX_BENCH_NO_REAL"):
            line["real_tiles"] = real_tiles_record(dev, min(args.steps, 200), lambda: torch.cuda.synchronize(dev))
        if os.environ.get("STAINX_BENCH_REHEARSE"):
            line["rehearsal"] = "ranks share ONE GPU (gloo for the barrier and the max): a run of the N > 1 code path, not a measurement"
        if world == 1 and not args.no_cpu:
            he = norm._stain_matrix.cpu().numpy()
            max_c = norm._target_max_conc.cpu().numpy()
            line["cpu_baseline"] = cpu_baseline(args.cpu_tiles, x_cpu, he, max_c)
            # parity of the timed output against the oracle on the same sample (0-255 scale)
            from oracle import stain_oracle as so

            want = so.macenko_transform(batches[last_batch][:2].cpu().numpy(), he, max_c)
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
