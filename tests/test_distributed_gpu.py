"""GPU tests of the staged (multi-rank) entry points: world size 1 directly, and two processes sharing the one
GPU of the test box over gloo (RCCL needs one device per rank; the driver's 8-GPU run uses nccl)."""
from __future__ import annotations

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import stain_oracle as so
from stainx_amd import distributed as sxd
from stainx_amd import synth
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_world_size_one_matches_fused_paths(golden):
    from stainx_amd.backends.torch_hip_backend import HistogramMatchingHIP, MacenkoHIP, ReinhardHIP

    dev = torch.device("cuda:0")
    g = golden("g3_macenko_fit.npz")
    for tag in ("pooled4x224", "pooled8x128", "single64"):
        tiles = torch.from_numpy(g[f"{tag}_u8"]).to(dev)
        he2, mc2 = MacenkoHIP(dev).compute_reference_stain_matrix(tiles)
        for method in ("brackets", "radix"):
            he, max_c = sxd.macenko_fit_pooled(tiles, method=method)
            np.testing.assert_allclose(he.cpu().numpy(), g[f"{tag}_he"], rtol=0, atol=5e-5)
            np.testing.assert_allclose(max_c.cpu().numpy(), g[f"{tag}_max_c"], rtol=1e-4, atol=0)
            assert torch.equal(he, he2) and torch.equal(max_c, mc2), method          # both staged forms == the fused fit, bit for bit
    # a pooled group far larger than one tile (2.1 M pixels): the candidate buffers scale with the group, no slow path
    many = synth.he_batch(32, 256, 256, seed0=400, scale_step=0.004).to(dev)
    be = MacenkoHIP(dev)
    he_b, mc_b = be.compute_reference_stain_matrix(many)
    assert int(be.tile_params(1)["fell_back"][0]) & 0xF == 0        # no whole-group select
    for method in ("brackets", "radix"):
        he_d, mc_d = sxd.macenko_fit_pooled(many, method=method)
        assert torch.equal(he_b, he_d) and torch.equal(mc_b, mc_d), method
    assert int(sxd._macenko_fit_pooled_brackets(many, None, be)[2].item()) == 0, "the bracket form is expected to hold on ordinary tiles (no fallback to the radix rounds)"
    he_o, mc_o = so.macenko_fit(many.cpu().numpy(), signs="positive_sum")
    np.testing.assert_allclose(he_b.cpu().numpy(), he_o, rtol=0, atol=5e-5)
    np.testing.assert_allclose(mc_b.cpu().numpy(), mc_o, rtol=1e-4, atol=0)
    noise = synth.noise_u8((3, 3, 67, 45), 43).to(dev)
    ref = synth.noise_u8((1, 3, 67, 45), 42).to(dev)
    rb = ReinhardHIP(dev)
    mean, std = rb.compute_reference_mean_std(ref)
    assert torch.equal(sxd.reinhard_transform_pooled(noise, mean, std), rb.transform(noise, mean, std))
    hb = HistogramMatchingHIP(dev)
    hists = hb.compute_reference_histograms(ref)
    assert torch.equal(sxd.hm_transform_pooled(noise, hists), hb.transform(noise, hists))


def _worker(rank: int, world_size: int, port: int, out_dir: str):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        dev = torch.device("cuda:0")
        tiles = synth.he_batch(8, 128, 128)
        lo, hi = sxd.shard_bounds(8, rank, world_size)
        he, max_c = sxd.macenko_fit_pooled(tiles[lo:hi].to(dev))                     # bracket machinery, three passes
        he_r, max_c_r = sxd.macenko_fit_pooled(tiles[lo:hi].to(dev), method="radix")   # radix rounds, nine passes
        # a sharding with unequal parts and a one-tile rank
        big = synth.he_batch(5, 96, 160, seed0=77)
        part = big[:4] if rank == 0 else big[4:]
        he_u, max_c_u = sxd.macenko_fit_pooled(part.to(dev))
        noise = synth.noise_u8((5, 3, 64, 64), 11)
        ref = synth.noise_u8((1, 3, 64, 64), 12)
        l2, h2 = sxd.shard_bounds(5, rank, world_size)
        from stainx_amd import HistogramMatching, Reinhard

        rn = Reinhard(device=dev).fit(ref)
        rein = sxd.reinhard_transform_pooled(noise[l2:h2].to(dev), rn._reference_mean, rn._reference_std)
        hn = HistogramMatching(device=dev).fit(ref)
        hm = sxd.hm_transform_pooled(noise[l2:h2].to(dev), hn._ref_histograms_256)
        torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), he=he.cpu().numpy(), max_c=max_c.cpu().numpy(), he_r=he_r.cpu().numpy(), max_c_r=max_c_r.cpu().numpy(),
                 he_u=he_u.cpu().numpy(), max_c_u=max_c_u.cpu().numpy(), rein=rein.cpu().numpy(), hm=hm.cpu().numpy())
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu(tmp_path, golden):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    g = golden("g3_macenko_fit.npz")
    assert np.array_equal(synth.he_batch(8, 128, 128).numpy(), g["pooled8x128_u8"])
    for r in (r0, r1):
        np.testing.assert_allclose(r["he"], g["pooled8x128_he"], rtol=0, atol=5e-5)
        np.testing.assert_allclose(r["max_c"], g["pooled8x128_max_c"], rtol=1e-4, atol=0)
    np.testing.assert_array_equal(r0["he"], r1["he"])
    np.testing.assert_array_equal(r0["max_c"], r1["max_c"])
    for r in (r0, r1):      # the two staged forms agree bit for bit, on every rank
        np.testing.assert_array_equal(r["he"], r["he_r"])
        np.testing.assert_array_equal(r["max_c"], r["max_c_r"])
    he_o, mc_o = so.macenko_fit(synth.he_batch(5, 96, 160, seed0=77).numpy(), signs="positive_sum")
    for r in (r0, r1):
        np.testing.assert_allclose(r["he_u"], he_o, rtol=0, atol=5e-5)
        np.testing.assert_allclose(r["max_c_u"], mc_o, rtol=1e-4, atol=0)
    np.testing.assert_array_equal(r0["he_u"], r1["he_u"])
    noise = synth.noise_u8((5, 3, 64, 64), 11).numpy()
    ref = synth.noise_u8((1, 3, 64, 64), 12).numpy()
    want_hm = so.hm_transform(noise, so.hm_fit(ref))
    np.testing.assert_array_equal(np.concatenate([r0["hm"], r1["hm"]]), want_hm)
    want_rein = so.reinhard_transform(noise, *so.reinhard_fit(ref))
    assert np.abs(np.concatenate([r0["rein"], r1["rein"]]).astype(int) - want_rein.astype(int)).max() <= 1


def _rccl_worker(rank: int, world_size: int, port: int, out_dir: str):
    """ONE rank, backend nccl (= RCCL on ROCm), collectives forced: every all_reduce / all_gather of the pooled paths really runs
    through RCCL -- fp64 moments, int64 counts, int32 candidate keys, fp32 samples."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), STAINX_FORCE_COLLECTIVES="1")
    import importlib

    importlib.reload(sxd)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=dev)
    try:
        calls = {"all_reduce": 0, "all_gather": 0}
        real_ar, real_ag, real_agt = dist.all_reduce, dist.all_gather, dist.all_gather_into_tensor

        def count_ar(*a, **k):
            calls["all_reduce"] += 1
            return real_ar(*a, **k)

        def count_ag(*a, **k):
            calls["all_gather"] += 1
            return real_ag(*a, **k)

        def count_agt(*a, **k):      # (device tensors are gathered straight into the stacked result)
            calls["all_gather"] += 1
            return real_agt(*a, **k)

        dist.all_reduce, dist.all_gather, dist.all_gather_into_tensor = count_ar, count_ag, count_agt
        from stainx_amd.backends.torch_hip_backend import HistogramMatchingHIP, MacenkoHIP, ReinhardHIP

        tiles = synth.he_batch(8, 128, 128).to(dev)
        he, max_c = sxd.macenko_fit_pooled(tiles)
        he_r, max_c_r = sxd.macenko_fit_pooled(tiles, method="radix")
        out, he_t, max_c_t = sxd.macenko_fit_transform_pooled(synth.as_dtype(synth.he_batch(8, 128, 128), torch.float32).to(dev))
        noise = synth.noise_u8((3, 3, 67, 45), 43).to(dev)
        ref = synth.noise_u8((1, 3, 67, 45), 42).to(dev)
        rb = ReinhardHIP(dev)
        mean, std = rb.compute_reference_mean_std(ref)
        rein = sxd.reinhard_transform_pooled(noise, mean, std)
        hb = HistogramMatchingHIP(dev)
        hists = hb.compute_reference_histograms(ref)
        hm = sxd.hm_transform_pooled(noise, hists)
        torch.cuda.synchronize()
        fused = MacenkoHIP(dev)
        he_f, mc_f = fused.compute_reference_stain_matrix(synth.as_dtype(synth.he_batch(8, 128, 128), torch.float32).to(dev))
        out_f = fused.transform(synth.as_dtype(synth.he_batch(8, 128, 128), torch.float32).to(dev), he_f, mc_f)
        np.savez(os.path.join(out_dir, "rccl.npz"), he=he.cpu().numpy(), max_c=max_c.cpu().numpy(), he_r=he_r.cpu().numpy(), max_c_r=max_c_r.cpu().numpy(),
                 out_equal=bool(torch.equal(out, out_f)) and bool(torch.equal(he_t, he_f)) and bool(torch.equal(max_c_t, mc_f)),
                 rein_equal=bool(torch.equal(rein, rb.transform(noise, mean, std))), hm_equal=bool(torch.equal(hm, hb.transform(noise, hists))),
                 all_reduce=calls["all_reduce"], all_gather=calls["all_gather"], backend=dist.get_backend())
    finally:
        dist.destroy_process_group()


def test_one_rank_rccl_runs_every_collective(tmp_path, golden):
    mp.spawn(_rccl_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    r = np.load(tmp_path / "rccl.npz")
    g = golden("g3_macenko_fit.npz")
    assert str(r["backend"]) == "nccl"
    # bracket fit twice (fit, fit_transform): 2 all-reduces + 3 all-gathers each (nine exchanges became five);
    # radix fit 9 + 1; Reinhard 2 + 1; HM 1 + 1
    assert int(r["all_reduce"]) == 2 * 2 + 9 + 2 + 1 and int(r["all_gather"]) == 2 * 3 + 1 + 1 + 1, (int(r["all_reduce"]), int(r["all_gather"]))
    np.testing.assert_allclose(r["he"], g["pooled8x128_he"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(r["max_c"], g["pooled8x128_max_c"], rtol=1e-4, atol=0)
    np.testing.assert_array_equal(r["he"], r["he_r"])
    np.testing.assert_array_equal(r["max_c"], r["max_c_r"])
    assert bool(r["out_equal"]) and bool(r["rein_equal"]) and bool(r["hm_equal"])


def _config4_worker(rank: int, world_size: int, port: int, out_dir: str):
    """BASELINE configs[3] at its per-rank size: 64 x 3 x 512 x 512 float32, one rank, every exchange through RCCL."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), STAINX_FORCE_COLLECTIVES="1")
    import importlib

    importlib.reload(sxd)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=dev)
    try:
        from stainx_amd.backends.torch_hip_backend import MacenkoHIP

        tiles = synth.he_batch(64, 512, 512)
        x = synth.as_dtype(tiles, torch.float32).to(dev)
        be = MacenkoHIP(dev)
        out, he, max_c = sxd.macenko_fit_transform_pooled(x, steps=be)
        # eight ranks split the compact candidate list eight ways: the same fit with an eighth of the room per rank.  (ONE rank then
        # holds eight ranks' worth of candidates of the picked bin in an eighth of the room: the list may overflow, which the last
        # step reports and the caller answers with the radix rounds -- either way the same bits.)
        he8, mc8, status8 = sxd._macenko_fit_pooled_brackets(x, None, be, _share=32768 // 8)
        if int(status8.item()) != 0:
            he8, mc8 = sxd.macenko_fit_pooled(x, steps=be, method="radix")
        he_f, mc_f = be.compute_reference_stain_matrix(x)                      # the single-GPU pooled fit
        torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, "cfg4.npz"), he=he.cpu().numpy(), max_c=max_c.cpu().numpy(), he8=he8.cpu().numpy(), mc8=mc8.cpu().numpy(), status8=int(status8.item()),
                 he_f=he_f.cpu().numpy(), mc_f=mc_f.cpu().numpy(), out_first=out[:2].cpu().numpy(), out_last=out[62:].cpu().numpy())
    finally:
        dist.destroy_process_group()


def test_config4_per_rank_size_against_the_oracle(tmp_path):
    """the pooled fit_transform at the size bench.py --workload fit_transform_pooled times (a pooled group of
    16.7 M pixels: candidate caps, the compact list's capacity and the 32-bit counters scale with it), against the CPU oracle's
    fit on the same 16.7 M pixels and its per-tile transform."""
    mp.spawn(_config4_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    r = np.load(tmp_path / "cfg4.npz")
    tiles = synth.he_batch(64, 512, 512)
    x = synth.as_dtype(tiles, torch.float32).numpy()
    he_o, mc_o = so.macenko_fit(x, signs="positive_sum")
    np.testing.assert_allclose(r["he"], he_o, rtol=0, atol=5e-5)
    np.testing.assert_allclose(r["max_c"], mc_o, rtol=1e-4, atol=0)
    # ... and against the REAL reference's pooled fit on these very tiles (tests/golden/g12: made with stainx 0.1.4, backend="torch")
    g12 = load_golden("g12_config4_pooled_fit.npz")
    np.testing.assert_allclose(r["he"], g12["rank0_f32_he"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(r["max_c"], g12["rank0_f32_max_c"], rtol=1e-4, atol=0)
    np.testing.assert_array_equal(r["he"], r["he_f"])                  # staged through RCCL == the fused single-GPU fit, bit for bit
    np.testing.assert_array_equal(r["max_c"], r["mc_f"])
    print("compact list at an eighth of its room, one rank holding everything: status", int(r["status8"]))
    np.testing.assert_array_equal(r["he8"], r["he"])
    np.testing.assert_array_equal(r["mc8"], r["max_c"])
    for got, sl in ((r["out_first"], slice(0, 2)), (r["out_last"], slice(62, 64))):
        want = so.macenko_transform(x[sl], r["he"], r["max_c"])
        assert np.abs(got.astype(np.float64) - want).max() <= 2.55e-2


def test_all_512_tiles_of_config4_pooled_on_one_gpu_against_the_reference():
    """The eight ranks' 512 tiles of BASELINE configs[3] (134 M pixels) pooled into ONE estimate by the single-GPU fit, against the REAL
    reference's fit on the same tiles (tests/golden/g12, world8_u8)."""
    from stainx_amd.backends.torch_hip_backend import MacenkoHIP

    dev = torch.device("cuda:0")
    g12 = load_golden("g12_config4_pooled_fit.npz")
    world = torch.cat([synth.he_batch(64, 512, 512, seed0=1000 + 64 * r) for r in range(8)], dim=0)
    he, max_c = MacenkoHIP(dev).compute_reference_stain_matrix(world.to(dev))
    np.testing.assert_allclose(he.cpu().numpy(), g12["world8_u8_he"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(max_c.cpu().numpy(), g12["world8_u8_max_c"], rtol=1e-4, atol=0)
