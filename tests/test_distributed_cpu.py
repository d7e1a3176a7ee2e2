"""world_size-2 gloo tests (CPU) of the multi-GPU choreography in stainx_amd/distributed.py: the sequence of
small all-reduces yields, on every rank, the result of the single-process oracle on the whole batch."""
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


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world_size: int, port: int, out_dir: str):
    from tests._numpy_steps import NumpyHMSteps, NumpyMacenkoBracketSteps, NumpyMacenkoPackedSteps, NumpyMacenkoSteps, NumpyReinhardSteps

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        tiles = synth.he_batch(5, 64, 64, seed0=200, scale_step=0.04)          # 5 tiles over 2 ranks: 3 + 2
        lo, hi = sxd.shard_bounds(tiles.shape[0], rank, world_size)
        local = tiles[lo:hi]
        he, max_c = sxd.macenko_fit_pooled(local, steps=NumpyMacenkoSteps())                 # no pfit_* steps: the radix form
        bracket_steps = NumpyMacenkoBracketSteps()
        he_b, max_c_b = sxd.macenko_fit_pooled(local, steps=bracket_steps)                  # the bracket form (default)
        assert hasattr(bracket_steps, "state"), "the bracket choreography did not run"
        packed_steps = NumpyMacenkoPackedSteps()
        he_p, max_c_p = sxd.macenko_fit_pooled(local, steps=packed_steps)                   # the same with the exchanges' records packed (what the HIP steps run)
        assert hasattr(packed_steps, "state")
        noise = synth.noise_u8((5, 3, 32, 32), 11)
        ref_mean, ref_std = so.reinhard_fit(synth.noise_u8((1, 3, 32, 32), 12).numpy())
        rein = sxd.reinhard_transform_pooled(noise[lo:hi], ref_mean, ref_std, steps=NumpyReinhardSteps())
        hists = so.hm_fit(synth.noise_u8((1, 3, 32, 32), 12).numpy())
        hm = sxd.hm_transform_pooled(noise[lo:hi], hists, steps=NumpyHMSteps())
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), he=he.numpy(), max_c=max_c.numpy(), he_b=he_b.numpy(), max_c_b=max_c_b.numpy(), he_p=he_p.numpy(), max_c_p=max_c_p.numpy(), rein=rein.numpy(), hm=hm.numpy(), lo=lo, hi=hi)
    finally:
        dist.destroy_process_group()


def test_shard_bounds():
    assert [sxd.shard_bounds(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert sxd.shard_bounds(2, 3, 4) == (2, 2)
    assert sxd.world() == (0, 1)
    t = torch.ones(3)
    assert sxd.all_reduce_sum(t) is t and float(t.sum()) == 3.0          # world size 1: no-op


def test_two_rank_gloo_matches_single_process_oracle(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    tiles = synth.he_batch(5, 64, 64, seed0=200, scale_step=0.04)
    he, max_c = so.macenko_fit(tiles.numpy(), signs="positive_sum")
    for r in (r0, r1):
        np.testing.assert_allclose(r["he"], he, atol=2e-5)
        np.testing.assert_allclose(r["max_c"], max_c, rtol=2e-5)
    np.testing.assert_array_equal(r0["he"], r1["he"])                    # rank-invariant bits
    for r in (r0, r1):                                                   # bracket form: same answer, same bits on both ranks
        np.testing.assert_array_equal(r["he_b"], r["he"])
        np.testing.assert_array_equal(r["max_c_b"], r["max_c"])
        np.testing.assert_array_equal(r["he_p"], r["he"])                # packed records: the same bits again
        np.testing.assert_array_equal(r["max_c_p"], r["max_c"])
    np.testing.assert_array_equal(r0["max_c"], r1["max_c"])
    assert (int(r0["lo"]), int(r0["hi"]), int(r1["lo"]), int(r1["hi"])) == (0, 3, 3, 5)
    noise = synth.noise_u8((5, 3, 32, 32), 11).numpy()
    ref = synth.noise_u8((1, 3, 32, 32), 12).numpy()
    want_rein = so.reinhard_transform(noise, *so.reinhard_fit(ref))
    got_rein = np.concatenate([r0["rein"], r1["rein"]])
    assert np.abs(got_rein.astype(int) - want_rein.astype(int)).max() <= 1
    want_hm = so.hm_transform(noise, so.hm_fit(ref))
    np.testing.assert_array_equal(np.concatenate([r0["hm"], r1["hm"]]), want_hm)
    # and pooling matters: per-shard statistics give a different answer
    assert not np.array_equal(so.hm_transform(noise[:3], so.hm_fit(ref)), r0["hm"])


def _empty_rank_worker(rank: int, world_size: int, port: int, out_dir: str):
    from tests._numpy_steps import NumpyHMSteps, NumpyMacenkoBracketSteps, NumpyMacenkoPackedSteps

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        tiles = synth.he_batch(1, 32, 32, seed0=3)                       # one tile over two ranks: rank 1 gets nothing
        lo, hi = sxd.shard_bounds(1, rank, world_size)
        raised = []
        for call in (lambda: sxd.macenko_fit_pooled(tiles[lo:hi], steps=NumpyMacenkoBracketSteps()),
                     lambda: sxd.macenko_fit_pooled(tiles[lo:hi], steps=NumpyMacenkoPackedSteps()),
                     lambda: sxd.macenko_fit_pooled(tiles[lo:hi], steps=NumpyMacenkoBracketSteps(), method="radix"),      # (the radix form skipped the guard)
                     lambda: sxd.hm_transform_pooled(synth.noise_u8((1, 3, 16, 16), 1)[lo:hi], so.hm_fit(synth.noise_u8((1, 3, 16, 16), 2).numpy()), steps=NumpyHMSteps())):
            try:
                call()
                raised.append(False)
            except ValueError as exc:
                raised.append("at least one tile" in str(exc))
        np.savez(os.path.join(out_dir, f"empty{rank}.npz"), raised=np.array(raised))
    finally:
        dist.destroy_process_group()


def test_a_rank_without_tiles_raises_on_every_rank_instead_of_hanging(tmp_path):
    mp.spawn(_empty_rank_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for rank in (0, 1):
        assert np.load(tmp_path / f"empty{rank}.npz")["raised"].all(), rank


def _forced_worker(rank: int, world_size: int, port: int, out_dir: str):
    from tests._numpy_steps import NumpyMacenkoBracketSteps

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        calls = {"n": 0}
        real, real_ag = dist.all_reduce, dist.all_gather

        def counting(*a, **k):
            calls["n"] += 1
            return real(*a, **k)

        def counting_ag(*a, **k):
            calls["n"] += 1
            return real_ag(*a, **k)

        dist.all_reduce, dist.all_gather = counting, counting_ag
        tiles = synth.he_batch(3, 48, 48, seed0=9)
        plain = sxd.macenko_fit_pooled(tiles, steps=NumpyMacenkoBracketSteps())
        assert calls["n"] == 0                                            # one rank: the collectives are skipped ...
        sxd.FORCE_COLLECTIVES = True
        forced = sxd.macenko_fit_pooled(tiles, steps=NumpyMacenkoBracketSteps())
        assert calls["n"] == 5                                            # ... unless forced (what the one-GPU RCCL test and bench mode use): five exchanges
        np.savez(os.path.join(out_dir, "forced.npz"), same=bool(torch.equal(plain[0], forced[0]) and torch.equal(plain[1], forced[1])))
    finally:
        sxd.FORCE_COLLECTIVES = None
        dist.destroy_process_group()


def test_forced_collectives_at_world_size_one(tmp_path):
    mp.spawn(_forced_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    assert bool(np.load(tmp_path / "forced.npz")["same"])


def _resharded_worker(rank: int, world_size: int, port: int, out_dir: str, packed: bool):
    """Two calls with the same local shape on rank 0 while rank 1's shard shrinks: rank 0's cached tile counts are stale in the second
    call; the flag travels with the stage exchange, so BOTH ranks repeat the fit (nobody is left alone in a collective) and the
    result is the one of the new sharding."""
    from tests._numpy_steps import NumpyMacenkoBracketSteps, NumpyMacenkoPackedSteps

    Steps = NumpyMacenkoPackedSteps if packed else NumpyMacenkoBracketSteps
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        sxd._TILE_COUNTS.clear()
        tiles = synth.he_batch(5, 48, 48, seed0=31)
        first = sxd.macenko_fit_pooled(tiles[:3] if rank == 0 else tiles[3:5], steps=Steps())
        cached_after_first = len(sxd._TILE_COUNTS)
        second = sxd.macenko_fit_pooled(tiles[:3] if rank == 0 else tiles[3:4], steps=Steps())      # rank 1: one tile now
        third = sxd.macenko_fit_pooled(tiles[:3] if rank == 0 else tiles[3:4], steps=Steps())       # (cached again, and right)
        np.savez(os.path.join(out_dir, f"reshard{rank}.npz"), he1=first[0].numpy(), he2=second[0].numpy(), mc2=second[1].numpy(), he3=third[0].numpy(), cached=cached_after_first)
    finally:
        sxd._TILE_COUNTS.clear()
        dist.destroy_process_group()


@pytest.mark.parametrize("packed", [False, True])
def test_cached_tile_counts_are_checked_collectively(tmp_path, packed):
    mp.spawn(_resharded_worker, args=(2, _free_port(), str(tmp_path), packed), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "reshard0.npz"), np.load(tmp_path / "reshard1.npz")
    assert int(r0["cached"]) == 1
    tiles = synth.he_batch(5, 48, 48, seed0=31)
    he5, _ = so.macenko_fit(tiles.numpy(), signs="positive_sum")
    he4, mc4 = so.macenko_fit(tiles[:4].numpy(), signs="positive_sum")
    for r in (r0, r1):
        np.testing.assert_allclose(r["he1"], he5, atol=2e-5)
        np.testing.assert_allclose(r["he2"], he4, atol=2e-5)          # the NEW sharding's estimate, although rank 0's cache said [3, 2]
        np.testing.assert_allclose(r["mc2"], mc4, rtol=2e-5)
        np.testing.assert_array_equal(r["he3"], r["he2"])
    np.testing.assert_array_equal(r0["he2"], r1["he2"])


def _emptied_worker(rank: int, world_size: int, port: int, out_dir: str, packed: bool):
    """after a cached call, rank 1's shard shrinks to NO tile.  Rank 1 reads fresh counts and knows; rank 0 trusts its cache and
    goes on into the stage exchanges.  Nobody may be left alone in a collective: both ranks end with the ValueError."""
    from tests._numpy_steps import NumpyMacenkoBracketSteps, NumpyMacenkoPackedSteps

    Steps = NumpyMacenkoPackedSteps if packed else NumpyMacenkoBracketSteps
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        sxd._TILE_COUNTS.clear()
        tiles = synth.he_batch(5, 48, 48, seed0=31)
        sxd.macenko_fit_pooled(tiles[:3] if rank == 0 else tiles[3:5], steps=Steps())
        cached = len(sxd._TILE_COUNTS)
        raised = ""
        try:
            sxd.macenko_fit_pooled(tiles[:3] if rank == 0 else tiles[5:5], steps=Steps())      # rank 1: no tile now
        except ValueError as exc:
            raised = str(exc)
        # ... and the process group is still usable afterwards (no rank is stuck in a collective): one more ordinary fit
        again = sxd.macenko_fit_pooled(tiles[:3] if rank == 0 else tiles[3:5], steps=Steps())
        np.savez(os.path.join(out_dir, f"emptied{rank}.npz"), raised=raised, cached=cached, he=again[0].numpy())
    finally:
        sxd._TILE_COUNTS.clear()
        dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("packed", [False, True])
def test_a_shard_that_shrinks_to_no_tile_raises_on_every_rank(tmp_path, packed):
    mp.spawn(_emptied_worker, args=(2, _free_port(), str(tmp_path), packed), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "emptied0.npz"), np.load(tmp_path / "emptied1.npz")
    assert int(r0["cached"]) == 1      # (rank 0 did take its counts on trust in the second call)
    for r in (r0, r1):
        assert "at least one tile" in str(r["raised"]), str(r["raised"])
    np.testing.assert_array_equal(r0["he"], r1["he"])
    he5, _ = so.macenko_fit(synth.he_batch(5, 48, 48, seed0=31).numpy(), signs="positive_sum")
    np.testing.assert_allclose(r0["he"], he5, atol=2e-5)
