"""The fused form of the two-pass Macenko transform (stainx_amd/csrc/macenko_fused.hpp: pass A, the per-tile stage jobs and the
reconstruct pass in ONE launch with tile-level dependencies) against the other two forms of the same library -- the two-pass form
as four launches (the default where the two-pass form is chosen) and the four passes (SX_MACENKO_CLASSIC, the form the golden / oracle tests pin to the
reference).  All three compute every number that reaches the output with the same device functions: BIT FOR BIT, whatever the
speculation did and in whatever order the launch's workgroups happened to run.
"""
from __future__ import annotations

import pytest
import torch

from stainx_amd import _native, synth

pytestmark = pytest.mark.gpu

F32 = _native.DTYPE_CODES[torch.float32]
FUSED = _native.MACENKO_TWO_PASS | _native.MACENKO_FUSE
SM = torch.tensor(synth.HE_REF, dtype=torch.float32)
TMC = torch.tensor([1.9705, 1.0308], dtype=torch.float32)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def be(dev):
    from stainx_amd.backends.torch_hip_backend import MacenkoHIP

    return MacenkoHIP(dev, diag=True)      # the diagnostic build: its flags force the forms and the rare paths


def _same(a: torch.Tensor, b: torch.Tensor) -> bool:
    return torch.equal(a.cpu().view(torch.uint8), b.cpu().view(torch.uint8))


def _three(be, x, extra=0, **kw):
    n, _, h, w = x.shape
    lib = _native.require_diag()
    assert lib.sx_macenko_form(F32, n, h, w, FUSED | extra) == 2, "not a fused-form shape"
    assert lib.sx_macenko_form(F32, n, h, w, _native.MACENKO_TWO_PASS | extra) == 1
    fused = be.transform(x, SM, TMC, _extra_flags=FUSED | extra, **kw)
    pf = be.tile_params(n)
    four = be.transform(x, SM, TMC, _extra_flags=_native.MACENKO_TWO_PASS | extra, **kw)
    classic = be.transform(x, SM, TMC, _extra_flags=_native.MACENKO_CLASSIC, **kw)
    pc = be.tile_params(n)
    return fused, pf, four, classic, pc


def _check(fused, pf, four, classic, pc, what):
    for k in ("n_kept", "use_all", "vecs", "he", "max_c", "phi_lo", "phi_hi", "cov"):
        assert torch.equal(pf[k], pc[k]), (what, k, (pf[k].double() - pc[k].double()).abs().max())
    assert _same(fused, classic), (what, "fused vs four-pass", (fused.double() - classic.double()).abs().max().item())
    assert _same(fused, four), (what, "fused vs four launches")


@pytest.mark.parametrize("shape", [(64, 512, 512), (64, 256, 256), (40, 384, 384), (256, 128, 128), (6, 512, 512), (3, 128, 128), (5, 224, 224), (7, 200, 328)])
def test_fused_equals_the_other_forms_bitwise(be, dev, shape):
    n, h, w = shape
    x = synth.as_dtype(synth.he_batch(n, h, w, seed0=7000 + h), torch.float32).to(dev)
    fused, pf, four, classic, pc = _three(be, x)
    _check(fused, pf, four, classic, pc, shape)
    assert int((pf["fell_back"] & 15).max()) == 0, (shape, pf["fell_back"], pf["n_candidates"])      # the speculation held on ordinary tiles
    assert int(pf["n_candidates"].min()) > 0


def test_fused_unit_scale_and_run_to_run(be, dev):
    x = synth.as_dtype(synth.he_batch(16, 256, 256, seed0=31), torch.float32).to(dev)
    _check(*_three(be, x, normalize_to_0_1=True), "unit")
    a = be.transform(x, SM, TMC, _extra_flags=FUSED)
    for _ in range(5):      # the order in which the launch's workgroups run is not fixed: the bits are
        assert _same(a, be.transform(x, SM, TMC, _extra_flags=FUSED))
    alone = be.transform(x[3:4].contiguous(), SM, TMC, _extra_flags=FUSED)
    assert _same(alone, a[3:4])


def test_fused_slow_paths(be, dev):
    """Every proof forced to fail (SX_MACENKO_SPEC_FAIL): all four slots of every tile go through the whole-tile exact select
    INSIDE the fused launch -- its stage jobs are 256 threads wide and the reconstruct items wait for them all the same."""
    x = synth.as_dtype(synth.he_batch(3, 128, 128, seed0=5), torch.float32).to(dev)
    fused, pf, four, classic, pc = _three(be, x, extra=_native.MACENKO_SPEC_FAIL)
    assert (pf["fell_back"] & 15).eq(15).all()
    _check(fused, pf, four, classic, pc, "spec_fail")


def test_fused_tiles_without_a_stable_plane_or_without_tissue(be, dev):
    """Noise, white tiles (fewer than 3 kept pixels: every pixel is selected, the all-pixel partial sums of pass A), a small tissue
    patch, few-colour tiles (heavy ties: crowded histogram bins, the second level and the radix rounds of the 256-thread select)."""
    g = torch.Generator().manual_seed(3)
    noise = (torch.rand(2, 3, 128, 128, generator=g) * 255).round().to(torch.uint8)
    white = torch.full((1, 3, 128, 128), 250, dtype=torch.uint8)
    patch = torch.full((1, 3, 128, 128), 248, dtype=torch.uint8)
    patch[:, :, 40:72, 40:72] = synth.he_batch(1, 32, 32, seed0=9)
    few = synth.he_batch(2, 128, 128, seed0=11)
    few[0] = (few[0] // 64) * 64 + 20
    few[1] = (few[1] // 16) * 16 + 3
    tissue = synth.he_batch(2, 128, 128, seed0=21)
    src = torch.cat([noise, white, patch, few, tissue], dim=0)
    x = synth.as_dtype(src, torch.float32).to(dev)
    _check(*_three(be, x), "odd tiles")


def test_fused_candidate_overflow_takes_the_slow_path(be, dev):
    """Tissue concentrated in a corner of a mostly saturated tile: every concentration candidate comes from a few work items.  The
    fused form has ONE dense candidate array per slot (no per-wave segments), so concentration is no problem for it; what it cannot
    hold -- more candidates than its array -- must fail over to the slow exact path, not lose records silently."""
    tiles = synth.he_batch(4, 512, 512, seed0=5)
    tiles[..., : int(512 * 0.9), :] = 255
    tiles[1, :, : int(512 * 0.5), :] = synth.he_batch(1, 512, 512, seed0=6)[0, :, : int(512 * 0.5), :]
    x = synth.as_dtype(tiles, torch.float32).to(dev)
    fused, pf, four, classic, pc = _three(be, x)
    _check(fused, pf, four, classic, pc, "concentrated tissue")
    # a tile whose pixels nearly all tie at the top of the concentration range: the candidate array overflows
    flat = synth.he_batch(2, 256, 256, seed0=40)
    flat[0, :, :, :] = flat[0, :, :1, :1]
    x = synth.as_dtype(flat, torch.float32).to(dev)
    _check(*_three(be, x), "one colour")


def test_fused_workspace_contents_never_reach_the_result(be, dev):
    x = synth.as_dtype(synth.he_batch(8, 256, 256, seed0=55), torch.float32).to(dev)
    ref = be.transform(x, SM, TMC, _extra_flags=_native.MACENKO_CLASSIC)
    for fill in (0xFF, 0x7F, 0x00):
        be.transform(x, SM, TMC, _extra_flags=FUSED)
        be.last_workspace.fill_(fill)
        assert _same(be.transform(x, SM, TMC, _extra_flags=FUSED), ref), hex(fill)


def test_fused_replays_from_a_graph_on_new_data(be, dev):
    """The prior launch zeroes the fused launch's ticket counter and per-tile counters, so a captured call replays."""
    a = synth.as_dtype(synth.he_batch(8, 256, 256, seed0=21), torch.float32).to(dev)
    b = synth.as_dtype(synth.he_batch(8, 256, 256, seed0=22), torch.float32).to(dev)
    sm, tmc = SM.to(dev), TMC.to(dev)
    x = a.clone()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):
            be.transform(x, sm, tmc, _extra_flags=FUSED)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        out = be.transform(x, sm, tmc, _extra_flags=FUSED)
    x.copy_(b)
    g.replay()
    torch.cuda.synchronize()
    replayed = out.clone()
    eager = be.transform(b, sm, tmc, _extra_flags=_native.MACENKO_CLASSIC)
    assert _same(replayed, eager)


def test_workspace_sizes_per_form():
    """a call is checked against what ITS form needs."""
    lib = _native.require_diag()
    mb = 1 << 20
    bf16, u8 = _native.DTYPE_CODES[torch.bfloat16], _native.DTYPE_CODES[torch.uint8]
    assert lib.sx_macenko_workspace_bytes_for(bf16, 256, 224, 224, 0) <= 60 * mb
    assert lib.sx_macenko_workspace_bytes_for(u8, 64, 512, 512, _native.MACENKO_CLASSIC) <= 60 * mb
    assert lib.sx_macenko_workspace_bytes_for(u8, 64, 512, 512, 0) <= 165 * mb      # (takes the two-pass form now: + its dense candidate records)
    assert lib.sx_macenko_workspace_bytes_for(F32, 64, 512, 512, 0) <= 215 * mb      # (+ the tiles' 8-bit codes, 3 bytes per pixel: 48 MB; + room for 24576 instead of 14336 candidate records per slot and their key spill)
    assert lib.sx_macenko_workspace_bytes_for(F32, 64, 512, 512, FUSED) <= 215 * mb
    assert lib.sx_macenko_workspace_bytes_for(F32, 64, 512, 512, _native.MACENKO_CLASSIC) < lib.sx_macenko_workspace_bytes_for(F32, 64, 512, 512, 0)
    assert lib.sx_macenko_workspace_bytes_for(F32, 64, 512, 512, _native.MACENKO_FUSE) <= lib.sx_macenko_workspace_bytes(64, 512, 512)


def test_router_does_not_synchronise_the_host(dev):
    """one MacenkoHIP driven alternately on two streams at a routed size (>= 4 M pixels, f32).  The telemetry
    base is kept per workspace and taken from the asynchronous read-back: no `.item()`, no blocking copy -- counted by patching
    the only two ways the backend could read device memory synchronously."""
    from stainx_amd.backends.torch_hip_backend import MacenkoHIP

    fresh = MacenkoHIP(dev, diag=True)
    x = synth.as_dtype(synth.he_batch(16, 512, 512, seed0=900), torch.float32).to(dev)
    sm, tmc = SM.to(dev), TMC.to(dev)
    ref = fresh.transform(x, sm, tmc, _extra_flags=_native.MACENKO_CLASSIC)
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    torch.cuda.synchronize()
    calls = {"item": 0, "sync": 0}
    real_item, real_sync = torch.Tensor.item, torch.cuda.Event.synchronize

    def counting_item(self):
        calls["item"] += 1
        return real_item(self)

    def counting_sync(self):
        calls["sync"] += 1
        return real_sync(self)

    torch.Tensor.item, torch.cuda.Event.synchronize = counting_item, counting_sync
    try:
        outs = []
        for step in range(8):
            with torch.cuda.stream(streams[step % 2]):
                outs.append(fresh.transform(x, sm, tmc))
            if step % 2 == 1:
                torch.cuda.synchronize()      # (the caller's own synchronisation, not the backend's: lets the read-backs arrive)
    finally:
        torch.Tensor.item, torch.cuda.Event.synchronize = real_item, real_sync
    torch.cuda.synchronize()
    assert calls == {"item": 0, "sync": 0}, calls
    assert len(fresh._tele_seen) == 2          # one base per workspace (= per stream)
    for o in outs:
        assert _same(o, ref)
