"""The two-pass Macenko transform (stainx_amd/csrc/macenko_twopass.hpp) against the four-pass form of the same library:
the two compute every number that reaches the output with the same device functions, so they must agree BIT FOR BIT --
outputs, stain vectors, maxC -- whatever the speculation did (held, failed its proof, or was never made).  The four-pass
form is the one the golden / oracle tests of test_macenko_gpu.py pin to the reference; those tests run the default
(two-pass) form as well.
"""
from __future__ import annotations

import numpy as np
import pytest
import torch

from stainx_amd import _native, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def be(dev):
    from stainx_amd.backends.torch_hip_backend import MacenkoHIP

    return MacenkoHIP(dev, diag=True)      # the diagnostic build: its flags force the forms and the rare paths


SM = torch.tensor(synth.HE_REF, dtype=torch.float32)
TMC = torch.tensor([1.9705, 1.0308], dtype=torch.float32)


def _both(be, x, **kw):
    two = be.transform(x, SM, TMC, _extra_flags=_native.MACENKO_TWO_PASS, **kw)
    p2 = be.tile_params(x.shape[0])
    classic = be.transform(x, SM, TMC, _extra_flags=_native.MACENKO_CLASSIC, **kw)
    p1 = be.tile_params(x.shape[0])
    return two, p2, classic, p1


def _same(a: torch.Tensor, b: torch.Tensor) -> bool:
    return torch.equal(a.cpu().view(torch.uint8), b.cpu().view(torch.uint8))


def _check_equal(two, p2, classic, p1, what):
    for k in ("n_kept", "use_all", "vecs", "he", "max_c", "phi_lo", "phi_hi", "cov"):
        assert torch.equal(p2[k], p1[k]), (what, k, (p2[k].double() - p1[k].double()).abs().max())
    assert _same(two, classic), (what, (two.double() - classic.double()).abs().max().item())


@pytest.mark.parametrize("dt", [torch.float32, torch.uint8, torch.bfloat16, torch.float16, torch.float64])
@pytest.mark.parametrize("hw", [(512, 512), (224, 224), (128, 128), (64, 64), (321, 199), (33, 47)])
def test_two_pass_equals_four_pass_bitwise(be, dev, dt, hw):
    n = 6 if hw[0] >= 224 else 3
    x = synth.as_dtype(synth.he_batch(n, *hw, seed0=4000 + hw[0]), dt).to(dev)
    two, p2, classic, p1 = _both(be, x)
    _check_equal(two, p2, classic, p1, (dt, hw))
    if hw[0] * hw[1] >= 64 * 64:
        assert int(p2["fell_back"].max()) == 0, (dt, hw, p2["fell_back"], p2["n_candidates"])      # the speculation held on ordinary tiles
        frac = p2["n_candidates"].double().sum(1) / (hw[0] * hw[1])
        assert float(frac.max()) < 0.5, frac


def test_two_pass_layouts_and_unit_scale(be, dev):
    src = synth.he_batch(4, 256, 256, seed0=77)
    for dt in (torch.float32, torch.uint8):
        x = synth.as_dtype(src, dt).to(dev)
        _check_equal(*_both(be, x, normalize_to_0_1=True), (dt, "unit"))
        xl = x.permute(0, 2, 3, 1).contiguous()
        _check_equal(*_both(be, xl, channels_last=True), (dt, "nhwc"))
    x = src.to(dev)
    _check_equal(*_both(be, x, out_dtype=torch.bfloat16, normalize_to_0_1=True), "u8->bf16")


def test_failed_speculation_takes_the_exact_slow_path(be, dev):
    """SX_MACENKO_SPEC_FAIL makes every proof fail: all four slots of every tile go through the whole-tile select."""
    x = synth.as_dtype(synth.he_batch(3, 128, 128, seed0=5), torch.float32).to(dev)
    classic = be.transform(x, SM, TMC, _extra_flags=_native.MACENKO_CLASSIC)
    p1 = be.tile_params(3)
    slow = be.transform(x, SM, TMC, _extra_flags=_native.MACENKO_SPEC_FAIL | _native.MACENKO_TWO_PASS)
    ps = be.tile_params(3)
    assert (ps["fell_back"] & 15).eq(15).all()
    _check_equal(slow, ps, classic, p1, "spec_fail")


def test_tiles_without_a_stable_plane_or_without_tissue(be, dev):
    """Noise (near-isotropic covariance: a presample cannot predict the plane), a white tile (fewer than 3 kept pixels: every
    pixel is selected), a tile with a small tissue patch, few-colour tiles: whatever the two-pass form decides, same bits."""
    g = torch.Generator().manual_seed(3)
    noise = (torch.rand(2, 3, 128, 128, generator=g) * 255).round().to(torch.uint8)
    white = torch.full((1, 3, 128, 128), 250, dtype=torch.uint8)
    patch = torch.full((1, 3, 128, 128), 248, dtype=torch.uint8)
    patch[:, :, 40:72, 40:72] = synth.he_batch(1, 32, 32, seed0=9)
    few = synth.he_batch(1, 128, 128, seed0=11)
    few = (few // 64) * 64 + 20
    tissue = synth.he_batch(2, 128, 128, seed0=21)
    src = torch.cat([noise, white, patch, few, tissue], dim=0)
    for dt in (torch.float32, torch.uint8):
        x = synth.as_dtype(src, dt).to(dev)
        two, p2, classic, p1 = _both(be, x)
        _check_equal(two, p2, classic, p1, dt)


def test_two_pass_is_deterministic_and_tile_independent(be, dev):
    x = synth.as_dtype(synth.he_batch(5, 224, 224, seed0=300), torch.bfloat16).to(dev)
    a = be.transform(x, SM, TMC, _extra_flags=_native.MACENKO_TWO_PASS)
    b = be.transform(x, SM, TMC, _extra_flags=_native.MACENKO_TWO_PASS)
    assert _same(a, b)
    alone = be.transform(x[2:3].contiguous(), SM, TMC, _extra_flags=_native.MACENKO_TWO_PASS)
    assert _same(alone, a[2:3])


def test_default_route_and_feedback(be, dev):
    """Without a flag the library takes the two-pass form where it pays (f32 / f64, >= 4 M pixels, mid-sized tiles) and the
    backend leaves it again when the library reports tiles it could not speculate on (here: two white tiles in the batch)."""
    from stainx_amd.backends.torch_hip_backend import MacenkoHIP

    fresh = MacenkoHIP(dev, diag=True)
    tiles = synth.he_batch(16, 512, 512, seed0=2000)
    x = synth.as_dtype(tiles, torch.float32).to(dev)
    out = fresh.transform(x, SM, TMC)
    assert int(fresh.tile_params(16)["n_candidates"].sum()) > 0          # candidates exist only in the two-pass form
    torch.cuda.synchronize()
    fresh.transform(x, SM, TMC)
    assert fresh._classic_left == 0
    small = fresh.transform(x[:2].contiguous(), SM, TMC)                 # 0.5 M pixels: the four-pass form
    assert int(fresh.tile_params(2)["n_candidates"].sum()) > 0 and _same(small, out[:2])      # (its candidate counters are the brackets')
    tiles[3] = 250
    tiles[9] = 250
    xb = synth.as_dtype(tiles, torch.float32).to(dev)
    first = fresh.transform(xb, SM, TMC)
    torch.cuda.synchronize()
    second = fresh.transform(xb, SM, TMC)                                # the feedback has arrived: this call is routed
    assert fresh._classic_left > 0 and _same(first, second)
    classic = fresh.transform(xb, SM, TMC, _extra_flags=_native.MACENKO_CLASSIC)
    assert _same(first, classic)
    # calls the library runs in its four-pass form anyway (small batches, narrow pixels on small tiles) take no part in the feedback:
    # no event, no side-stream copy (7 us of host time on a launch-bound call)
    quiet = MacenkoHIP(dev, diag=True)
    quiet.transform(x[:1].contiguous(), SM, TMC)
    quiet.transform(synth.as_dtype(tiles[:8], torch.uint8).to(dev), SM, TMC)
    quiet.transform(synth.as_dtype(synth.he_batch(96, 224, 224, seed0=5), torch.bfloat16).to(dev), SM, TMC)
    assert quiet._tele_event is None and quiet._tele_stream is None


def test_tissue_concentrated_in_a_few_work_items(be, dev):
    """A tile that is 90 % saturated background has all its concentration candidates in two of its sixteen work items: their
    waves' segments overflow into the tile's overflow area (not into the slow path), and the bits are the four-pass form's."""
    tiles = synth.he_batch(4, 512, 512, seed0=5)
    tiles[..., : int(512 * 0.9), :] = 255
    tiles[1, :, : int(512 * 0.5), :] = synth.he_batch(1, 512, 512, seed0=6)[0, :, : int(512 * 0.5), :]      # one tile with half tissue
    for dt in (torch.float32, torch.uint8):
        x = synth.as_dtype(tiles, dt).to(dev)
        two, p2, classic, p1 = _both(be, x)
        _check_equal(two, p2, classic, p1, dt)
        assert int((p2["fell_back"] & 15).max()) == 0, (dt, p2["fell_back"])


def test_two_pass_replays_from_a_graph_on_new_data(be, dev):
    """The four launches captured with the two-pass form forced (the router itself falls back to the four-pass form inside a
    capture: it cannot read its telemetry there): a replay on other pixels in the same buffer gives the eager call's bits."""
    a = synth.as_dtype(synth.he_batch(8, 256, 256, seed0=21), torch.float32).to(dev)
    b = synth.as_dtype(synth.he_batch(8, 256, 256, seed0=22), torch.float32).to(dev)
    sm, tmc = SM.to(dev), TMC.to(dev)
    x = a.clone()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):
            be.transform(x, sm, tmc, _extra_flags=_native.MACENKO_TWO_PASS)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        out = be.transform(x, sm, tmc, _extra_flags=_native.MACENKO_TWO_PASS)
    x.copy_(b)
    g.replay()
    torch.cuda.synchronize()
    replayed = out.clone()
    eager = be.transform(b, sm, tmc, _extra_flags=_native.MACENKO_CLASSIC)
    assert _same(replayed, eager)
