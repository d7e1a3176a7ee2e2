"""The tile-resident Macenko transform (stainx_amd/csrc/macenko_resident.hpp: one launch, a tile's pixels kept on chip as
8-bit codes, exact percentiles by on-chip histogram selection) against the four-pass form of the same library and against the
reference's recorded outputs.

The two forms accumulate the moments in the same order and select exact order statistics of the same keys, so they agree BIT
FOR BIT wherever their work items are the same (every float32 / uint8 shape; 16-bit tiles except where the four-pass form
adds a work item: there the covariance differs in its last bits and the outputs are held to one unit in the last place).
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
CODES = {torch.uint8: 0, torch.float16: 1, torch.bfloat16: 2, torch.float32: 3}


def _both(be, x, **kw):
    n, _, h, w = x.shape
    assert be._lib.sx_macenko_form(CODES[x.dtype], n, h, w, _native.MACENKO_RESIDENT | (_native.MACENKO_NORMALIZE_0_1 if kw.get("normalize_to_0_1") else 0)) == 3, "the resident form should serve this call"
    res = be.transform(x, SM, TMC, _extra_flags=_native.MACENKO_RESIDENT, **kw)
    pr = be.tile_params(x.shape[0])
    classic = be.transform(x, SM, TMC, _extra_flags=_native.MACENKO_CLASSIC, **kw)
    pc = be.tile_params(x.shape[0])
    return res, pr, classic, pc


def _same(a: torch.Tensor, b: torch.Tensor) -> bool:
    return torch.equal(a.cpu().view(torch.uint8), b.cpu().view(torch.uint8))


def _check_equal(res, pr, classic, pc, what):
    # (bits 0..3: the slots whose bracket missed -- the general path, same results; bits 8..: an error inside the kernel)
    assert int((pr["fell_back"] >> 8).max()) == 0, (what, "the resident kernel reported an error", pr["fell_back"])
    for k in ("n_kept", "use_all", "cov", "vecs", "phi_lo", "phi_hi", "he", "max_c"):
        assert torch.equal(pr[k], pc[k]), (what, k, (pr[k].double() - pc[k].double()).abs().max())
    assert _same(res, classic), (what, (res.double() - classic.double()).abs().max().item())


@pytest.mark.parametrize("dt", [torch.float32, torch.uint8, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("hw,n", [((512, 512), 6), ((256, 256), 5), ((128, 128), 3), ((64, 64), 3), ((384, 640), 2)])
def test_resident_equals_four_pass_bitwise(be, dev, dt, hw, n):
    x = synth.as_dtype(synth.he_batch(n, *hw, seed0=4000 + hw[0]), dt).to(dev)
    _check_equal(*_both(be, x), (dt, hw))


@pytest.mark.parametrize("dt", [torch.float32, torch.uint8, torch.bfloat16])
def test_resident_unit_scale(be, dev, dt):
    """normalize_to_0_1 fused (uint8 in: float32 out through the LDS-staged stores)."""
    x = synth.as_dtype(synth.he_batch(4, 256, 256, seed0=77), dt).to(dev)
    _check_equal(*_both(be, x, normalize_to_0_1=True), (dt, "unit"))


def test_resident_config5_shape(be, dev):
    """256 x 224 x 224 bf16 with /255 (BASELINE configs[4]): one workgroup per tile.  The four-pass form splits such a tile into
    five work items, the resident form into four: the fp64 partial sums are grouped differently, so the scalars agree to
    rounding, not to the bit."""
    x = synth.as_dtype(synth.he_batch(256, 224, 224, seed0=900), torch.bfloat16).to(dev)
    res, pr, classic, pc = _both(be, x, normalize_to_0_1=True)
    assert int(pr["fell_back"].max()) == 0      # (no bracket missed on ordinary tiles)
    assert torch.equal(pr["n_kept"], pc["n_kept"])
    assert float((pr["he"] - pc["he"]).abs().max()) < 5e-6 and float(((pr["max_c"] - pc["max_c"]) / pc["max_c"]).abs().max()) < 5e-6
    assert float((res.float() - classic.float()).abs().max()) <= 1.0 / 128      # one bf16 step at 1.0


def test_resident_against_the_reference_golden(be, dev, golden):
    """The reference's recorded run (tests/golden/g1: stainx 0.1.4, backend="torch", CPU) on the 128 x 128 tiles."""
    g = golden("g1_macenko_128x128.npz")
    src = torch.from_numpy(g["src_u8"])
    sm, tmc = torch.from_numpy(g["stain_matrix"]), torch.from_numpy(g["target_max_conc"])
    x = synth.as_dtype(src, torch.float32).to(dev)
    out = be.transform(x, sm, tmc, _extra_flags=_native.MACENKO_RESIDENT)
    p = be.tile_params(x.shape[0])
    assert int(p["fell_back"].max()) == 0
    assert float((out.cpu() - torch.from_numpy(g["out_f32"])).abs().max()) <= 2.55e-2
    xu = src.to(dev)
    outu = be.transform(xu, sm, tmc, _extra_flags=_native.MACENKO_RESIDENT)
    assert int((outu.cpu().int() - torch.from_numpy(g["out_u8"]).int()).abs().max()) <= 1


def test_tiles_that_are_not_8_bit_levels_stream_from_memory(be, dev):
    """Float tiles that are not `u8 / 255` (here: jittered, and scaled beyond 1) cannot be held as codes: their workgroups re-read the
    pixels in every sweep.  Same bits as the four-pass form; mixed with ordinary tiles in one batch."""
    src = synth.he_batch(6, 256, 256, seed0=55)
    x = synth.as_dtype(src, torch.float32)
    g = torch.Generator().manual_seed(8)
    x[1] = (x[1] * (1.0 + 0.01 * torch.rand(x[1].shape, generator=g))).clamp(0, 1)
    x[3] = x[3] * 1.7
    x[4, 0, 17, 33] = 0.123456
    _check_equal(*_both(be, x.to(dev)), "stream f32")
    xb = x.to(torch.bfloat16)
    xb[2] = (xb[2].float() * 0.93).to(torch.bfloat16)
    _check_equal(*_both(be, xb.to(dev)), "stream bf16")


def test_tiles_without_tissue_heavy_ties_and_noise(be, dev):
    """A white tile (fewer than three kept pixels: every pixel is selected), a constant tile, a tile with a small tissue patch,
    few-colour tiles (thousands of equal keys: crowded bins), black borders (a tie group at the 99 % end), noise."""
    g = torch.Generator().manual_seed(3)
    noise = (torch.rand(2, 3, 256, 256, generator=g) * 255).round().to(torch.uint8)
    white = torch.full((1, 3, 256, 256), 250, dtype=torch.uint8)
    flat = torch.full((1, 3, 256, 256), 97, dtype=torch.uint8)
    patch = torch.full((1, 3, 256, 256), 248, dtype=torch.uint8)
    patch[:, :, 40:72, 40:72] = synth.he_batch(1, 32, 32, seed0=9)
    few = synth.he_batch(1, 256, 256, seed0=11)
    few = (few // 64) * 64 + 20
    border = synth.he_batch(1, 256, 256, seed0=12)
    border[:, :, :, :40] = 0
    tissue = synth.he_batch(2, 256, 256, seed0=21)
    src = torch.cat([noise, white, flat, patch, few, border, tissue], dim=0)
    for dt in (torch.float32, torch.uint8):
        x = synth.as_dtype(src, dt).to(dev)
        _check_equal(*_both(be, x), (dt, "odd tiles"))


def test_more_tiles_than_fit_the_chip_and_big_tiles(be, dev):
    """70 tiles of 512 x 512 (four workgroups each: two rounds of the persistent launch) and 1024 x 1024 tiles (sixteen each)."""
    x = synth.as_dtype(synth.he_batch(70, 512, 512, seed0=7000), torch.uint8).to(dev)
    _check_equal(*_both(be, x), "70 tiles")
    x = synth.as_dtype(synth.he_batch(3, 1024, 1024, seed0=7100), torch.float32).to(dev)
    _check_equal(*_both(be, x), "1024 x 1024")


def test_resident_is_deterministic_tile_independent_and_replayable(be, dev):
    x = synth.as_dtype(synth.he_batch(9, 512, 512, seed0=300), torch.float32).to(dev)
    a = be.transform(x, SM, TMC, _extra_flags=_native.MACENKO_RESIDENT)
    b = be.transform(x, SM, TMC, _extra_flags=_native.MACENKO_RESIDENT)
    assert _same(a, b)
    alone = be.transform(x[2:3].contiguous(), SM, TMC, _extra_flags=_native.MACENKO_RESIDENT)
    assert _same(alone, a[2:3])


def test_workspace_contents_never_reach_the_result(be, dev):
    x = synth.as_dtype(synth.he_batch(8, 512, 512, seed0=41), torch.float32).to(dev)
    a = be.transform(x, SM, TMC, _extra_flags=_native.MACENKO_RESIDENT)
    be.last_workspace.fill_(0xA5)
    b = be.transform(x, SM, TMC, _extra_flags=_native.MACENKO_RESIDENT)
    assert _same(a, b)


def test_real_tissue_quadrants(be, dev, golden):
    """The reference's own example images (tests/golden/g11): the resident form on the twenty 512 x 512 quadrants, float32 and uint8,
    bit for bit the four-pass form (which tests/test_real_tissue.py holds to the reference's recorded outputs)."""
    from tests.golden.cases import real_quadrants_512

    imgs = golden("g11_real_images.npz")["images_u8"]
    tiles = torch.from_numpy(np.stack([imgs[i, :, y:y + 512, x:x + 512] for i, y, x in real_quadrants_512()]))
    for dt in (torch.float32, torch.uint8):
        _check_equal(*_both(be, synth.as_dtype(tiles, dt).to(dev)), (dt, "real quadrants"))
