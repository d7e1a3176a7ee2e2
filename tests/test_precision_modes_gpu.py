"""``Macenko(precision=...)``: "stable" and "fast" (the reference's two values) and the extension "sampled".

The reference's fast mode keeps EXACT nearest-rank percentiles and moves its big tensors to float16 (restated in
``oracle.macenko_transform_fast`` from src/stainx_cuda_torch/csrc/macenko.cu:116-191); its published accuracy is MAE ~0.05 grey
levels.  Here "fast" runs the exact kernels: its result must lie inside that bound of the restated fp16 path and at the stable
tolerance of the float32 oracle.  "sampled" is an approximation and is only held to its documented error."""
from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle import stain_oracle as so
from stainx_amd import _native, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def test_precision_fast_within_the_reference_fast_modes_bound(dev):
    from stainx_amd import Macenko

    ref = synth.reference_tile(128, 128)
    tiles = synth.he_batch(6, 192, 160, seed0=5100, scale_step=0.03)
    he, mc = so.macenko_fit(ref.numpy())
    for dt in (torch.float32, torch.uint8):
        x = synth.as_dtype(tiles, dt)
        fast = Macenko(device=dev, precision="fast").fit(ref.to(dev))
        stable = Macenko(device=dev, precision="stable").fit(ref.to(dev))
        out = fast.transform(x.to(dev))
        assert torch.equal(out, stable.transform(x.to(dev)))                          # the same kernels
        want_fast = so.macenko_transform_fast(x.numpy(), he, mc, signs="positive_sum").astype(np.float64)
        want_stable = so.macenko_transform(x.numpy(), he, mc).astype(np.float64)
        got = out.cpu().numpy().astype(np.float64)
        # the published accuracy of the reference's fast mode is MAE ~0.05 grey levels against the float32 result: a precision="fast"
        # result has to be at least that close to the float32 path (uint8: + the truncation to grey levels) ...
        assert float(np.abs(got - want_stable).mean()) <= 0.05 + (0.5 if dt == torch.uint8 else 0.0), dt
        # ... and about as far from the restated fp16 path as that path is from float32 (measured: 0.09; numpy's emulation of the
        # fp16 products rounds a little differently from cuBLAS): the distance is the mode's own error, not this library's
        mae_modes = float(np.abs(want_fast - want_stable).mean())
        assert 0.01 <= mae_modes <= 0.2, mae_modes
        assert float(np.abs(got - want_fast).mean()) <= mae_modes + 0.05 + (0.5 if dt == torch.uint8 else 0.0), dt
        tol = 1.0 if dt == torch.uint8 else 2.55e-2
        assert float(np.abs(got - want_stable).max()) <= tol, dt


def test_precision_sampled_is_an_opt_in_approximation(dev):
    """4096-pixel sample percentiles (moments pass + one stage + reconstruct): close to the exact transform statistically (measured
    mean 0.5 / worst tile 1.5 grey levels on 64 tiles of 512x512), deterministic, and never chosen unless asked for."""
    from stainx_amd import Macenko

    ref = synth.reference_tile(128, 128).to(dev)
    x = synth.as_dtype(synth.he_batch(6, 256, 256, seed0=5100), torch.float32).to(dev)
    exact = Macenko(device=dev).fit(ref).transform(x)
    sampled_norm = Macenko(device=dev, precision="sampled").fit(ref)
    sampled = sampled_norm.transform(x)
    assert sampled.shape == exact.shape and sampled.dtype == exact.dtype
    err = (sampled - exact).abs().reshape(6, -1)
    assert float(err.mean()) < 1.0 and float(err.mean(1).max()) < 2.5 and float(err.max()) < 15.0, err.mean(1)
    assert torch.equal(sampled, sampled_norm.transform(x))
    assert not torch.equal(sampled, exact)
    with pytest.raises(ValueError, match="precision must be"):
        Macenko(device=dev, precision="quick")
    for dt in (torch.uint8, torch.bfloat16):
        xi = synth.as_dtype(synth.he_batch(2, 64, 96, seed0=5200), dt).to(dev)
        out = sampled_norm.transform(xi)
        assert out.dtype == xi.dtype and out.shape == xi.shape


@pytest.mark.parametrize("hw", [(300, 300), (127, 129), (257, 257)])
def test_workspace_contents_never_reach_the_result(dev, hw):
    """The header's contract: the workspace needs no initialisation.  Tile sizes whose last sample cell is partial used to leave
    one sample unwritten: poison the workspace and compare with a clean run, in every form of the transform."""
    from stainx_amd.backends.torch_hip_backend import MacenkoHIP

    x = synth.as_dtype(synth.he_batch(3, *hw, seed0=31), torch.float32).to(dev)
    sm, tmc = torch.tensor(synth.HE_REF).to(dev), torch.tensor([1.9705, 1.0308]).to(dev)
    for precision, flags in (("stable", _native.MACENKO_CLASSIC), ("stable", _native.MACENKO_TWO_PASS), ("sampled", 0)):
        be = MacenkoHIP(dev, precision=precision, diag=bool(flags & _native.MACENKO_DIAG_BITS))
        first = be.transform(x, sm, tmc, _extra_flags=flags)
        ws = be.last_workspace
        for fill in (0xFF, 0x00, 0x7F):
            torch.cuda.synchronize()
            ws.fill_(fill)
            again = be.transform(x, sm, tmc, _extra_flags=flags)
            assert be.last_workspace is ws and torch.equal(again, first), (precision, flags, fill)
