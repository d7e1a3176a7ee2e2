"""GPU parity tests of the Reinhard and histogram-matching paths against the
committed reference outputs and the CPU oracle."""
from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle import stain_oracle as so
from stainx_amd import synth
from tests.conftest import TORCH_DTYPES, golden_tensor

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


# ------------------------------------------------------------------ Reinhard
def test_reinhard_matches_reference_golden(dev, golden):
    from stainx_amd.backends.torch_hip_backend import ReinhardHIP

    g = golden("g4_reinhard.npz")
    be = ReinhardHIP(dev)
    for tag, hw, n in (("cfg1_512", (512, 512), 1), ("b2_128", (128, 128), 2), ("odd_67x45", (67, 45), 3)):
        ref = synth.noise_u8((1, 3, *hw), 42)
        src = synth.noise_u8((n, 3, *hw), 43)
        for name in ("f32", "u8", "bf16"):
            dt = TORCH_DTYPES[name]
            rin, sin = synth.as_dtype(ref, dt).to(dev), synth.as_dtype(src, dt).to(dev)
            mean, std = be.compute_reference_mean_std(rin)
            np.testing.assert_allclose(mean.cpu().numpy(), g[f"{tag}_{name}_ref_mean"], rtol=0, atol=2e-3)   # LAB units (0..255)
            np.testing.assert_allclose(std.cpu().numpy(), g[f"{tag}_{name}_ref_std"], rtol=1e-4, atol=1e-3)
            out = be.transform(sin, torch.from_numpy(g[f"{tag}_{name}_ref_mean"]), torch.from_numpy(g[f"{tag}_{name}_ref_std"]))
            assert out.dtype == dt and out.shape == sin.shape
            got = out.cpu()
            if f"{tag}_{name}_out" in g:
                want = golden_tensor(g[f"{tag}_{name}_out"], name)
            else:
                want = golden_tensor(g[f"{tag}_{name}_out_sub"], name)
                got = got.reshape(n, 3, -1)[:, :, ::61]
            diff = (got.double() - want.double()).abs()
            if name == "f32":
                assert diff.max().item() <= 1e-4, (tag, diff.max().item())        # [0,1] scale
            elif name == "u8":
                assert diff.max().item() <= 1 and (diff > 0).float().mean().item() < 5e-3, tag
            else:
                assert diff.max().item() <= 2.0 ** -8 and (diff > 0).float().mean().item() < 2e-2, tag


def test_reinhard_config1_fit_transform_vs_oracle(dev):
    """BASELINE configs[0]: Reinhard fit + transform, 1x3x512x512 fp32."""
    from stainx_amd import Reinhard

    ref = synth.as_dtype(synth.noise_u8((1, 3, 512, 512), 42), torch.float32)
    src = synth.as_dtype(synth.noise_u8((1, 3, 512, 512), 43), torch.float32)
    norm = Reinhard(device=dev, backend="torch_hip")
    out = norm.fit(ref.to(dev)).transform(src.to(dev))
    mean, std = so.reinhard_fit(ref.numpy())
    want = so.reinhard_transform(src.numpy(), mean, std)
    np.testing.assert_allclose(norm._reference_mean.cpu().numpy(), mean, atol=2e-3)
    assert np.abs(out.cpu().numpy() - want).max() <= 1e-4
    # Beer-Lambert tiles, uint8, batch statistics are pooled over the batch (not per tile)
    tiles = synth.he_batch(3, 96, 96, seed0=500, scale_step=0.1)
    got = norm.fit(synth.reference_tile(96, 96).to(dev)).transform(tiles.to(dev)).cpu().numpy()
    m2, s2 = so.reinhard_fit(synth.reference_tile(96, 96).numpy())
    want = so.reinhard_transform(tiles.numpy(), m2, s2)
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1
    solo = norm.transform(tiles[:1].to(dev)).cpu().numpy()
    assert np.abs(solo.astype(int) - got[:1].astype(int)).max() > 1      # batch-coupled, like the reference


@pytest.mark.parametrize("shape", [(3, 30, 30), (2, 33, 47), (1, 17, 20), (4, 50, 50), (1, 5, 4)])
def test_reinhard_partial_last_sweep(dev, shape):
    """the colour matrices run on the matrix core (v_mfma_f32_4x4x1 ignores the execution mask and reads its matrix column from
    all four lanes of a block), so a sweep in which some lanes of a block have no pixels left must still give every live lane its
    result: tiles whose pixel count is not a multiple of a wave's 256 (16-byte packs: 900, 2500 pixels) or of four (1551, 340: the
    scalar path), fit and transform against the oracle."""
    from stainx_amd import Reinhard

    n, h, w = shape
    ref = synth.as_dtype(synth.noise_u8((1, 3, h, w), 142), torch.float32)
    src_u8 = synth.noise_u8((n, 3, h, w), 143)
    mean, std = so.reinhard_fit(ref.numpy())
    for dt, tol in ((torch.float32, 1e-4), (torch.uint8, 1)):
        src = synth.as_dtype(src_u8, dt)
        norm = Reinhard(device=dev, backend="torch_hip").fit(ref.to(dev))
        np.testing.assert_allclose(norm._reference_mean.cpu().numpy(), mean, atol=2e-3)
        got = norm.transform(src.to(dev)).cpu().numpy()
        want = so.reinhard_transform(src.numpy(), mean, std)
        assert np.abs(got.astype(np.float64) - want.astype(np.float64)).max() <= tol, (shape, dt)


# ------------------------------------------------------------------ histogram matching
def test_histogram_matching_matches_reference_golden(dev, golden):
    from stainx_amd.backends.torch_hip_backend import HistogramMatchingHIP

    g = golden("g5_histogram_matching.npz")
    for tag in ("noise", "he"):
        ref8, src8 = torch.from_numpy(g[f"{tag}_ref_u8"]), torch.from_numpy(g[f"{tag}_src_u8"])
        for name in ("u8", "f32", "bf16"):
            dt = TORCH_DTYPES[name]
            for layout, axis in (("nchw", 1), ("nhwc", -1)):
                be = HistogramMatchingHIP(dev, channel_axis=axis)
                rin, sin = synth.as_dtype(ref8, dt), synth.as_dtype(src8, dt)
                if axis == -1:
                    rin, sin = rin.permute(0, 2, 3, 1).contiguous(), sin.permute(0, 2, 3, 1).contiguous()
                key = f"{tag}_{name}_{layout}"
                hists = be.compute_reference_histograms(rin.to(dev))
                np.testing.assert_array_equal(torch.stack(hists).cpu().numpy(), g[f"{key}_ref_hists"])       # bit-exact
                out = be.transform(sin.to(dev), hists)
                want = golden_tensor(g[f"{key}_out"], name)
                assert out.dtype == dt and out.shape == sin.shape
                assert torch.equal(out.cpu(), want), key                                                      # bit-exact
                if name == "u8" and axis == 1:
                    tab = be.tables()
                    np.testing.assert_array_equal(tab["counts"].numpy(), g[f"{tag}_counts"])                  # integer histogram
                    present = g[f"{tag}_lut_u8_trunc"] >= 0
                    np.testing.assert_array_equal(np.trunc(tab["lut"].numpy())[present], g[f"{tag}_lut_u8_trunc"][present])


def test_histogram_matching_config3_properties(dev):
    """BASELINE configs[2] shape family (uint8, 1024x1024 tiles; 8 tiles here, 64 in the bench): the pooled
    histogram is an exact integer count, the LUT is monotone, and the result equals the oracle's."""
    from stainx_amd import HistogramMatching

    ref = synth.noise_u8((1, 3, 1024, 1024), 42)
    src = synth.noise_u8((8, 3, 1024, 1024), 43)
    hm = HistogramMatching(device=dev, backend="torch_hip")
    out = hm.fit(ref.to(dev)).transform(src.to(dev))
    tab = hm._get_backend_impl().tables()
    counts = np.stack([np.bincount(src[:, c].reshape(-1).numpy(), minlength=256) for c in range(3)])
    np.testing.assert_array_equal(tab["counts"].numpy(), counts)
    assert int(tab["counts"].sum()) == src.numel()
    assert bool((tab["lut"][:, 1:] >= tab["lut"][:, :-1]).all())
    want = so.hm_transform(src.numpy(), so.hm_fit(ref.numpy()))
    assert np.array_equal(out.cpu().numpy(), want)
    # a single (256,) reference histogram applies to every channel (torch_backend.py:224-226)
    one = hm._ref_histograms_256[0]
    got1 = hm._get_backend_impl().transform(src[:1].to(dev), one)
    want1 = so.hm_transform(src[:1].numpy(), one.cpu().numpy())
    assert np.array_equal(got1.cpu().numpy(), want1)


def test_histogram_matching_errors(dev):
    from stainx_amd.backends.torch_hip_backend import HistogramMatchingHIP

    be = HistogramMatchingHIP(dev)
    x = synth.noise_u8((1, 3, 16, 16), 1)
    with pytest.raises(ValueError, match="cannot be empty"):
        be.transform(x, [])
    with pytest.raises(ValueError, match="256 elements"):
        be.transform(x, torch.rand(128))
    with pytest.raises(TypeError, match="must be a torch.Tensor"):
        be.transform(x, [np.zeros(256)])


def test_histogram_matching_random_cases_bit_exact(golden):
    """g9: 80 random small cases from the real reference (odd sizes, four dtypes).  The output hinges on the last bit of
    the float32 sums of the 256-bin histograms; the kernels add them in ATen's order (torch_sum_256)."""
    from stainx_amd.backends.torch_hip_backend import HistogramMatchingHIP
    from tests.golden.cases import g9_cases

    dev = torch.device("cuda:0")
    g = golden("g9_hm_random.npz")
    hb = HistogramMatchingHIP(dev, channel_axis=1)
    for i, (n, h, w, name, s_src, s_ref) in enumerate(g9_cases()):
        x = synth.as_dtype(synth.noise_u8((n, 3, h, w), s_src), TORCH_DTYPES[name])
        ref = synth.as_dtype(synth.noise_u8((1, 3, h, w), s_ref), TORCH_DTYPES[name])
        got = hb.transform(x.to(dev), hb.compute_reference_histograms(ref.to(dev))).cpu().numpy()
        want = g[f"c{i}_out"]
        assert got.dtype == want.dtype and np.array_equal(got, want), (i, n, h, w, name)


def test_hm_fit_and_transform_under_inference_mode(dev):
    """the stacked-reference cache read `_version`, which inference tensors do not have -- every transform after a fit
    under torch.inference_mode() raised.  Fit and transform inside inference mode, twice, and against the call outside it."""
    from stainx_amd import HistogramMatching

    ref = synth.noise_u8((1, 3, 96, 80), 42).to(dev)
    src = synth.noise_u8((3, 3, 96, 80), 43).to(dev)
    want = HistogramMatching(device=dev, backend="torch_hip").fit(ref).transform(src)
    with torch.inference_mode():
        hm = HistogramMatching(device=dev, backend="torch_hip").fit(ref)
        first = hm.transform(src)
        second = hm.transform(src)
    assert torch.equal(first.cpu(), want.cpu()) and torch.equal(second.cpu(), want.cpu())


def test_histogram_matching_ready_workspace_calls(dev):
    """include/stainx_hip.h, sx_hm_*_ready: the calls without the clearing launch.  Through the C ABI: the plain call accepts a
    workspace full of garbage and leaves it ready; the ready calls give the same bits on a ready workspace, one after the other
    (each leaves it ready: transform, fit, counts); a workspace that was NOT ready is noticed and reported."""
    from stainx_amd import _native
    from stainx_amd.backends.torch_hip_backend import HistogramMatchingHIP

    lib = _native.require()
    be = HistogramMatchingHIP(dev)
    src = synth.noise_u8((3, 3, 200, 328), 7).to(dev)
    ref = torch.stack(be.compute_reference_histograms(synth.noise_u8((1, 3, 200, 328), 8).to(dev))).contiguous()
    want = torch.from_numpy(so.hm_transform(src.cpu().numpy(), [r.cpu().numpy() for r in ref])).to(dev)
    n, _, h, w = src.shape
    u8 = _native.DTYPE_CODES[torch.uint8]
    stream = _native.stream_ptr(dev)
    ws = torch.full((int(lib.sx_hm_workspace_bytes(n, h, w)),), 0xA5, dtype=torch.uint8, device=dev)
    off = int(lib.sx_hm_workspace_status_offset())

    def status():
        return int(ws[off:off + 4].view(torch.int32).item())

    def run(fn):
        out = torch.empty_like(src)
        assert fn(src.data_ptr(), out.data_ptr(), u8, n, h, w, 0, ref.data_ptr(), ws.data_ptr(), ws.numel(), stream) == 0
        return out

    assert torch.equal(run(lib.sx_hm_transform), want) and status() == 0          # garbage in: the plain call clears what it needs
    for _ in range(3):
        assert torch.equal(run(lib.sx_hm_transform_ready), want) and status() == 0
    hists = torch.empty((3, 256), dtype=torch.float32, device=dev)
    assert lib.sx_hm_fit_ready(src.data_ptr(), u8, n, h, w, 0, hists.data_ptr(), ws.data_ptr(), ws.numel(), stream) == 0
    counts = torch.empty((3, 256), dtype=torch.int64, device=dev)
    assert lib.sx_hm_counts_ready(src.data_ptr(), u8, n, h, w, 0, counts.data_ptr(), ws.data_ptr(), ws.numel(), stream) == 0
    assert int(counts.sum()) == src.numel()
    assert torch.equal(hists, torch.stack(be.compute_reference_histograms(src)))
    assert torch.equal(run(lib.sx_hm_transform_ready), want) and status() == 0
    ws.fill_(0x01)                                                                   # somebody else wrote here
    run(lib.sx_hm_transform_ready)
    assert status() & 1
    assert lib.sx_hm_workspace_init(ws.data_ptr(), ws.numel(), stream) == 0
    assert torch.equal(run(lib.sx_hm_transform_ready), want) and status() == 0
    # the backend's own workspaces are zero-filled when they are made, and it only uses the ready calls
    for _ in range(3):
        assert torch.equal(be.transform(src, list(ref)), want)
    assert be.workspace_status() == 0
    assert int(be.tables()["counts"].sum()) == src.numel()


def test_reinhard_ready_workspace_calls(dev):
    """include/stainx_hip.h, sx_reinhard_transform_ready: the transform without the launch that clears the arrival counters.  Same bits
    as the plain call; a workspace that was not ready is noticed; calls of different shapes may alternate on one ready workspace (the
    counters lie at a place the shape does not move)."""
    from stainx_amd import _native
    from stainx_amd.backends.torch_hip_backend import ReinhardHIP

    lib = _native.require()
    be = ReinhardHIP(dev)
    src = synth.as_dtype(synth.noise_u8((5, 3, 200, 328), 7), torch.float32).to(dev)
    mean = torch.tensor([150.0, 130.0, 120.0], device=dev)
    std = torch.tensor([40.0, 9.0, 12.0], device=dev)
    n, _, h, w = src.shape
    f32 = _native.DTYPE_CODES[torch.float32]
    stream = _native.stream_ptr(dev)
    ws = torch.full((int(lib.sx_reinhard_workspace_bytes(n, h, w)),), 0xA5, dtype=torch.uint8, device=dev)
    off = int(lib.sx_reinhard_workspace_status_offset())

    def status():
        return int(ws[off:off + 4].view(torch.int32).item())

    def run(fn):
        out = torch.empty_like(src)
        assert fn(src.data_ptr(), out.data_ptr(), f32, n, h, w, mean.data_ptr(), std.data_ptr(), ws.data_ptr(), ws.numel(), stream) == 0
        return out

    want = run(lib.sx_reinhard_transform)                    # garbage in: the plain call clears what it needs
    oracle = so.reinhard_transform(src.cpu().numpy(), mean.cpu().numpy(), std.cpu().numpy())
    assert np.abs(want.cpu().numpy() - oracle).max() <= 1e-4
    for _ in range(3):
        assert torch.equal(run(lib.sx_reinhard_transform_ready), want) and status() == 0
    ws.fill_(0x01)                                            # somebody else wrote here
    run(lib.sx_reinhard_transform_ready)
    assert status() & 1
    assert lib.sx_reinhard_workspace_init(ws.data_ptr(), ws.numel(), stream) == 0
    assert torch.equal(run(lib.sx_reinhard_transform_ready), want) and status() == 0
    # shapes alternate on ONE ready workspace through the C ABI, no init in between (the arrival counters used to lie behind the
    # partial sums, i.e. where the SHAPE put them -- a call of one shape left its sums where the other shape's counters are)
    shapes = [src, src[:2, :, :64, :96].contiguous(), src[:5, :, :200, :200].contiguous(), src[:1].contiguous()]

    def run_on(x, fn):
        out = torch.empty_like(x)
        assert fn(x.data_ptr(), out.data_ptr(), f32, x.shape[0], x.shape[2], x.shape[3], mean.data_ptr(), std.data_ptr(), ws.data_ptr(), ws.numel(), stream) == 0
        return out

    wants = [run_on(x, lib.sx_reinhard_transform) for x in shapes]
    for _ in range(2):
        for x, w_ in zip(shapes + shapes[::-1], wants + wants[::-1]):
            assert torch.equal(run_on(x, lib.sx_reinhard_transform_ready), w_) and status() == 0, tuple(x.shape)
    # the backend: shapes alternate on one workspace, a fit in between
    small = src[:2, :, :64, :96].contiguous()
    want_small = be.transform(small, mean, std)
    for _ in range(2):
        assert torch.equal(be.transform(src, mean, std), want) and be.workspace_status() == 0
        be.compute_reference_mean_std(small)
        assert torch.equal(be.transform(small, mean, std), want_small) and be.workspace_status() == 0
        assert torch.equal(be.transform(src, mean, std), want)
