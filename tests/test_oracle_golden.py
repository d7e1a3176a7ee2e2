"""Pin the numpy oracle against outputs of the real reference (tests/golden/*.npz).

These run on CPU (`-m "not gpu"`).  Tolerances are on the 0-255 scale unless
stated and sit at the reference's own reproducibility floor (SURVEY.md 8c:
run-to-run 1.3e-4, thread-count 2.7e-4).
"""
from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle import stain_oracle as so
from stainx_amd import synth
from tests.conftest import TORCH_DTYPES, golden_tensor

FLOOR_255 = 2e-3          # fp32 outputs, 0-255 scale


def _oracle_macenko(x: torch.Tensor, sm, tmc, **kw):
    """bf16/fp16 have no numpy dtype: upcast exactly, run in float32, cast back with torch."""
    if x.dtype in (torch.bfloat16, torch.float16):
        out = so.macenko_transform(x.float().numpy(), sm, tmc, **kw)
        if isinstance(out, tuple):
            return torch.from_numpy(out[0]).to(x.dtype), out[1]
        return torch.from_numpy(out).to(x.dtype)
    out = so.macenko_transform(x.numpy(), sm, tmc, **kw)
    if isinstance(out, tuple):
        return torch.from_numpy(out[0]), out[1]
    return torch.from_numpy(out)


@pytest.mark.parametrize("size", ["64x64", "128x128", "321x199"])
def test_macenko_transform_matches_reference(golden, size):
    g = golden(f"g1_macenko_{size}.npz")
    src = torch.from_numpy(g["src_u8"])
    sm, tmc = g["stain_matrix"], g["target_max_conc"]
    for name, dt in TORCH_DTYPES.items():
        if f"out_{name}" not in g:
            continue
        want = golden_tensor(g[f"out_{name}"], name)
        got = _oracle_macenko(synth.as_dtype(src, dt), sm, tmc)
        assert got.dtype == want.dtype == dt
        diff = (got.double() - want.double()).abs().max().item()
        if name == "u8":
            assert diff <= 1, (name, diff)       # truncation boundary
            assert (got != want).float().mean().item() < 1e-3
        elif name in ("bf16", "f16"):
            # one ulp of the storage type at 255 when the fp32 value sits on a rounding boundary
            assert diff <= (1.0 if name == "bf16" else 0.125), (name, diff)
            assert (got != want).float().mean().item() < 1e-3
        else:
            assert diff <= FLOOR_255, (name, diff)


@pytest.mark.parametrize("size", ["64x64", "128x128", "321x199"])
def test_macenko_intermediates_match_reference(golden, size):
    g = golden(f"g1_macenko_{size}.npz")
    src = torch.from_numpy(g["src_u8"])
    _, params = _oracle_macenko(synth.as_dtype(src, torch.float32), g["stain_matrix"], g["target_max_conc"], return_params=True)
    for i, p in enumerate(params):
        assert p["n_kept"] == g["f32_n_kept"][i]
        np.testing.assert_allclose(p["cov"], g["f32_cov"][i], rtol=0, atol=2e-6)
        np.testing.assert_allclose(p["he"], g["f32_he"][i], rtol=0, atol=2e-5)
        np.testing.assert_allclose(p["max_c"], g["f32_max_c"][i], rtol=2e-5, atol=0)
        # eigenvector signs are solver-dependent; the angles mirror with them
        for col in range(2):
            a, b = p["vecs"][:, col], g["f32_vecs"][i][:, col]
            assert min(np.abs(a - b).max(), np.abs(a + b).max()) < 2e-5


def test_macenko_normalize_to_0_1_matches_reference(golden):
    g = golden("g1_macenko_64x64.npz")
    src = torch.from_numpy(g["src_u8"])
    for name, dt in TORCH_DTYPES.items():
        raw = golden_tensor(g[f"out_{name}"], name)
        want = golden_tensor(g[f"out01_{name}"], "f32" if name == "u8" else name)
        if name in ("bf16", "f16"):
            got = raw / 255.0
        else:
            got = torch.from_numpy(so.apply_normalize_to_0_1(raw.numpy()))
        assert got.dtype == want.dtype
        # the two reference runs behind out / out01 differ by the reference's own run-to-run noise
        # (SURVEY.md 8c), so only quantised dtypes can be compared bit for bit
        if name in ("f32", "f64"):
            assert (got - want).abs().max().item() <= 1e-5, name
        else:
            assert (got.double() - want.double()).abs().max().item() <= 2.0 ** -8, name
            assert (got != want).float().mean().item() < 1e-3, name
    # `tensor / 255.0` on CPU is a correctly rounded division, not a multiply by the reciprocal
    x = torch.rand(4096, generator=torch.Generator().manual_seed(1)) * 255
    assert torch.equal(x / 255.0, torch.from_numpy(so.apply_normalize_to_0_1(x.numpy())))


def test_macenko_fit_matches_reference(golden):
    g = golden("g3_macenko_fit.npz")
    for tag in ("single64", "single224", "pooled4x224", "pooled8x128"):
        he, max_c = so.macenko_fit(g[f"{tag}_u8"])
        np.testing.assert_allclose(he, g[f"{tag}_he"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(max_c, g[f"{tag}_max_c"], rtol=2e-5, atol=0)


def test_pooled_fit_at_config4_size_matches_reference(golden):
    """g12: the REAL reference's pooled fit on the 64 x 512 x 512 tiles one rank holds in BASELINE configs[3] (16.7 M pixels pooled).  At this
    size a float32 running mean loses digits (round 3: the oracle's did, and turned the stain plane by degrees); the oracle is held to the
    reference here, not only the other way round."""
    from stainx_amd import synth
    import torch

    g = golden("g12_config4_pooled_fit.npz")
    tiles = synth.he_batch(64, 512, 512)
    for tag, x in (("rank0_u8", tiles.numpy()), ("rank0_f32", synth.as_dtype(tiles, torch.float32).numpy())):
        he, max_c = so.macenko_fit(x)
        np.testing.assert_allclose(he, g[f"{tag}_he"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(max_c, g[f"{tag}_max_c"], rtol=2e-5, atol=0)
    assert int(g["rank0_f32_n_kept"][0]) == 64 * 512 * 512      # (every pixel of the synthetic tiles is tissue)


def test_macenko_edge_cases_match_reference(golden):
    g = golden("g6_edge_cases.npz")
    sm, tmc = g["stain_matrix"], g["target_max_conc"]
    for tag in ("white", "jitter", "flat"):
        out, params = so.macenko_transform(g[f"{tag}_in"], sm, tmc, return_params=True)
        assert params[0]["n_kept"] == g[f"{tag}_n_kept"][0], tag
        want = g[f"{tag}_out"]
        assert out.dtype == want.dtype
        diff = np.abs(out.astype(np.float64) - want.astype(np.float64)).max()
        assert diff <= (1 if out.dtype == np.uint8 else FLOOR_255), (tag, diff)
    # the fallback really is exercised: fewer than 3 pixels pass the filter, all 48*48 are used
    assert g["white_n_kept"][0] == 48 * 48 and g["white2_n_kept"][0] == 48 * 48
    # white2 (near-white noise + two dark pixels) is near-isotropic: its angles wrap around +-pi, so the
    # output depends on LAPACK's arbitrary eigenvector signs.  With the signs of the reference's LAPACK
    # the oracle reproduces the reference; that is the only discrepancy.
    best = None
    for signs in ((1, 1), (1, -1), (-1, 1), (-1, -1)):
        out, params = so.macenko_transform(g["white2_in"], sm, tmc, return_params=True, signs=signs)
        if np.abs(params[0]["vecs"] - g["white2_vecs"][0]).max() < 1e-4:
            best = np.abs(out.astype(int) - g["white2_out"].astype(int)).max()
    assert best is not None and best <= 1


def test_macenko_config2_subsample_matches_reference(golden):
    """Four of the 64 config-2 tiles (oracle speed); the GPU test covers all 64."""
    g = golden("g2_macenko_config2.npz")
    stride = int(g["sub_stride"])
    for i in (0, 21, 42, 63):
        tile = synth.he_tile(512, 512, 1000 + i, 1.0 + 0.005 * i)
        out, params = so.macenko_transform(synth.as_dtype(tile, torch.float32).numpy(), g["stain_matrix"], g["target_max_conc"], return_params=True)
        sub = out.reshape(3, -1)[:, ::stride]
        assert np.abs(sub - g["out_sub"][i]).max() <= FLOOR_255
        assert params[0]["n_kept"] == g["n_kept"][i]
        np.testing.assert_allclose(params[0]["he"], g["he"][i], atol=2e-5)


def test_reinhard_matches_reference(golden):
    g = golden("g4_reinhard.npz")
    for tag, hw, n in (("cfg1_512", (512, 512), 1), ("b2_128", (128, 128), 2), ("odd_67x45", (67, 45), 3)):
        ref = synth.noise_u8((1, 3, *hw), 42)
        src = synth.noise_u8((n, 3, *hw), 43)
        for name in ("f32", "u8", "bf16"):
            dt = TORCH_DTYPES[name]
            rin, sin = synth.as_dtype(ref, dt), synth.as_dtype(src, dt)
            as_np = (lambda t: t.float().numpy()) if name == "bf16" else (lambda t: t.numpy())
            mean, std = so.reinhard_fit(as_np(rin))
            np.testing.assert_allclose(mean, golden_tensor(g[f"{tag}_{name}_ref_mean"], "f32").numpy(), rtol=2e-6, atol=2e-5)
            np.testing.assert_allclose(std, golden_tensor(g[f"{tag}_{name}_ref_std"], "f32").numpy(), rtol=2e-5, atol=2e-5)
            out = so.reinhard_transform(as_np(sin), mean, std)
            got = torch.from_numpy(out).to(dt)
            if f"{tag}_{name}_out" in g:
                want = golden_tensor(g[f"{tag}_{name}_out"], name)
            else:
                want = golden_tensor(g[f"{tag}_{name}_out_sub"], name)
                got = got.reshape(n, 3, -1)[:, :, ::61]
            diff = (got.double() - want.double()).abs().max().item()
            tol = {"f32": 2e-5, "u8": 1, "bf16": 2 ** -8}[name]
            assert diff <= tol, (tag, name, diff)
    he_out = so.reinhard_transform(synth.he_batch(2, 96, 96, seed0=500, scale_step=0.1).numpy(), g["he96_ref_mean"], g["he96_ref_std"])
    assert np.abs(he_out.astype(int) - g["he96_out_u8"].astype(int)).max() <= 1


def test_histogram_matching_matches_reference(golden):
    g = golden("g5_histogram_matching.npz")
    for tag in ("noise", "he"):
        ref8, src8 = torch.from_numpy(g[f"{tag}_ref_u8"]), torch.from_numpy(g[f"{tag}_src_u8"])
        for name in ("u8", "f32", "bf16"):
            dt = TORCH_DTYPES[name]
            for layout, axis in (("nchw", 1), ("nhwc", -1)):
                rin, sin = synth.as_dtype(ref8, dt), synth.as_dtype(src8, dt)
                if axis == -1:
                    rin, sin = rin.permute(0, 2, 3, 1).contiguous(), sin.permute(0, 2, 3, 1).contiguous()
                as_np = (lambda t: t.float().numpy()) if name == "bf16" else (lambda t: t.numpy())
                hists = so.hm_fit(as_np(rin), axis)
                key = f"{tag}_{name}_{layout}"
                np.testing.assert_array_equal(np.stack(hists), g[f"{key}_ref_hists"])       # bit-exact
                out, tables = so.hm_transform(as_np(sin), hists, axis, return_tables=True)
                got = torch.from_numpy(np.ascontiguousarray(out)).to(dt)
                want = golden_tensor(g[f"{key}_out"], name)
                assert torch.equal(got, want), key                                          # bit-exact
                if name == "u8" and axis == 1:
                    np.testing.assert_array_equal(np.stack(tables["counts"]), g[f"{tag}_counts"])
                    lut_t = np.trunc(np.stack(tables["lut"]))
                    present = g[f"{tag}_lut_u8_trunc"] >= 0
                    np.testing.assert_array_equal(lut_t[present], g[f"{tag}_lut_u8_trunc"][present])


def test_color_round_trip():
    """Reference tests/test_torch_backend_color_space.py:12-35."""
    rng = np.random.default_rng(0)
    rgb = rng.random((1, 3, 8, 8), dtype=np.float32)
    back = so.lab_to_rgb(so.rgb_to_lab(rgb))
    assert np.abs(back - rgb).max() < 3e-2
    lab = so.rgb_to_lab(np.full((1, 3, 4, 4), 1.2, dtype=np.float32))
    assert lab[:, 0].min() > 50.0


def test_histogram_matching_random_cases_bit_exact(golden):
    """g9: 80 random small cases.  The LUT hinges on the last bit of `ref_hist.sum()`, which torch adds up in its
    vectorised order (oracle._torch_sum_f32); every output must equal the reference's, bit for bit."""
    from tests.golden.cases import g9_cases

    g = golden("g9_hm_random.npz")
    for i, (n, h, w, name, s_src, s_ref) in enumerate(g9_cases()):
        x = synth.as_dtype(synth.noise_u8((n, 3, h, w), s_src), TORCH_DTYPES[name]).numpy()
        ref = synth.as_dtype(synth.noise_u8((1, 3, h, w), s_ref), TORCH_DTYPES[name]).numpy()
        got = so.hm_transform(x, so.hm_fit(ref))
        want = g[f"c{i}_out"]
        assert got.dtype == want.dtype and np.array_equal(got, want), (i, n, h, w, name)
