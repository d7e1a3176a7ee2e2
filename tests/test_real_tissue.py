"""Real tissue: the six 1024 x 1024 H&E tiles the reference ships for its own example (examples/data/*.png; pixel arrays in
tests/golden/g11_real_images.npz) and what the REAL reference makes of them (tests/golden/g11_real_tissue.npz, written by
tests/golden/make_golden.py g11).  Every other golden is an i.i.d. Beer-Lambert tile without spatial structure; these have
clustered tails, background, JPEG-like quantisation.

CPU part (`-m "not gpu"`): the numpy oracle is held to the reference's outputs on these tiles.
GPU part (`-m gpu`): the library -- in ALL its forms of the Macenko transform -- against the same files.
"""
from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle import stain_oracle as so
from stainx_amd import synth
from tests.conftest import golden_tensor
from tests.golden.cases import real_crops_224, real_quadrants_512

FLOOR_255 = 2e-3
TOL_255, TOL_UNIT = 2.55e-2, 1e-4


def thin(t: torch.Tensor) -> torch.Tensor:
    """The fixture keeps every 5th pixel of a float32 result (make_golden.py g11)."""
    return t.reshape(t.shape[0], 3, -1)[:, :, ::5].contiguous()


@pytest.fixture(scope="module")
def real(golden):
    imgs = torch.from_numpy(golden("g11_real_images.npz")["images_u8"])
    g = golden("g11_real_tissue.npz")
    quads = torch.stack([imgs[i, :, y:y + 512, x:x + 512] for i, y, x in real_quadrants_512()]).contiguous()
    crops = torch.stack([imgs[i, :, y:y + 224, x:x + 224] for i, y, x in real_crops_224()]).contiguous()
    return imgs, g, quads, crops


# ------------------------------------------------------------------------------------------------ CPU: the oracle
def test_oracle_fit_on_the_target_image(real):
    imgs, g, _, _ = real
    he, max_c = so.macenko_fit(imgs[0:1].numpy(), signs="positive_sum")
    np.testing.assert_allclose(he, g["stain_matrix"], atol=2e-5)
    np.testing.assert_allclose(max_c, g["target_max_conc"], rtol=2e-5)


def test_oracle_transform_on_real_crops(real):
    _, g, _, crops = real
    sm, tmc = g["stain_matrix"], g["target_max_conc"]
    out, params = so.macenko_transform(synth.as_dtype(crops, torch.float32).numpy(), sm, tmc, return_params=True)
    assert np.abs(thin(torch.from_numpy(out)).numpy().astype(np.float64) - g["c224_f32_out"]).max() <= FLOOR_255
    for i, p in enumerate(params):
        assert p["n_kept"] == g["c224_n_kept"][i]
        np.testing.assert_allclose(p["cov"], g["c224_cov"][i], rtol=0, atol=2e-6)
        np.testing.assert_allclose(p["he"], g["c224_he"][i], rtol=0, atol=2e-5)
        np.testing.assert_allclose(p["max_c"], g["c224_max_c"][i], rtol=2e-5, atol=0)
    out8 = so.macenko_transform(crops.numpy(), sm, tmc)
    d = np.abs(out8.astype(int) - g["c224_u8_out"].astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3
    he, max_c = so.macenko_fit(crops.numpy(), signs="positive_sum")
    np.testing.assert_allclose(he, g["c224_pooled_he"], atol=2e-5)
    np.testing.assert_allclose(max_c, g["c224_pooled_max_c"], rtol=2e-5)


def test_oracle_siblings_on_real_crops(real):
    imgs, g, _, crops = real
    ref512 = imgs[0:1, :, :512, :512].contiguous().numpy()
    mean, std = so.reinhard_fit(ref512)
    np.testing.assert_allclose(mean, g["c224_reinhard_ref_mean"], rtol=1e-5)
    np.testing.assert_allclose(std, g["c224_reinhard_ref_std"], rtol=1e-5)
    out = so.reinhard_transform(crops.numpy(), mean, std)
    assert np.abs(out.astype(int) - g["c224_reinhard_u8"].astype(int)).max() <= 1
    np.testing.assert_array_equal(so.hm_transform(crops.numpy(), so.hm_fit(ref512)), g["c224_hm_u8"])      # integer work: bit-exact


# ------------------------------------------------------------------------------------------------ GPU: the library
@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _forms():
    from stainx_amd import _native

    return {"default": 0, "four_pass": _native.MACENKO_CLASSIC, "two_pass": _native.MACENKO_TWO_PASS, "fused": _native.MACENKO_TWO_PASS | _native.MACENKO_FUSE}


@pytest.mark.gpu
def test_real_quadrants_in_every_form(dev, real):
    """The twenty 512 x 512 quadrants of test_1..5 (float32 and uint8 input): per-tile intermediates and the output subsample
    against the reference in the four-pass form; the other forms bit-equal to it.  Records which tiles the two-pass speculation
    could NOT serve (test_5 is 56 % background) instead of demanding that none exist."""
    from stainx_amd.backends.torch_hip_backend import MacenkoHIP

    _, g, quads, _ = real
    be = MacenkoHIP(dev, diag=True)      # (two of the forms below exist in the diagnostic build only; the product's forms are the same code there)
    sm, tmc = torch.from_numpy(g["stain_matrix"]), torch.from_numpy(g["target_max_conc"])
    stride = (512 * 512) // 4096
    for name, dt in (("f32", torch.float32), ("u8", torch.uint8)):
        x = synth.as_dtype(quads, dt).to(dev)
        outs = {}
        for form, flags in _forms().items():
            if form == "fused" and name != "f32":
                continue
            outs[form] = be.transform(x, sm, tmc, _extra_flags=flags)
            params = be.tile_params(x.shape[0])
            if form == "four_pass":
                for i in range(x.shape[0]):
                    assert int(params["n_kept"][i]) == int(g[f"q512_{name}_n_kept"][i]), (name, i)
                    np.testing.assert_allclose(params["cov"][i].numpy(), g[f"q512_{name}_cov"][i], rtol=0, atol=5e-6)
                    np.testing.assert_allclose(params["he"][i].numpy(), g[f"q512_{name}_he"][i], rtol=0, atol=5e-5)
                    np.testing.assert_allclose(params["max_c"][i].numpy(), g[f"q512_{name}_max_c"][i], rtol=1e-4, atol=0)
                assert int(params["fell_back"].max()) == 0, (name, params["fell_back"])      # the four-pass brackets hold on real tissue
            if form in ("two_pass", "fused"):
                slow = (params["fell_back"] & 15) != 0
                print(f"real quadrants {name} {form}: {int(slow.sum())} of {len(slow)} tiles left the speculative path: {slow.nonzero().flatten().tolist()}; "
                      f"candidates % per slot (median over tiles) {[round(float(v), 2) for v in (params['n_candidates'].double() / (512 * 512) * 100).median(0).values]}")
        sub = outs["four_pass"].cpu().reshape(len(quads), 3, -1)[:, :, ::stride]
        want = torch.from_numpy(g[f"q512_{name}_out_sub"])
        diff = (sub.double() - want.double()).abs()
        if name == "u8":
            assert diff.max().item() <= 1 and (diff > 0).float().mean().item() < 2e-3
        else:
            assert diff.max().item() <= TOL_255, diff.max().item()
        out01 = be.transform(x, sm, tmc, normalize_to_0_1=True).cpu().reshape(len(quads), 3, -1)[:, :, ::stride]
        want01 = torch.from_numpy(g[f"q512_{name}_out01_sub"])
        assert (out01.double() - want01.double()).abs().max().item() <= (TOL_UNIT if name == "f32" else 1.0 / 255 + 1e-7)
        for form, o in outs.items():
            assert torch.equal(o.view(torch.uint8), outs["four_pass"].view(torch.uint8)), (name, form)


@pytest.mark.gpu
def test_real_crops_all_dtypes_and_siblings(dev, real):
    from stainx_amd import HistogramMatching, Macenko, Reinhard, StainNormalizerTransform

    imgs, g, _, crops = real
    sm, tmc = torch.from_numpy(g["stain_matrix"]), torch.from_numpy(g["target_max_conc"])
    for name, dt in (("f32", torch.float32), ("u8", torch.uint8), ("bf16", torch.bfloat16)):
        m = Macenko(device=dev, backend="torch_hip")
        m._stain_matrix, m._target_max_conc, m._is_fitted = sm.to(dev), tmc.to(dev), True
        x = synth.as_dtype(crops, dt).to(dev)
        out = m.transform(x).cpu()
        want = golden_tensor(g[f"c224_{name}_out"], name)
        got = thin(out) if name == "f32" else out
        diff = (got.double() - want.double()).abs()
        if name == "u8":
            assert diff.max().item() <= 1 and (diff > 0).float().mean().item() < 2e-3
        elif name == "bf16":
            assert diff.max().item() <= 1.0 and (diff > 0).float().mean().item() < 2e-3
        else:
            assert diff.max().item() <= TOL_255, diff.max().item()
        m.normalize_to_0_1 = True
        out01 = m.transform(x).cpu()
        want01 = golden_tensor(g[f"c224_{name}_out01"], "f32" if name == "u8" else name)
        got01 = thin(out01) if out01.dtype == torch.float32 else out01
        assert (got01.double() - want01.double()).abs().max().item() <= {"f32": TOL_UNIT, "u8": 1.0 / 255 + 1e-7}.get(name, 2.0 ** -8)
    # the fit (the whole target image; the pooled fit over the six crops)
    fit = Macenko(device=dev, backend="torch_hip").fit(imgs[0:1].to(dev))
    np.testing.assert_allclose(fit._stain_matrix.cpu().numpy(), g["stain_matrix"], atol=5e-5)
    np.testing.assert_allclose(fit._target_max_conc.cpu().numpy(), g["target_max_conc"], rtol=1e-4)
    pooled = Macenko(device=dev, backend="torch_hip").fit(crops.to(dev))
    np.testing.assert_allclose(pooled._stain_matrix.cpu().numpy(), g["c224_pooled_he"], atol=5e-5)
    np.testing.assert_allclose(pooled._target_max_conc.cpu().numpy(), g["c224_pooled_max_c"], rtol=1e-4)
    # siblings
    ref512 = imgs[0:1, :, :512, :512].contiguous().to(dev)
    r = Reinhard(device=dev, backend="torch_hip").fit(ref512)
    assert (r.transform(crops.to(dev)).cpu().int() - torch.from_numpy(g["c224_reinhard_u8"]).int()).abs().max().item() <= 1
    rf = thin(r.transform(synth.as_dtype(crops, torch.float32).to(dev)).cpu())
    assert (rf.double() - torch.from_numpy(g["c224_reinhard_f32"]).double()).abs().max().item() <= 2e-4      # [0, 1] scale
    h = HistogramMatching(device=dev, backend="torch_hip").fit(ref512)
    assert torch.equal(h.transform(crops.to(dev)).cpu(), torch.from_numpy(g["c224_hm_u8"]))                  # bit-exact
    hf = thin(h.transform(synth.as_dtype(crops, torch.float32).to(dev)).cpu())
    assert torch.equal(hf, torch.from_numpy(g["c224_hm_f32"]))
    # the module as the reference's example builds it, and batch mode
    t = StainNormalizerTransform(method="macenko", mode="reference", reference=synth.as_dtype(imgs[0:1], torch.bfloat16).to(dev), device=dev, backend="torch_hip")
    got = t(synth.as_dtype(crops, torch.bfloat16).to(dev)).cpu()
    want = golden_tensor(g["c224_module_reference_bf16"], "bf16")
    d = (got.double() - want.double()).abs()
    assert d.max().item() <= 2.0 ** -7 and (d > 0).float().mean().item() < 5e-3, d.max().item()
    tb = StainNormalizerTransform(method="macenko", mode="batch", device=dev, backend="torch_hip", batch_ref_index=2)
    gotb = thin(tb(synth.as_dtype(crops, torch.float32).to(dev)).cpu())
    assert (gotb.double() - torch.from_numpy(g["c224_module_batch_f32"]).double()).abs().max().item() <= TOL_UNIT
