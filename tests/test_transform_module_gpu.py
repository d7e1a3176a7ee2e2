"""GPU tests of ``StainNormalizerTransform`` and the normaliser classes, mirroring the reference's
tests/torch_interface/test_stain_normalizer_transform.py on the HIP backend, plus parity with the
reference outputs recorded in g8 (config-5 shape family: bf16 reference mode)."""
from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle import stain_oracle as so
from stainx_amd import HistogramMatching, Macenko, Reinhard, StainNormalizerTransform, synth
from tests.conftest import TORCH_DTYPES, golden_tensor

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture
def reference():
    return synth.reference_tile(64, 64)


@pytest.fixture
def source():
    return synth.he_batch(2, 64, 64, seed0=123, scale_step=0.15)


def test_matches_reference_golden_reference_and_batch_mode(dev, golden):
    g = golden("g8_transform_module.npz")
    ref, src = torch.from_numpy(g["ref_u8"]), torch.from_numpy(g["src_u8"])
    for name in ("bf16", "f32", "u8"):
        dt = TORCH_DTYPES[name]
        out_name = "f32" if name == "u8" else name            # normalize_to_0_1 promotes uint8 to float32
        tol = {"bf16": 2.0 ** -8, "f32": 1e-4, "u8": 1.0 / 255 + 1e-6}[name]
        t = StainNormalizerTransform(method="macenko", mode="reference", reference=synth.as_dtype(ref, dt).to(dev), backend="torch_hip")
        out = t(synth.as_dtype(src, dt).to(dev))
        want = golden_tensor(g[f"reference_{name}"], out_name)
        assert out.dtype == want.dtype and out.device.type == "cuda"
        assert (out.cpu().double() - want.double()).abs().max().item() <= tol, name
        tb = StainNormalizerTransform(method="macenko", mode="batch", backend="torch_hip", batch_ref_index=1)
        outb = tb(synth.as_dtype(src, dt).to(dev))
        wantb = golden_tensor(g[f"batch_{name}"], out_name)
        assert (outb.cpu().double() - wantb.double()).abs().max().item() <= tol, name


def test_reference_mode_shape_device(dev, reference, source):
    t = StainNormalizerTransform(method="reinhard", mode="reference", reference=reference, device=dev)
    out = t(source)                       # CPU input is moved to the requested device
    assert out.shape == source.shape and out.device.type == "cuda"


def test_default_device_follows_cuda_input(dev, reference, source):
    t = StainNormalizerTransform(method="reinhard", mode="reference", reference=reference.to(dev))
    out = t(source.to(dev))
    assert out.device.type == "cuda" and torch.device(t.normalizer.device).type == "cuda"


def test_device_none_cpu_input_fails_loudly(reference):
    with pytest.raises(ValueError, match="requires a CUDA"):
        StainNormalizerTransform(method="reinhard", mode="reference", reference=reference)     # CPU reference, device=None


def test_explicit_backend_alias_with_device_none(dev, reference, source):
    t = StainNormalizerTransform(method="reinhard", mode="reference", reference=reference.to(dev), backend="torch_cuda")
    assert t(source.to(dev)).device.type == "cuda"
    with pytest.raises(ValueError, match="requires CUDA tensors"):
        t(source)


def test_single_image_roundtrip_rank(dev, reference):
    t = StainNormalizerTransform(method="reinhard", mode="reference", reference=reference, device=dev)
    img = synth.he_tile(64, 64, 5)[0]
    assert t(img).shape == img.shape


def test_macenko_normalize_to_0_1_default_and_override(dev, reference, source):
    ref, src = synth.as_dtype(reference, torch.float32), synth.as_dtype(source, torch.float32)
    t = StainNormalizerTransform(method="macenko", mode="reference", reference=ref, device=dev)
    assert t.normalizer.normalize_to_0_1 is True
    out = t(src)
    assert out.dtype.is_floating_point and float(out.amin()) >= -1e-5 and float(out.amax()) <= 1.0 + 1e-5
    raw = StainNormalizerTransform(method="macenko", mode="reference", reference=ref, device=dev, normalize_to_0_1=False)(src)
    assert float(raw.amax()) > 1.0
    assert torch.equal(out.cpu(), raw.cpu() / 255.0)
    n = Macenko(device=dev, normalize_to_0_1=True).fit(ref.to(dev))
    assert torch.allclose(out, n.transform(src.to(dev)), rtol=0, atol=1e-6)


def test_float_jitter_above_one_not_treated_as_255(dev, reference, source):
    ref = synth.as_dtype(reference, torch.float32)
    src = (synth.as_dtype(source, torch.float32) * 1.6).clamp(0.0, 1.5)
    assert float(src.amax()) > 1.0
    out = StainNormalizerTransform(method="macenko", mode="reference", reference=ref, device=dev)(src)
    assert float(out.mean()) > 0.05 and float(out.amax()) <= 1.0 + 1e-4
    he, mc = so.macenko_fit(ref.numpy())
    want = so.apply_normalize_to_0_1(so.macenko_transform(src.numpy(), he, mc))
    assert np.abs(out.cpu().numpy() - want).max() <= 1e-4


def test_batch_mode_refits(dev, source):
    t = StainNormalizerTransform(method="reinhard", mode="batch", device=dev, batch_ref_index=0)
    out = t(source)
    assert out.shape == source.shape and t.normalizer._is_fitted


def test_hm_channels_last(dev):
    ref = synth.noise_u8((1, 32, 32, 3), 6)
    src = synth.noise_u8((2, 32, 32, 3), 7)
    t = StainNormalizerTransform(method="histogram_matching", mode="reference", reference=ref, device=dev, channel_axis=-1)
    out = t(src)
    assert out.shape == src.shape
    want = so.hm_transform(src.numpy(), so.hm_fit(ref.numpy(), -1), -1)
    assert np.array_equal(out.cpu().numpy(), want)
    n = HistogramMatching(device=dev, channel_axis=-1).fit(ref)
    t2 = StainNormalizerTransform(mode="reference", normalizer=n, device=dev)
    assert t2.channel_axis == -1 and torch.equal(t2(src), out)


def test_prebuilt_normalizer_follows_input_device(dev, reference, source):
    n = Macenko(device=dev, normalize_to_0_1=False).fit(synth.as_dtype(reference, torch.float32))
    t = StainNormalizerTransform(mode="reference", normalizer=n, device=dev, normalize_to_0_1=True)
    assert t.normalizer.normalize_to_0_1 is True
    assert float(t(synth.as_dtype(source, torch.float32)).amax()) <= 1.0 + 1e-5
    # fitted tensors live on the device and are not part of the state dict
    assert n._stain_matrix.device.type == "cuda"
    assert not any("stain" in k or "max_conc" in k for k in t.state_dict())


def test_fit_transform_preserves_shape_and_dtype(dev):
    x = synth.he_batch(1, 40, 24, seed0=3)
    for cls in (Reinhard, Macenko, HistogramMatching):
        y = cls(device=dev, backend="torch_hip").fit_transform(x)
        assert isinstance(y, torch.Tensor) and y.shape == x.shape and y.dtype == x.dtype


def test_dataloader_style_pipeline(dev, reference):
    """Config 5 in miniature: bf16 batches through the module inside a torch DataLoader loop."""
    from torch.utils.data import DataLoader, TensorDataset

    tiles = synth.as_dtype(synth.he_batch(12, 56, 56, seed0=40, scale_step=0.02), torch.bfloat16)
    t = StainNormalizerTransform(method="macenko", mode="reference", reference=synth.as_dtype(synth.reference_tile(56, 56), torch.bfloat16).to(dev))
    he, mc = so.macenko_fit(synth.as_dtype(synth.reference_tile(56, 56), torch.bfloat16).float().numpy())
    for (batch,) in DataLoader(TensorDataset(tiles), batch_size=4):
        out = t(batch.to(dev))
        assert out.dtype == torch.bfloat16 and out.shape == batch.shape
        want = torch.from_numpy(so.macenko_transform(batch.float().numpy(), he, mc)).to(torch.bfloat16) / 255.0
        assert (out.cpu().float() - want.float()).abs().max().item() <= 2.0 ** -8


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_module_follows_its_input_to_another_gpu(reference):
    """device=None: the result lives where the input lives, also when the normaliser was built index-less ("cuda") while GPU 0 was
    current and the batch arrives on GPU 1 ('cuda' and 'cuda:1' used to count as the same device)."""
    t = StainNormalizerTransform(method="macenko", mode="reference", reference=synth.as_dtype(reference, torch.float32).to("cuda:0"))
    x = synth.as_dtype(synth.he_batch(2, 56, 56, seed0=41), torch.float32)
    out0 = t(x.to("cuda:0"))
    out1 = t(x.to("cuda:1"))
    assert out0.device == torch.device("cuda:0") and out1.device == torch.device("cuda:1")
    assert torch.equal(out0.cpu(), out1.cpu())
