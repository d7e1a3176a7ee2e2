"""Host-side behaviour that needs no GPU: the C-ABI library loads and exports every declared symbol,
backend-id validation, the StainNormalizerTransform validation matrix (reference
tests/torch_interface/test_stain_normalizer_transform.py and tests/test_normalizer_template_unit.py),
and that the product never falls back to the CPU."""
from __future__ import annotations

import ctypes
import os
import re
from pathlib import Path

import pytest
import torch

import stainx_amd
from stainx_amd import HistogramMatching, Macenko, Reinhard, StainNormalizerTransform, _native
from stainx_amd.utils import ChannelFormatConverter

ROOT = Path(__file__).resolve().parents[1]


def test_library_exports_every_declared_symbol():
    header = (ROOT / "include" / "stainx_hip.h").read_text()
    # (what stands between `#ifdef SX_DIAG` and its `#endif` is the diagnostic build's: exported by libstainx_diag.so only)
    diag_part = "".join(re.findall(r"#ifdef SX_DIAG(.*?)#endif", header, flags=re.S))
    diag_only = set(re.findall(r"\b(sx_[a-z0-9_]+)\s*\(", diag_part))
    declared = set(re.findall(r"\b(sx_[a-z0-9_]+)\s*\(", header)) - diag_only
    assert declared == set(_native.SIGNATURES), declared ^ set(_native.SIGNATURES)
    assert diag_only == set(_native.DIAG_SIGNATURES), diag_only ^ set(_native.DIAG_SIGNATURES)
    assert _native.library_available(), _native._load_error
    lib = ctypes.CDLL(str(_native.LIB_PATH))
    for name in declared:
        assert hasattr(lib, name), name
    for name in diag_only:
        assert not hasattr(lib, name), name
    diag = ctypes.CDLL(str(_native.DIAG_LIB_PATH))
    for name in declared | diag_only:
        assert hasattr(diag, name), name
    assert _native.require().sx_version() == _native.ABI_VERSION
    # size queries are pure host functions
    assert _native.require().sx_macenko_workspace_bytes(64, 512, 512) > 0
    assert _native.require().sx_macenko_workspace_bytes(0, 512, 512) == 0
    assert _native.require().sx_hm_workspace_bytes(1, 8, 8) >= 3 * 256 * 8
    # which form of the Macenko transform a call takes (host logic only): wide pixels in big batches of mid-sized tiles
    takes = _native.require().sx_macenko_takes_two_pass
    f32, u8, bf16 = _native.DTYPE_CODES[torch.float32], _native.DTYPE_CODES[torch.uint8], _native.DTYPE_CODES[torch.bfloat16]
    assert takes(f32, 64, 512, 512, 0) == 1 and takes(_native.DTYPE_CODES[torch.float64], 64, 512, 512, 0) == 0      # (float64: the four passes)
    assert takes(f32, 1, 512, 512, 0) == 0 and takes(f32, 4, 2048, 2048, 0) == 0 and takes(f32, 1024, 64, 64, 0) == 0 and takes(f32, 1024, 128, 128, 0) == 0
    assert takes(u8, 64, 512, 512, 0) == 1 and takes(bf16, 64, 384, 384, 0) == 1      # narrow pixels: tiles of ~360 x 360 ... 512 x 512 (round 3: dense candidate records)
    assert takes(bf16, 256, 224, 224, 0) == 0 and takes(u8, 164, 320, 320, 0) == 0 and takes(u8, 36, 724, 724, 0) == 0 and takes(u8, 4, 512, 512, 0) == 0
    assert takes(f32, 64, 512, 512, _native.MACENKO_CLASSIC) == 0 and takes(f32, 64, 512, 512, _native.MACENKO_SAMPLED) == 0
    # the diagnostic build's flag forces the two-pass form wherever it is able to run; the product library knows no such flag
    takes_d = _native.require_diag().sx_macenko_takes_two_pass
    assert takes_d(u8, 4, 128, 128, _native.MACENKO_TWO_PASS) == 1 and takes_d(f32, 1, 8, 8, _native.MACENKO_TWO_PASS) == 0
    assert takes(u8, 4, 128, 128, _native.MACENKO_TWO_PASS) == 0
    # six public flag bits; the diagnostic ones are declared only under SX_DIAG
    public = re.findall(r"^#define (SX_MACENKO_[A-Z0-9_]+) (\d+)u", header.split("#ifdef SX_DIAG")[0], flags=re.M)
    assert sorted(name for name, _ in public if name != "SX_MACENKO_PARAM_FLOATS") == ["SX_MACENKO_CHANNELS_LAST", "SX_MACENKO_CLASSIC", "SX_MACENKO_NORMALIZE_0_1", "SX_MACENKO_OUT_BF16",
                                                                                     "SX_MACENKO_OUT_F16", "SX_MACENKO_SAMPLED"]


def test_public_surface():
    assert set(stainx_amd.__all__) >= {"Macenko", "Reinhard", "HistogramMatching", "StainNormalizerTransform", "StainNormalizerBase"}
    for cls in (Macenko, Reinhard, HistogramMatching):
        assert issubclass(cls, stainx_amd.StainNormalizerBase)
        for method in ("fit", "transform", "fit_transform"):
            assert callable(getattr(cls, method))


def test_transform_requires_fit():
    n = Reinhard(backend="torch_hip", device="cuda")
    with pytest.raises(ValueError, match="fit"):
        n.transform(torch.rand(1, 3, 8, 8))


def test_backend_ids():
    assert Macenko(device="cuda", backend="torch_cuda").backend == "torch_hip"       # the reference's id is an alias
    assert Macenko(device="cuda").backend == "torch_hip"
    with pytest.raises(ValueError, match="Unsupported backend"):
        Macenko(backend="numpy")
    with pytest.raises(ValueError, match="Unsupported backend 'torch'"):
        Reinhard(backend="torch")
    with pytest.raises(ValueError, match="precision must be"):
        Macenko(precision="ultra")
    assert Macenko(device="cuda", precision="fast").engine_options() == {"precision": "fast"}
    assert Macenko(device="cuda", precision="sampled").engine_options() == {"precision": "sampled"}


def test_no_cpu_fallback():
    """A CPU device is refused by the backend itself; nothing is computed with torch ops instead."""
    n = Reinhard(device="cpu")
    with pytest.raises(ValueError, match="requires a CUDA"):
        n.fit(torch.rand(1, 3, 8, 8))
    if not torch.cuda.is_available():
        with pytest.raises((RuntimeError, AssertionError, ValueError)):
            Macenko(device="cuda").fit(torch.rand(1, 3, 8, 8))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "_load_error", None)
    monkeypatch.setenv("STAINX_HIP_LIB", str(tmp_path / "nope.so"))
    assert not _native.library_available()
    with pytest.raises(ImportError, match="not built or not loadable"):
        _native.require()
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "_load_error", None)
    monkeypatch.delenv("STAINX_HIP_LIB")
    assert _native.library_available()


class TestPrepareForNormalizer:
    def test_nhwc_batch_permute_keeps_n_and_device(self):
        x = torch.randn(4, 32, 48, 3)
        out = ChannelFormatConverter(channel_axis=-1).prepare_for_normalizer(x)
        assert out.shape == (4, 3, 32, 48) and out.device == x.device
        assert torch.allclose(out.permute(0, 2, 3, 1), x)

    def test_hwc_gains_batch_axis(self):
        assert ChannelFormatConverter(channel_axis=3).prepare_for_normalizer(torch.randn(8, 9, 3)).shape == (1, 3, 8, 9)

    def test_channels_first_passthrough(self):
        x = torch.randn(2, 3, 16, 16)
        assert ChannelFormatConverter(channel_axis=1).prepare_for_normalizer(x).data_ptr() == x.data_ptr()

    def test_unknown_channel_axis_raises(self):
        with pytest.raises(ValueError, match="Unsupported channel_axis"):
            ChannelFormatConverter(channel_axis=0)

    def test_to_hwc(self):
        x = torch.arange(2 * 3 * 4 * 5).reshape(2, 3, 4, 5)
        assert ChannelFormatConverter(1).to_hwc(x[:1], squeeze_batch=True).shape == (4, 5, 3)


class TestTransformValidation:
    """The constructor / layout validation matrix of reference transforms.py:93-140, 200-216."""

    def ref(self):
        return (torch.rand(1, 3, 16, 16) * 255).round().to(torch.uint8)

    def test_bad_mode(self):
        with pytest.raises(ValueError, match="Unsupported mode"):
            StainNormalizerTransform(method="reinhard", mode="online")

    def test_unknown_method(self):
        with pytest.raises(ValueError, match="Unknown method"):
            StainNormalizerTransform(method="vahadane", mode="batch")

    def test_reference_mode_needs_reference(self):
        with pytest.raises(ValueError, match="requires a reference tensor"):
            StainNormalizerTransform(method="reinhard", mode="reference")

    def test_normalize_to_0_1_rejected_for_reinhard(self):
        with pytest.raises(ValueError, match="only applies to Macenko"):
            StainNormalizerTransform(method="reinhard", mode="batch", normalize_to_0_1=True)
        with pytest.raises(ValueError, match="only applies to Macenko"):
            StainNormalizerTransform(mode="batch", normalizer=Reinhard(device="cuda"), normalize_to_0_1=True)

    def test_gpu_backend_plus_cpu_device_rejected(self):
        for backend in ("torch_hip", "torch_cuda"):
            with pytest.raises(ValueError, match="requires a CUDA device"):
                StainNormalizerTransform(method="reinhard", mode="batch", backend=backend, device="cpu")

    def test_macenko_defaults_and_prebuilt_flag(self):
        t = StainNormalizerTransform(method="macenko", mode="batch")
        assert t.normalizer.normalize_to_0_1 is True
        assert StainNormalizerTransform(method="macenko", mode="batch", normalize_to_0_1=False).normalizer.normalize_to_0_1 is False
        n = Macenko(device="cuda", normalize_to_0_1=True)
        assert StainNormalizerTransform(mode="batch", normalizer=n, normalize_to_0_1=False).normalizer.normalize_to_0_1 is False
        n2 = Macenko(device="cuda", normalize_to_0_1=False)
        assert StainNormalizerTransform(mode="batch", normalizer=n2).normalizer.normalize_to_0_1 is False   # left alone

    def test_macenko_rejects_nhwc_channel_axis(self):
        with pytest.raises(ValueError, match="only supported for histogram_matching"):
            StainNormalizerTransform(method="macenko", mode="batch", channel_axis=-1)
        with pytest.raises(ValueError, match="only supported for histogram_matching"):
            StainNormalizerTransform(mode="batch", normalizer=Reinhard(device="cuda"), channel_axis=3)

    def test_prebuilt_hm_channel_axis(self):
        n = HistogramMatching(device="cuda", channel_axis=-1)
        assert StainNormalizerTransform(mode="batch", normalizer=n).channel_axis == -1
        with pytest.raises(ValueError, match="conflicts with prebuilt"):
            StainNormalizerTransform(mode="batch", normalizer=HistogramMatching(device="cuda", channel_axis=1), channel_axis=-1)

    def test_layout_checks_happen_before_any_gpu_work(self):
        t = StainNormalizerTransform(method="macenko", mode="batch")
        with pytest.raises(ValueError, match="Expected NCHW"):
            t((torch.rand(2, 16, 16, 3) * 255).to(torch.uint8))
        with pytest.raises(ValueError, match="Expected CHW/NCHW"):
            t(torch.rand(16, 16))
        hm = StainNormalizerTransform(method="histogram_matching", mode="batch", channel_axis=-1)
        with pytest.raises(ValueError, match="channels-last histogram matching expects"):
            hm(torch.rand(2, 3, 16, 16))

    def test_state_dict_has_no_fitted_parameters(self):
        t = StainNormalizerTransform(method="macenko", mode="batch")
        assert not any("stain" in k or "max_conc" in k for k in t.state_dict())

    def test_batch_ref_index_out_of_range(self):
        t = StainNormalizerTransform(method="reinhard", mode="batch", batch_ref_index=5, device="cpu")
        with pytest.raises(IndexError, match="out of range"):
            t(torch.rand(2, 3, 8, 8))


def test_hm_reference_cache_survives_inference_tensors():
    """the stacked-reference cache read `_version`, which inference tensors do not have (RuntimeError on every transform
    after a fit under torch.inference_mode()).  Host logic only: the object is built without its device checks."""
    from stainx_amd.backends.torch_hip_backend import HistogramMatchingHIP

    be = HistogramMatchingHIP.__new__(HistogramMatchingHIP)
    be.device = torch.device("cpu")
    with torch.inference_mode():
        hists = [torch.full((256,), 1.0 / 256) for _ in range(3)]
    assert hists[0].is_inference()
    a = be._stack_reference(hists, 3)
    b = be._stack_reference(hists, 3)
    assert a.shape == (3, 256) and torch.equal(a, b) and getattr(be, "_ref_cache", None) is None      # never cached
    plain = [torch.full((256,), 1.0 / 256) for _ in range(3)]
    c = be._stack_reference(plain, 3)
    assert be._stack_reference(plain, 3) is c                                                      # cached while unchanged
    plain[1].mul_(2.0)
    d = be._stack_reference(plain, 3)
    assert d is not c and float(d[1, 0]) == pytest.approx(2.0 / 256)                               # an in-place change is seen


def test_force_collectives_flag_is_parsed_and_read_late(monkeypatch):
    import stainx_amd.distributed as sxd

    monkeypatch.setattr(sxd, "FORCE_COLLECTIVES", None)
    for value, want in (("", False), ("0", False), ("false", False), ("1", True), ("true", True), ("YES", True)):
        monkeypatch.setenv("STAINX_FORCE_COLLECTIVES", value)
        assert sxd.force_collectives() is want, value
    monkeypatch.setattr(sxd, "FORCE_COLLECTIVES", True)
    monkeypatch.setenv("STAINX_FORCE_COLLECTIVES", "0")
    assert sxd.force_collectives() is True


def test_bench_gpus_n_starts_n_ranks_or_refuses():
    """`python bench.py --gpus N` without a rank environment starts the N ranks itself (one per GPU, torch.distributed.run on 127.0.0.1)
    and refuses loudly on a node with fewer GPUs (the flag used to be parsed and never read)."""
    import subprocess
    import sys

    import bench

    cmd = bench.launcher_command(4, ["--gpus", "4", "--steps", "7"], 29777)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29777"
    assert cmd[-5].endswith("bench.py") and cmd[-4:] == ["--gpus", "4", "--steps", "7"]
    # no GPU in this container: the launcher path is taken and refuses (a non-zero exit, the reason on stderr) before any rank starts
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "STAINX_BENCH_REHEARSE")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=env, timeout=300)
    if torch.cuda.device_count() < 2:
        assert r.returncode != 0 and "refusing to time fewer GPUs" in r.stderr, (r.returncode, r.stderr[-500:])
