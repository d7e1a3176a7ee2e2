"""Histogram matching in one launch (stainx_amd/csrc/histmatch_resident.hpp, diagnostic build: a design study, measured slower): bit for bit the two-kernel form (reached here through
the pooled entry points sx_hm_counts + sx_hm_apply, which never take the one-launch form) and the oracle; repeated calls on a READY workspace."""
import numpy as np
import pytest
import torch

from stainx_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def be(dev):
    from stainx_amd.backends.torch_hip_backend import HistogramMatchingHIP

    return HistogramMatchingHIP(dev, diag=True)


def _ref_hist(dev):
    g = torch.Generator().manual_seed(11)
    h = torch.rand(3, 256, generator=g) + 0.01
    h = (h / h.sum(dim=1, keepdim=True)).to(dev)
    return [h[0].contiguous(), h[1].contiguous(), h[2].contiguous()]


def _parity(be) -> int:
    """Tables::res_parity: toggled by every call that took the one-launch form (include/stainx_hip.h: sx_hm_workspace_parity_offset)."""
    ws = be.last_workspace
    off = int(be._lib.sx_hm_workspace_parity_offset())
    return int(ws[off:off + 4].view(torch.int32).item())


def _two_kernels(be, x, ref):
    counts = be.local_counts(x)
    return be.apply_with_counts(x, counts, int(x.shape[0] * x.shape[2] * x.shape[3]), ref)


@pytest.mark.parametrize("shape", [(48, 512, 512), (171, 256, 256), (6, 1024, 2048), (64, 1024, 1024), (23, 512, 1024)])
def test_one_launch_equals_two_kernels(be, dev, shape):
    n, h, w = shape
    x = synth.he_batch(n, h, w, seed0=5100 + n).to(dev)
    assert x.dtype == torch.uint8
    ref = _ref_hist(dev)
    before = _parity(be) if be.last_workspace is not None else None
    got = be.transform(x, ref)
    want = _two_kernels(be, x, ref)
    assert torch.equal(got, want)
    assert be.workspace_status() == 0
    got2 = be.transform(x, ref)      # the READY workspace again
    assert torch.equal(got2, want) and be.workspace_status() == 0
    if before is not None:
        assert _parity(be) == before      # two one-launch calls: the parity is back where it was


def test_one_launch_is_taken_and_alternates_its_counter_sets(be, dev):
    x = synth.he_batch(48, 512, 512, seed0=5200).to(dev)
    ref = _ref_hist(dev)
    be.transform(x, ref)
    p0 = _parity(be)
    be.transform(x, ref)
    assert _parity(be) == 1 - p0
    small = synth.he_batch(2, 256, 256, seed0=5201).to(dev)      # too small for the one-launch form: the parity stays
    p1 = _parity(be)
    be.transform(small, ref)
    assert _parity(be) == p1


def test_shapes_alternating_on_one_workspace(be, dev):
    ref = _ref_hist(dev)
    xs = [synth.he_batch(48, 512, 512, seed0=5300).to(dev), synth.he_batch(4, 300, 300, seed0=5301).to(dev), synth.he_batch(171, 256, 256, seed0=5302).to(dev)]
    wants = [_two_kernels(be, x, ref) for x in xs]
    for _ in range(3):
        for x, want in zip(xs, wants):
            assert torch.equal(be.transform(x, ref), want) and be.workspace_status() == 0


def test_one_launch_matches_the_oracle(be, dev):
    from oracle import stain_oracle as so

    x = synth.he_batch(48, 512, 512, seed0=5400)
    ref = _ref_hist(dev)
    got = be.transform(x.to(dev), ref).cpu().numpy()
    want = so.hm_transform(x.numpy(), [r.cpu().numpy() for r in ref])
    assert np.array_equal(got, want)


def test_degenerate_batches(be, dev):
    ref = _ref_hist(dev)
    for fill in (0, 255, 137):
        x = torch.full((48, 3, 512, 512), fill, dtype=torch.uint8, device=dev)
        assert torch.equal(be.transform(x, ref), _two_kernels(be, x, ref))
    x = synth.he_batch(48, 512, 512, seed0=5500).to(dev)
    x[:, 1] = 255      # a constant channel beside two ordinary ones
    assert torch.equal(be.transform(x, ref), _two_kernels(be, x, ref))
