"""The four passes over float32 tiles with 8-bit codes behind the first pass (stainx_amd/csrc/macenko.hip: Coded<F>):
bit for bit what the same passes produce reading the float pixels every time (SX_MACENKO_NO_CODES, diagnostic build), for tiles that
are grey levels, tiles that are not, and batches that mix them; and the product library against the oracle's golden vectors."""
import numpy as np
import pytest
import torch

from stainx_amd import _native, synth

pytestmark = pytest.mark.gpu

CLASSIC = _native.MACENKO_CLASSIC
NO_CODES = _native.MACENKO_NO_CODES


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def be(dev):
    from stainx_amd.backends.torch_hip_backend import MacenkoHIP

    return MacenkoHIP(dev, diag=True)


def _ref():
    sm = torch.tensor([[0.5626, 0.2159], [0.7201, 0.8012], [0.4062, 0.5581]], dtype=torch.float32)
    return sm, torch.tensor([1.9705, 1.0308], dtype=torch.float32)


TWO_PASS = _native.MACENKO_TWO_PASS


def _both(be, x, form=CLASSIC, **kw):
    sm, tmc = _ref()
    coded = be.transform(x, sm, tmc, _extra_flags=form, **kw)
    plain = be.transform(x, sm, tmc, _extra_flags=form | NO_CODES, **kw)
    torch.cuda.synchronize()
    return coded, plain


def _takes_codes(be, x):
    n, _, h, w = x.shape
    lib = be._lib
    code = _native.DTYPE_CODES[torch.float32]
    return lib.sx_macenko_workspace_bytes_for(code, n, h, w, CLASSIC) > lib.sx_macenko_workspace_bytes_for(_native.DTYPE_CODES[torch.uint8], n, h, w, CLASSIC)


@pytest.mark.parametrize("shape", [(16, 256, 256), (8, 512, 512), (5, 448, 512), (24, 224, 224), (3, 1024, 1024), (64, 128, 128)])
@pytest.mark.parametrize("unit", [False, True])
@pytest.mark.parametrize("form", [CLASSIC, TWO_PASS])
def test_grey_level_tiles_coded_equals_plain(be, dev, shape, unit, form):
    n, h, w = shape
    x = synth.as_dtype(synth.he_batch(n, h, w, seed0=4100 + n), torch.float32).to(dev)
    assert _takes_codes(be, x)
    coded, plain = _both(be, x, form, normalize_to_0_1=unit)
    assert torch.equal(coded, plain)
    if form == TWO_PASS:      # and the two forms agree with each other, codes or not
        assert torch.equal(coded, _both(be, x, CLASSIC, normalize_to_0_1=unit)[0])


def test_tiles_that_are_not_grey_levels_stay_float(be, dev):
    g = torch.Generator().manual_seed(7)
    x = synth.as_dtype(synth.he_batch(8, 256, 256, seed0=4200), torch.float32)
    x = (x + (torch.rand(x.shape, generator=g) - 0.5) * 1e-3).clamp_(0.0, 1.0).to(dev)      # no element is k / 255 any more
    for form in (CLASSIC, TWO_PASS):
        coded, plain = _both(be, x, form)
        assert torch.equal(coded, plain)


def test_mixed_batch_one_odd_element(be, dev):
    """One element of one tile off by one ulp: that tile is read as floats, its neighbours as codes -- same bits either way."""
    x = synth.as_dtype(synth.he_batch(16, 256, 256, seed0=4300), torch.float32)
    x[5, 1, 100, 37] = torch.nextafter(x[5, 1, 100, 37], torch.tensor(2.0))
    x[11, 2, 255, 255] = 0.123456
    x = x.to(dev)
    coded, plain = _both(be, x)
    assert torch.equal(coded, plain)
    two, two_plain = _both(be, x, TWO_PASS)
    assert torch.equal(two, two_plain) and torch.equal(two, coded)
    # and the odd tiles differ from their all-grey-level versions (the odd element mattered: the check is not vacuous)
    y = synth.as_dtype(synth.he_batch(16, 256, 256, seed0=4300), torch.float32).to(dev)
    sm, tmc = _ref()
    clean = be.transform(y, sm, tmc, _extra_flags=CLASSIC)
    assert torch.equal(clean[0], coded[0]) and not torch.equal(clean[11], coded[11])


def test_values_outside_the_unit_range_and_nans_do_not_pass_as_codes(be, dev):
    x = synth.as_dtype(synth.he_batch(16, 256, 256, seed0=4400), torch.float32)
    x[2, 0, 0, 0] = -0.25
    x[3, 1, 17, 19] = 1.5
    x[4, 2, 200, 100] = 300.0
    x = x.to(dev)
    for form in (CLASSIC, TWO_PASS):
        coded, plain = _both(be, x, form)
        assert torch.equal(coded, plain)


def test_repeated_calls_on_one_workspace_alternate_kinds(be, dev):
    """The per-tile flag is the call's number: a tile flagged by one call is not flagged for the next."""
    a = synth.as_dtype(synth.he_batch(16, 256, 256, seed0=4500), torch.float32).to(dev)
    b = (a + 1e-4).clamp(0.0, 1.0)
    sm, tmc = _ref()
    want_a = be.transform(a, sm, tmc, _extra_flags=CLASSIC | NO_CODES)
    want_b = be.transform(b, sm, tmc, _extra_flags=CLASSIC | NO_CODES)
    for _ in range(3):
        assert torch.equal(be.transform(b, sm, tmc, _extra_flags=CLASSIC), want_b)
        assert torch.equal(be.transform(a, sm, tmc, _extra_flags=CLASSIC), want_a)
        assert torch.equal(be.transform(b, sm, tmc, _extra_flags=TWO_PASS), want_b)
        assert torch.equal(be.transform(a, sm, tmc, _extra_flags=TWO_PASS), want_a)


def test_real_tissue_coded_equals_two_pass_and_plain(be, dev):
    d = np.load("tests/golden/g11_real_images.npz")
    imgs = torch.from_numpy(d["images_u8"])      # (6, 3, 1024, 1024)
    tiles = [imgs[i, :, y0:y0 + 256, x0:x0 + 256] for i in range(imgs.shape[0]) for (y0, x0) in ((0, 0), (768, 768), (300, 200))]
    x = (torch.stack(tiles).to(torch.float32) / 255.0).to(dev)
    coded, plain = _both(be, x)
    assert torch.equal(coded, plain)
    sm, tmc = _ref()
    two = be.transform(x, sm, tmc, _extra_flags=_native.MACENKO_TWO_PASS)
    assert torch.equal(coded, two)


def test_product_library_takes_the_codes_and_matches_the_oracle(dev):
    from oracle import stain_oracle as so
    from stainx_amd.backends.torch_hip_backend import MacenkoHIP

    be = MacenkoHIP(dev)
    x = synth.as_dtype(synth.he_batch(16, 256, 256, seed0=4600), torch.float32)
    sm, tmc = _ref()
    got = be.transform(x.to(dev), sm, tmc, _extra_flags=CLASSIC).cpu().numpy()
    want = so.macenko_transform(x.numpy(), sm.numpy(), tmc.numpy())
    assert np.abs(got - want).max() <= 2.55e-2      # 0-255 scale (the tolerance of tests/test_macenko_gpu.py)


# ---- Reinhard: the same codes between its statistics pass and its apply pass (stainx_amd/csrc/reinhard.hip: Codes) ---------------------
def _reinhard_both(dev, x, mean, std):
    """(with the codes, without): the C ABI on a workspace with / without room for them."""
    lib = _native.require()
    n, _, h, w = x.shape
    code = _native.DTYPE_CODES[x.dtype]
    outs = []
    for nbytes in (int(lib.sx_reinhard_workspace_bytes_for(code, n, h, w)), int(lib.sx_reinhard_workspace_bytes(n, h, w))):
        ws = torch.full((nbytes,), 0x5A, dtype=torch.uint8, device=dev)
        out = torch.empty_like(x)
        rc = lib.sx_reinhard_transform(x.data_ptr(), out.data_ptr(), code, n, h, w, mean.data_ptr(), std.data_ptr(), ws.data_ptr(), ws.numel(), _native.stream_ptr(dev))
        _native.check(rc, "sx_reinhard_transform")
        torch.cuda.synchronize()
        outs.append(out)
    return outs


@pytest.mark.parametrize("shape", [(16, 256, 256), (8, 512, 512), (24, 224, 224), (9, 300, 500), (5, 300, 500)])
def test_reinhard_coded_equals_plain(dev, shape):
    n, h, w = shape
    lib = _native.require()
    f32 = _native.DTYPE_CODES[torch.float32]
    if n * h * w >= 1 << 20:      # (smaller batches run without: their passes are bound by latency, not by bytes)
        assert lib.sx_reinhard_workspace_bytes_for(f32, n, h, w) >= lib.sx_reinhard_workspace_bytes(n, h, w) + 3 * n * h * w
    assert lib.sx_reinhard_workspace_bytes_for(_native.DTYPE_CODES[torch.uint8], n, h, w) == lib.sx_reinhard_workspace_bytes(n, h, w)
    mean = torch.tensor([170.0, 150.0, 120.0], dtype=torch.float32, device=dev)
    std = torch.tensor([40.0, 12.0, 9.0], dtype=torch.float32, device=dev)
    x = synth.as_dtype(synth.he_batch(n, h, w, seed0=4700 + n), torch.float32).to(dev)
    coded, plain = _reinhard_both(dev, x, mean, std)
    assert torch.equal(coded, plain)
    # tiles that are not grey levels, and a batch that mixes the two kinds
    y = x.clone()
    y[1] = (y[1] + 3e-4).clamp(0.0, 1.0)
    y[n - 1, 2, h - 1, w - 1] = float("nan") if n > 5 else 0.777
    coded, plain = _reinhard_both(dev, y, mean, std)
    assert torch.equal(torch.nan_to_num(coded, nan=-1.0), torch.nan_to_num(plain, nan=-1.0))


def test_reinhard_backend_takes_the_codes_and_matches_the_oracle(dev):
    from oracle import stain_oracle as so
    from stainx_amd.backends.torch_hip_backend import ReinhardHIP

    be = ReinhardHIP(dev)
    x = synth.as_dtype(synth.he_batch(16, 256, 256, seed0=4800), torch.float32)
    mean, std = so.reinhard_fit(synth.reference_tile(96, 96).numpy())
    want = so.reinhard_transform(x.numpy(), mean, std)
    for _ in range(2):      # (the second call runs on the READY workspace)
        got = be.transform(x.to(dev), torch.from_numpy(mean), torch.from_numpy(std)).cpu().numpy()
        assert np.abs(got - want).max() <= 1e-4
    assert be.workspace_status() == 0


def test_captured_call_replayed_on_other_data_keeps_its_number(be, dev):
    """A captured call carries ITS number (the value its first pass flags a tile with) into every replay: a tile flagged in one replay
    stays flagged in the next -- it is then read as floats, which is always right -- and a tile that stops being grey levels is flagged."""
    sm, tmc = (t.to(dev) for t in _ref())      # (on the device: a capture admits no host-to-device copy)
    clean = synth.as_dtype(synth.he_batch(16, 256, 256, seed0=5600), torch.float32).to(dev)
    odd = clean.clone()
    odd[3] = (odd[3] + 2e-4).clamp(0.0, 1.0)
    odd[9, 1, 7, 7] = 0.4242
    want_clean = be.transform(clean, sm, tmc, _extra_flags=CLASSIC | NO_CODES)
    want_odd = be.transform(odd, sm, tmc, _extra_flags=CLASSIC | NO_CODES)
    for form in (CLASSIC, TWO_PASS):
        x = clean.clone()
        s = torch.cuda.Stream(dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(2):
                be.transform(x, sm, tmc, _extra_flags=form)
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            out = be.transform(x, sm, tmc, _extra_flags=form)
        for data, want in ((clean, want_clean), (odd, want_odd), (clean, want_clean), (odd, want_odd)):
            x.copy_(data)
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, want)
