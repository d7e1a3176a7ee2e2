"""numpy restatement of stainx's ``backend="torch"`` CPU algorithms (TEST INFRASTRUCTURE).

This file is the parity oracle of the repository: a plain numpy (float32
arithmetic, LAPACK through numpy) restatement of the three normalisers of
rendeirolab/stainx v0.1.4.  Every function cites the reference lines it
follows (paths relative to /root/reference).  It is pinned against outputs of
the reference itself -- see ``tests/golden/make_golden.py`` and
``tests/test_oracle_golden.py``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module.  The product never does.

All arrays are numpy; images are NCHW unless stated.  float32 is kept
throughout wherever the reference computes in float32.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32

IO = F32(240.0)      # src/stainx/backends/torch_backend.py:541
BETA = F32(0.15)     # src/stainx/backends/torch_backend.py:542
ALPHA = 1.0          # src/stainx/backends/torch_backend.py:543


# --------------------------------------------------------------------------
# dtype gates                       (torch_backend.py:104-131)
# --------------------------------------------------------------------------
def to_unit_float(images: np.ndarray) -> np.ndarray:
    """uint8 -> float32/255, any float -> float32 as is (torch_backend.py:104-113)."""
    if images.dtype == np.uint8:
        return images.astype(F32) / F32(255.0)
    return images.astype(F32)


def _cast_like_torch(result: np.ndarray, dtype) -> np.ndarray:
    """``tensor.to(dtype)``: uint8 truncates, floats round to nearest even."""
    if np.dtype(dtype) == np.uint8:
        return np.trunc(result).astype(np.uint8)
    return result.astype(dtype)


def restore_dtype(result: np.ndarray, dtype, *, in_0_255: bool) -> np.ndarray:
    """torch_backend.py:122-131 (``preserve_dtype_torch``)."""
    if not in_0_255 and np.dtype(dtype) == np.uint8:
        result = np.clip(result * F32(255.0), F32(0.0), F32(255.0))
    elif in_0_255:
        result = np.clip(result, F32(0.0), F32(255.0))
    return _cast_like_torch(result, dtype)


# --------------------------------------------------------------------------
# Macenko                           (torch_backend.py:358-560)
# --------------------------------------------------------------------------
def optical_density(unit: np.ndarray) -> np.ndarray:
    """OD = -log((x*255 + 1)/240), op order mul, add, div, log, neg (torch_backend.py:550)."""
    return -np.log((unit * F32(255.0) + F32(1.0)) / IO)


def nearest_rank(values: np.ndarray, q: float) -> np.float32:
    """k = 1 + round(0.01*q*(n-1)) (half-to-even), k-th smallest (torch_backend.py:363-365)."""
    flat = np.asarray(values, dtype=F32).reshape(-1)
    k = 1 + round(0.01 * float(q) * (flat.size - 1))
    return np.partition(flat, k - 1)[k - 1]


def od_covariance(od_rows: np.ndarray) -> np.ndarray:
    """Unbiased 3x3 covariance of (n,3) rows, centred float32 matmul (torch_backend.py:384-397)."""
    n = od_rows.shape[0]
    if n <= 1:
        return np.zeros((3, 3), dtype=F32)
    # torch's float32 `mean` adds with a cascade of vector accumulators (error ~1e-7 relative whatever n); numpy's float32 sum over
    # a strided axis is a plain running sum, which on the 16.7 M pixels of a pooled 64-tile fit loses two digits of the mean and
    # turns the stain plane by several degrees (found by tests/test_distributed_gpu.py::test_config4_...: the real reference agrees
    # with a float64 evaluation there, this restatement did not).  The float64-accumulated mean, rounded to float32, is torch's
    # value to the last bit or two.
    xt = np.ascontiguousarray(od_rows.T.astype(F32))
    centred = xt - xt.mean(axis=1, keepdims=True, dtype=np.float64).astype(F32)
    return (centred @ centred.T) / F32(n - 1)


def plane_vectors(cov: np.ndarray, signs=None) -> np.ndarray:
    """Eigenvectors of the middle and largest eigenvalue, columns [mid, max] (torch_backend.py:414-415).

    The sign of an eigenvector is arbitrary and LAPACK builds disagree about it.  The transform is
    invariant to it whenever the angles do not wrap around +-pi (every real H&E tile); on near-isotropic
    inputs (noise, near-white tiles) it is not, which is the ill-posedness the reference's own tests
    describe (tests/torch_interface/test_correctness_against_references.py:30-34).  ``signs``:
    ``None`` keeps LAPACK's choice; ``"positive_sum"`` makes each column's component sum non-negative
    (the convention of the HIP kernels); a pair ``(s_mid, s_max)`` of +-1 multiplies the columns.
    """
    _, vecs = np.linalg.eigh(cov.astype(F32))
    vecs = vecs[:, [1, 2]].astype(F32)
    if signs is None:
        return vecs
    if isinstance(signs, str):
        assert signs == "positive_sum", signs
        flip = np.where(vecs.astype(np.float64).sum(axis=0) < 0, F32(-1), F32(1))
        return (vecs * flip[None, :]).astype(F32)
    return (vecs * np.asarray(signs, dtype=F32)[None, :]).astype(F32)


def stain_vectors_from_angles(vecs: np.ndarray, phi_lo: np.float32, phi_hi: np.float32) -> np.ndarray:
    """vMin/vMax and the H-before-E ordering test (torch_backend.py:425-439)."""
    lo = np.array([np.cos(F32(phi_lo)), np.sin(F32(phi_lo))], dtype=F32)
    hi = np.array([np.cos(F32(phi_hi)), np.sin(F32(phi_hi))], dtype=F32)
    v_min = (vecs @ lo).astype(F32)
    v_max = (vecs @ hi).astype(F32)
    if v_min[0] > v_max[0]:
        return np.stack([v_min, v_max], axis=1)
    return np.stack([v_max, v_min], axis=1)


def concentrations(he: np.ndarray, od_3xp: np.ndarray) -> np.ndarray:
    """Least-squares C with HE @ C ~= OD for all pixels (torch_backend.py:376-381, 444)."""
    sol = np.linalg.lstsq(he.astype(F32), od_3xp.astype(F32), rcond=None)[0]
    return sol.astype(F32)


def macenko_tile_params(od_chw: np.ndarray, *, allow_fallback: bool = True, signs=None) -> dict:
    """Per-tile stain estimate: everything up to maxC (torch_backend.py:399-449).

    Returns the intermediates the golden fixtures record.
    """
    od_rows = od_chw.reshape(3, -1).T                      # (P,3)   :401
    keep = od_rows.min(axis=1) >= BETA                      # :404-405
    kept = od_rows[keep]
    if allow_fallback and kept.shape[0] < 3:                # :409-410
        kept = od_rows
    cov = od_covariance(kept)                               # :413
    vecs = plane_vectors(cov, signs)                        # :414-415
    proj = (kept @ vecs).astype(F32)                        # :417
    phi = np.arctan2(proj[:, 1], proj[:, 0]).astype(F32)    # :418
    phi_lo = nearest_rank(phi, ALPHA)                       # :421
    phi_hi = nearest_rank(phi, 100.0 - ALPHA)               # :422
    he = stain_vectors_from_angles(vecs, phi_lo, phi_hi)    # :425-439
    conc = concentrations(he, od_chw.reshape(3, -1))        # :442-444
    max_c = np.array([nearest_rank(conc[0], 99), nearest_rank(conc[1], 99)], dtype=F32)  # :447-449
    return {"n_kept": int(kept.shape[0]), "cov": cov, "vecs": vecs, "phi_lo": F32(phi_lo), "phi_hi": F32(phi_hi),
            "he": he, "max_c": max_c, "conc": conc}


def macenko_transform(images: np.ndarray, stain_matrix: np.ndarray, target_max_conc: np.ndarray,
                      *, return_params: bool = False, signs=None):
    """``MacenkoTorch.transform`` (torch_backend.py:521-560): output ~[0,255] in the input dtype."""
    sm = np.asarray(stain_matrix, dtype=F32)
    if sm.shape != (3, 2):
        raise ValueError(f"stain_matrix must have shape (3, 2), got {sm.shape}")
    if images.ndim != 4:
        raise ValueError(f"Macenko expects NCHW images, got shape {images.shape}")
    n_img, chans, height, width = images.shape
    if chans != 3:
        raise ValueError(f"Macenko expects 3 channels in dim 1 (NCHW), got C={chans}")
    tmc = np.asarray(target_max_conc, dtype=F32).reshape(-1)
    od_all = optical_density(to_unit_float(images))          # :550
    out = np.empty((n_img, 3, height, width), dtype=F32)
    params = []
    for n in range(n_img):                                   # :556-558
        p = macenko_tile_params(od_all[n], signs=signs)
        conc = p.pop("conc")
        scaled = conc * (tmc / p["max_c"])[:, None]          # :452-453
        od_new = (sm @ scaled).astype(F32)                   # :455
        rgb = np.clip(IO * np.exp(-od_new), F32(0), F32(255))  # :458-459
        out[n] = rgb.reshape(3, height, width)
        params.append(p)
    result = restore_dtype(out, images.dtype, in_0_255=True)  # :560
    return (result, params) if return_params else result


def macenko_transform_fast(images: np.ndarray, stain_matrix: np.ndarray, target_max_conc: np.ndarray, *, signs=None) -> np.ndarray:
    """The reference's ``precision="fast"`` native path (src/stainx_cuda_torch/csrc/macenko.cu:116-191), restated: the same
    algorithm with EXACT nearest-rank percentiles, but the big per-pixel tensors in float16 -- optical densities and plane
    vectors rounded to float16 before the projection (:135-137), angles in float16 (:140), concentrations rounded to float16
    for the percentile and the rescale (:181-191), stain matrix in float16 for the reconstruction (:196-198); covariance and
    eigenvectors float32 (:124), the 2x2 normal equations float32 (:160-176).  cuBLAS accumulates the float16 products in
    float32 and rounds once; numpy's float16 matmul is emulated the same way here (float32 product of float16 operands,
    rounded to float16).  Published accuracy of that mode: MAE ~0.05 grey levels against torchstain (docs/benchmarks.md,
    BASELINE.md section 1); this restatement is what tests/test_precision_modes_gpu.py holds ``precision="fast"`` to."""
    f16 = np.float16
    sm = np.asarray(stain_matrix, dtype=F32)
    tmc = np.asarray(target_max_conc, dtype=F32).reshape(-1)
    n_img, _, height, width = images.shape
    od_all = optical_density(to_unit_float(images))                       # float32 (:99)
    out = np.empty((n_img, 3, height, width), dtype=F32)
    for n in range(n_img):
        od = od_all[n].reshape(3, -1)                                     # (3, P)
        rows = od.T                                                       # (P, 3)
        keep = rows.min(axis=1) >= BETA                                   # :105-106 (float32 mask)
        if keep.sum() < 3:
            keep = np.ones_like(keep)
        vecs = plane_vectors(od_covariance(rows[keep]), signs)            # float32 covariance + eigh (:124)
        proj = (rows.astype(f16).astype(F32) @ vecs.astype(f16).astype(F32)).astype(f16)      # fp16 bmm (:135-137)
        phi = np.arctan2(proj[:, 1].astype(F32), proj[:, 0].astype(F32)).astype(f16)          # :140
        phi_kept = phi[keep].astype(F32)
        phi_lo, phi_hi = nearest_rank(phi_kept, ALPHA), nearest_rank(phi_kept, 100.0 - ALPHA)   # exact ranks of the fp16 values (:144-148)
        he = stain_vectors_from_angles(vecs, phi_lo, phi_hi)              # float32 (:151-161)
        a2 = (he.T @ he).astype(F32)                                      # 2x2 normal equations, float32 (:165-176)
        rhs = (he.T @ od).astype(F32)
        det = a2[0, 0] * a2[1, 1] - a2[0, 1] * a2[0, 1]
        c0 = (a2[1, 1] / det) * rhs[0] + (-a2[0, 1] / det) * rhs[1]
        c1 = (-a2[0, 1] / det) * rhs[0] + (a2[0, 0] / det) * rhs[1]
        c0h, c1h = c0.astype(f16), c1.astype(f16)                         # :181-183
        max_c = np.array([nearest_rank(c0h.astype(F32), 99), nearest_rank(c1h.astype(F32), 99)], dtype=F32)
        scale = (tmc / max_c).astype(f16)                                 # :191-192
        cn = np.stack([(c0h * scale[0]).astype(f16), (c1h * scale[1]).astype(f16)])             # (2, P) float16
        od_new = (sm.astype(f16).astype(F32) @ cn.astype(F32)).astype(f16).astype(F32)          # fp16 matmul (:196-198)
        out[n] = np.clip(IO * np.exp(-od_new), F32(0), F32(255)).reshape(3, height, width)
    return restore_dtype(out, images.dtype, in_0_255=True)


def macenko_fit(images: np.ndarray, *, signs=None) -> tuple[np.ndarray, np.ndarray]:
    """``compute_reference_stain_matrix_torch`` (torch_backend.py:463-519): pooled over the batch, no <3 fallback."""
    if images.ndim != 4 or images.shape[1] != 3:
        raise ValueError(f"Macenko fit expects NCHW with C=3, got shape {images.shape}")
    od = optical_density(to_unit_float(images))               # :475
    pooled = np.transpose(od, (1, 0, 2, 3)).reshape(3, 1, -1)  # :477  (3, 1, N*H*W)
    p = macenko_tile_params(pooled, allow_fallback=False, signs=signs)  # :483-516
    return p["he"], p["max_c"]


def apply_normalize_to_0_1(result: np.ndarray) -> np.ndarray:
    """``result / 255.0`` in the result dtype; uint8 promotes to float32 (normalizers/_template.py:111-112)."""
    if result.dtype == np.uint8:
        return result.astype(F32) / F32(255.0)
    return (result / result.dtype.type(255.0)).astype(result.dtype)


# --------------------------------------------------------------------------
# Reinhard                          (torch_backend.py:17-101, 304-355)
# --------------------------------------------------------------------------
_RGB2XYZ = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]], dtype=F32)  # :32
_XYZ2RGB = np.array([[3.2404542, -1.5371385, -0.4985314], [-0.9692660, 1.8760108, 0.0415560], [0.0556434, -0.2040259, 1.0572252]], dtype=F32)  # :89
_D65 = np.array([0.95047, 1.0, 1.08883], dtype=F32).reshape(1, 3, 1, 1)  # :37, :84


def rgb_to_lab(unit_rgb: np.ndarray) -> np.ndarray:
    """sRGB [0,1] -> scaled LAB (L*2.55, a+128, b+128), NCHW (torch_backend.py:17-60)."""
    rgb = unit_rgb.astype(F32)
    lin = np.where(rgb > F32(0.04045), np.power((rgb + F32(0.055)) / F32(1.055), F32(2.4)), rgb / F32(12.92))  # :28-29
    xyz = np.einsum("ij,njhw->nihw", _RGB2XYZ, lin).astype(F32)   # :34
    xyz = xyz / _D65                                              # :38
    f = np.where(xyz > F32(0.008856), np.power(xyz, F32(1.0 / 3.0)), F32(7.787) * xyz + F32(16.0 / 116.0))  # :41-42
    fx, fy, fz = f[:, 0:1], f[:, 1:2], f[:, 2:3]
    lum = (F32(116.0) * fy - F32(16.0)) * F32(2.55)               # :51
    a = F32(500.0) * (fx - fy) + F32(128.0)                       # :52
    b = F32(200.0) * (fy - fz) + F32(128.0)                       # :53
    return np.concatenate([lum, a, b], axis=1).astype(F32)


def lab_to_rgb(lab: np.ndarray) -> np.ndarray:
    """Inverse of :func:`rgb_to_lab`, clamped to [0,1] (torch_backend.py:63-101)."""
    lum = lab[:, 0:1] / F32(2.55)
    a = lab[:, 1:2] - F32(128.0)
    b = lab[:, 2:3] - F32(128.0)
    fy = (lum + F32(16.0)) / F32(116.0)
    fx = a / F32(500.0) + fy
    fz = fy - b / F32(200.0)

    def f_inv(t):
        return np.where(t > F32(0.2068966), t ** 3, (t - F32(16.0 / 116.0)) / F32(7.787))  # :78-80

    xyz = np.concatenate([f_inv(fx), f_inv(fy), f_inv(fz)], axis=1).astype(F32) * _D65
    lin = np.einsum("ij,njhw->nihw", _XYZ2RGB, xyz).astype(F32)
    with np.errstate(invalid="ignore"):
        rgb = np.where(lin > F32(0.0031308), F32(1.055) * np.power(lin, F32(1.0 / 2.4)) - F32(0.055), F32(12.92) * lin)  # :93-94
    return np.clip(rgb, F32(0), F32(1)).astype(F32)


def _pooled_mean_std(lab: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """mean / unbiased std over (N,H,W) per channel (torch_backend.py:320-321, 345-346)."""
    flat = np.transpose(lab, (1, 0, 2, 3)).reshape(3, -1).astype(np.float64)
    mean = flat.mean(axis=1)
    std = flat.std(axis=1, ddof=1) if flat.shape[1] > 1 else np.full(3, np.nan)
    return mean.astype(F32), std.astype(F32)


def reinhard_fit(images: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """``compute_reference_mean_std_torch`` (torch_backend.py:308-323)."""
    return _pooled_mean_std(rgb_to_lab(to_unit_float(images)))


def reinhard_transform(images: np.ndarray, ref_mean: np.ndarray, ref_std: np.ndarray) -> np.ndarray:
    """``ReinhardTorch.transform`` (torch_backend.py:325-355): output in the input dtype/range."""
    lab = rgb_to_lab(to_unit_float(images))
    mean, std = _pooled_mean_std(lab)                                          # :345-346
    rm = np.asarray(ref_mean, dtype=F32).reshape(1, 3, 1, 1)
    rs = np.asarray(ref_std, dtype=F32).reshape(1, 3, 1, 1)
    lab_n = ((lab - mean.reshape(1, 3, 1, 1)) / (std.reshape(1, 3, 1, 1) + F32(1e-8))) * rs + rm   # :349
    rgb = np.clip(lab_to_rgb(lab_n.astype(F32)), F32(0), F32(1))               # :351-353
    return restore_dtype(rgb, images.dtype, in_0_255=False)                    # :355


# --------------------------------------------------------------------------
# Histogram matching                (torch_backend.py:134-301)
# --------------------------------------------------------------------------
def _channels_first(images: np.ndarray, channel_axis: int) -> tuple[np.ndarray, bool]:
    if channel_axis == -1 or (channel_axis == 3 and images.ndim == 4):       # :182
        return np.transpose(images, (0, 3, 1, 2)), True
    return images, False


def images_to_uint8(images: np.ndarray) -> tuple[np.ndarray, bool]:
    """float -> trunc(clamp(x*255, 0, 255)) (torch_backend.py:115-120)."""
    if images.dtype == np.uint8:
        return images, False
    return np.trunc(np.clip(images.astype(F32) * F32(255.0), F32(0), F32(255))).astype(np.uint8), True


def _cumsum_f32(values: np.ndarray) -> np.ndarray:
    """torch.cumsum on CPU float32: running sum kept in double, each prefix rounded to float32."""
    return np.cumsum(values.astype(np.float64)).astype(F32)


def _torch_sum_f32(values: np.ndarray) -> np.float32:
    """``torch.sum`` of a contiguous float32 vector on CPU whose length is a multiple of 32 (the 256-bin histograms here):
    ATen's vectorised reduction -- four accumulators of eight lanes over blocks of 32, the accumulators added in order,
    then the lanes in order.  Not a running sum: the last bit differs from one for ~20 % of histograms (checked against
    torch 2.10 by tools/check_torch_sum.py)."""
    x = np.ascontiguousarray(values, dtype=F32)
    assert x.size % 32 == 0
    acc = np.zeros((4, 8), dtype=F32)
    for block in x.reshape(-1, 4, 8):
        acc = (acc + block).astype(F32)
    lanes = acc[0]
    for k in range(1, 4):
        lanes = (lanes + acc[k]).astype(F32)
    total = lanes[0]
    for lane in lanes[1:]:
        total = F32(total + lane)
    return F32(total)


def hm_fit(images: np.ndarray, channel_axis: int = 1) -> list[np.ndarray]:
    """Per-channel normalised 256-bin histograms (torch_backend.py:139-179); what ``transform`` receives."""
    chw, _ = _channels_first(images, channel_axis)
    u8, _ = images_to_uint8(chw)
    hists = []
    for c in range(u8.shape[1]):
        counts = np.bincount(u8[:, c].reshape(-1), minlength=256).astype(F32)
        hists.append(counts / (_torch_sum_f32(counts) + F32(1e-8)))          # :140-141
    return hists


def hm_lut(counts: np.ndarray, ref_hist: np.ndarray, num_pixels: int) -> np.ndarray:
    """256-entry float32 LUT for one channel (torch_backend.py:234-281)."""
    src_hist = counts.astype(F32) / F32(num_pixels + 1e-8)                   # :235
    src_cdf = _cumsum_f32(src_hist)                                          # :236
    ref = ref_hist.astype(F32)
    ref_cdf = _cumsum_f32(ref / (_torch_sum_f32(ref) + F32(1e-8)))            # :222-223
    values = np.arange(256, dtype=F32)
    idx = np.searchsorted(ref_cdf, src_cdf, side="left")                     # :260
    idx = np.clip(idx, 1, 255)                                               # :261
    q_lo, q_hi = ref_cdf[idx - 1], ref_cdf[idx]
    diff = q_hi - q_lo
    with np.errstate(divide="ignore", invalid="ignore"):
        alpha = np.where(diff > F32(1e-10), (src_cdf - q_lo) / diff, F32(0))  # :272-273
    lut = values[idx - 1] + alpha.astype(F32) * (values[idx] - values[idx - 1])  # :276
    lut = np.where(src_cdf <= ref_cdf[0], values[0], lut)                    # :268, :279
    lut = np.where(src_cdf >= ref_cdf[-1], values[-1], lut)                  # :269, :280
    return np.clip(lut, F32(0), F32(255)).astype(F32)                        # :281


def hm_transform(images: np.ndarray, ref_hists, channel_axis: int = 1, *, return_tables: bool = False):
    """``HistogramMatchingTorch.transform`` (torch_backend.py:194-301)."""
    chw, permuted = _channels_first(images, channel_axis)
    dtype = chw.dtype
    u8, scaled_back = images_to_uint8(chw)
    n_img, chans, height, width = u8.shape
    if isinstance(ref_hists, np.ndarray) and ref_hists.ndim == 1:
        ref_hists = [ref_hists] * chans
    out = np.empty((n_img, chans, height, width), dtype=F32)
    tables = {"counts": [], "lut": []}
    for c in range(chans):
        flat = u8[:, c].reshape(-1)
        counts = np.bincount(flat, minlength=256)                           # :234 (pooled over N*H*W)
        lut = hm_lut(counts, np.asarray(ref_hists[min(c, len(ref_hists) - 1)]), flat.size)
        out[:, c] = lut[flat].reshape(n_img, height, width)                 # :285
        tables["counts"].append(counts.astype(np.int64))
        tables["lut"].append(lut)
    if scaled_back:                                                          # :290-296
        out = np.clip(out / F32(255.0), F32(0), F32(1))
        result = restore_dtype(out, dtype, in_0_255=False)
    else:
        out = np.clip(out, F32(0), F32(255))
        result = restore_dtype(out, dtype, in_0_255=True)
    if permuted:
        result = np.transpose(result, (0, 2, 3, 1))
    return (result, tables) if return_tables else result
