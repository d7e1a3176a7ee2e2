"""Seeded synthetic inputs for parity tests and benchmarks.

Macenko needs tiles with a well-defined stain plane: uniform RGB noise gives a
near-isotropic optical-density covariance, so the leading eigenspace is
eigensolver-dependent and parity is ill-posed (see the module note of the
reference's tests/torch_interface/test_correctness_against_references.py:30-34).
Tiles are therefore synthesised from the Beer-Lambert model, following the
recipe of that file's ``_synthetic_he_tile`` (:45-54): low-frequency
concentration maps, ``I = 240 * exp(-(HE @ C))`` rounded to uint8.

Everything is generated on the CPU with an explicit ``torch.Generator`` so the
same tiles are rebuilt on any machine.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F  # noqa: N812

# torchstain's default HERef (columns = hematoxylin, eosin); reference tests :41.
HE_REF = ((0.5626, 0.2159), (0.7201, 0.8012), (0.4062, 0.5581))
IO = 240.0


def he_tile(height: int, width: int, seed: int, he_scale: float = 1.0) -> torch.Tensor:
    """One (1,3,H,W) uint8 Beer-Lambert H&E tile."""
    gen = torch.Generator().manual_seed(int(seed))
    gh, gw = max(height // 8, 1), max(width // 8, 1)

    def smooth_map() -> torch.Tensor:
        coarse = torch.rand(1, 1, gh, gw, generator=gen)
        return F.interpolate(coarse, size=(height, width), mode="bilinear", align_corners=False).squeeze()

    c_h = smooth_map()
    c_e = smooth_map()
    conc = torch.stack([0.3 + 1.8 * c_h, 0.2 + 1.0 * c_e], dim=0).reshape(2, height, width)
    he = torch.tensor(HE_REF, dtype=torch.float32) * he_scale
    od = torch.einsum("cs,shp->chp", he, conc)
    return (IO * torch.exp(-od)).clamp(0, 255).round().to(torch.uint8).unsqueeze(0)


def he_batch(n_tiles: int, height: int, width: int, *, seed0: int = 1000, scale_step: float = 0.005) -> torch.Tensor:
    """(N,3,H,W) uint8 batch: tile i uses seed ``seed0+i`` and ``he_scale = 1 + scale_step*i``."""
    return torch.cat([he_tile(height, width, seed0 + i, 1.0 + scale_step * i) for i in range(n_tiles)], dim=0)


def reference_tile(height: int, width: int) -> torch.Tensor:
    """The fit target used by the benchmarks: seed 42, scale 1.0."""
    return he_tile(height, width, 42, 1.0)


def as_dtype(tiles_u8: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """uint8 tiles -> requested input dtype: floats are ``u8/255`` in float32, then cast."""
    if dtype == torch.uint8:
        return tiles_u8
    return (tiles_u8.float() / 255.0).to(dtype)


def noise_u8(shape: tuple[int, ...], seed: int) -> torch.Tensor:
    """Uniform uint8 noise, ``(rand*255).round()`` -- what the reference tests use for Reinhard / HM."""
    gen = torch.Generator().manual_seed(int(seed))
    return (torch.rand(*shape, generator=gen) * 255).round().to(torch.uint8)
