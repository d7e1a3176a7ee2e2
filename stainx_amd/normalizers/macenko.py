"""Macenko normaliser (mirrors reference normalizers/macenko.py:11-73)."""
from __future__ import annotations

from typing import Any

from stainx_amd.normalizers._template import NormalizerTemplate


class Macenko(NormalizerTemplate):
    """``normalize_to_0_1`` defaults to False here (output ~[0,255]); ``StainNormalizerTransform``
    defaults it to True.  ``precision`` is validated like the reference's (macenko.py:35-44)."""

    def __init__(self, device: Any | None = None, backend: str | None = None, normalize_to_0_1: bool = False, precision: str = "stable"):
        if precision not in ("stable", "fast"):
            raise ValueError(f"precision must be 'stable' or 'fast', got {precision!r}")
        self._precision = precision
        self.normalize_to_0_1 = normalize_to_0_1
        super().__init__(device=device, backend=backend)

    def _init_algorithm_attributes(self):
        self._stain_matrix = None
        self._concentration_matrix = None
        self._target_max_conc = None

    def _get_torch_hip_class(self):
        from stainx_amd.backends.torch_hip_backend import MacenkoHIP

        return MacenkoHIP

    def _get_backend_kwargs(self) -> dict:
        return {"precision": self._precision} if self._precision != "stable" else {}

    def _compute_reference_params(self, images: Any) -> None:
        self._stain_matrix, self._target_max_conc = self._get_backend_impl().compute_reference_stain_matrix(images)
        self._concentration_matrix = None

    def _get_reference_params(self) -> tuple:
        return (self._stain_matrix, self._target_max_conc)

    def _run_transform(self, impl, images, params):
        # `/255` after the cast to the input dtype (_template.py:111-112) is fused into the last kernel
        return impl.transform(images, *params, normalize_to_0_1=bool(self.normalize_to_0_1))
