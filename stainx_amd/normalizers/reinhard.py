"""Reinhard normaliser (mirrors reference normalizers/reinhard.py)."""
from __future__ import annotations

from typing import Any

from stainx_amd.normalizers._template import NormalizerTemplate


class Reinhard(NormalizerTemplate):
    def _init_algorithm_attributes(self):
        self._reference_mean = None
        self._reference_std = None

    def _get_torch_hip_class(self):
        from stainx_amd.backends.torch_hip_backend import ReinhardHIP

        return ReinhardHIP

    def _compute_reference_params(self, images: Any) -> None:
        self._reference_mean, self._reference_std = self._get_backend_impl().compute_reference_mean_std(images)

    def _get_reference_params(self) -> tuple:
        return (self._reference_mean, self._reference_std)
