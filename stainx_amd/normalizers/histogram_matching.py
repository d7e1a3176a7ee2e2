"""Histogram-matching normaliser (mirrors reference normalizers/histogram_matching.py)."""
from __future__ import annotations

from typing import Any

from stainx_amd.normalizers._template import NormalizerTemplate


class HistogramMatching(NormalizerTemplate):
    def __init__(self, device: Any | None = None, backend: str | None = None, channel_axis: int = 1):
        self.channel_axis = channel_axis
        super().__init__(device=device, backend=backend)

    def _init_algorithm_attributes(self):
        self._reference_histogram = None
        self._ref_vals = None
        self._ref_cdf = None
        self._ref_histograms_256 = None

    def _get_torch_hip_class(self):
        from stainx_amd.backends.torch_hip_backend import HistogramMatchingHIP

        return HistogramMatchingHIP

    def _get_backend_kwargs(self) -> dict:
        return {"channel_axis": self.channel_axis}

    def _compute_reference_params(self, images: Any) -> None:
        self._ref_histograms_256 = self._get_backend_impl().compute_reference_histograms(images)
        self._reference_histogram = self._ref_histograms_256[0]

    def _get_reference_params(self) -> tuple:
        if self._ref_histograms_256:
            return (self._ref_histograms_256,)
        return (self._reference_histogram,)
