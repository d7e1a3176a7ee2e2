"""Histogram matching against a reference image (API of stainx.HistogramMatching, incl. ``channel_axis``)."""
from __future__ import annotations

from typing import Any

from stainx_amd.normalizers._template import NormalizerTemplate


class HistogramMatching(NormalizerTemplate):
    engine = "HistogramMatchingHIP"
    # three normalised 256-bin histograms (one per channel); `_reference_histogram` is the first of them, `_ref_vals` /
    # `_ref_cdf` exist for attribute compatibility with the reference and stay unset (its transform never reads them)
    fitted_slots = ("_ref_histograms_256", "_reference_histogram", "_ref_vals", "_ref_cdf")

    def __init__(self, device: Any | None = None, backend: str | None = None, channel_axis: int = 1):
        self.channel_axis = channel_axis
        super().__init__(device=device, backend=backend)

    def engine_options(self) -> dict:
        return {"channel_axis": self.channel_axis}

    def learn(self, engine, images):
        per_channel = engine.compute_reference_histograms(images)
        return per_channel, per_channel[0], None, None

    def arguments(self) -> tuple:
        return (self._ref_histograms_256 if self._ref_histograms_256 else self._reference_histogram,)
